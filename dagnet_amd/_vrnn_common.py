"""Pieces shared by the three drop-in models: parameter containers with the reference's
state-dict keys, and the calls into the HIP rollout."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import _capi
from ._runtime import HipBacked, draw_seed, infer_scene_ptr_from_adj, scene_ptr_from_sse
from .gat import GAT
from .gcn import GCN


def mlp2(n_in, n_hid):
    """Linear -> LeakyReLU -> Linear -> LeakyReLU; indices 0 and 2 carry the parameters
    (models/dagnet/model.py:71-113)."""
    return nn.Sequential(nn.Linear(n_in, n_hid), nn.LeakyReLU(), nn.Linear(n_hid, n_hid), nn.LeakyReLU())


def make_graph(graph_model, nin, nhid, nout, alpha, n_heads):
    if graph_model == "gcn":
        return GCN(nin, nhid, nout)
    if graph_model == "gat":
        assert n_heads is not None
        assert alpha is not None
        return GAT(nin, nhid, nout, alpha, n_heads)
    raise ValueError(f"graph_model must be 'gat' or 'gcn', got {graph_model!r}")


class RolloutModule(nn.Module, HipBacked):
    """Common sample()/forward() plumbing. Sub-classes define the parameters and _hip_cfg()."""

    _has_batchnorm = False

    def _single_agent_ptr(self, n):
        return np.arange(n + 1, dtype=np.int32)

    def _run_sample(self, scene_ptr, K, T, h, x0=None, g0=None, eps=None, q=None, force_goals=None, seed=None,
                    return_goals=False):
        self._require_eval("sample")
        dev = h.device
        plan = self.hip_plan(dev, scene_ptr, K)
        if seed is None and (eps is None or (q is None and g0 is not None)):
            seed = draw_seed()
        with torch.no_grad():
            return _capi.sample(plan.model, plan, int(T), h.detach().float(),      # (plan.model: hip_plan has just checked the weights)
                                None if x0 is None else x0.detach().float(), g0, eps, q, force_goals,
                                seed or 0, return_goals)

    def _run_forward(self, scene_ptr, traj_rel, traj=None, goals=None, adj=None, adj_dense=True, eps=None, seed=None):
        if self.training and (self._has_batchnorm or
                              (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))):
            # model.train(): batch-statistics BatchNorm and a differentiable result -- the layer-by-layer operators
            # (training.py); the fused rollout kernels below are the eval-mode path
            from .training import forward_train
            kld, nll, ce, h = forward_train(self, scene_ptr, traj_rel, traj=traj, goals=goals,
                                            adj=None if adj is None else adj.float(), eps=eps)
            return (kld, nll, ce), h
        dev = traj_rel.device
        plan = self.hip_plan(dev, scene_ptr, 1)
        if seed is None and eps is None:
            seed = draw_seed()
        with torch.no_grad():
            losses, h = _capi.forward(plan.model, plan, traj_rel.detach().float(),
                                      None if traj is None else traj.detach().float(), goals, adj, adj_dense, eps, seed or 0)
        return losses.to(torch.float32), h


def resolve_adj(adj, n_rows, seq_start_end=None):
    """-> (scene_ptr, tensor, dense?) from a dense [T,N,N] tensor or a BlockDiagAdj."""
    from .adjacency_matrix import BlockDiagAdj
    if isinstance(adj, BlockDiagAdj):
        return adj.scene_ptr, adj.values, False
    if seq_start_end is not None:
        sp = scene_ptr_from_sse(seq_start_end, n_rows)
    else:
        sp = infer_scene_ptr_from_adj(adj)
    return sp, adj.float(), True
