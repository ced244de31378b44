"""Adjacency construction on the device, same function names/arguments as the reference
(models/utils/adjacency_matrix.py:21-28,49-58,86-103,108-152).

The reference returns dense block-diagonal fp32 tensors ([N,N] or [T,N,N]) built with scipy on the
host; here one HIP kernel writes them (fp64 distance and exp, cast to fp32 on store, like the
reference's assignment into a float32 buffer, models/utils/utils.py:297-301).  ``compact=True``
(an addition) returns only the per-scene blocks as a ``BlockDiagAdj``, which the models accept in
place of the dense tensor: at N=5120 the dense form is 105 MB per timestep.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import _capi
from ._runtime import scene_ptr_from_sse


@dataclass
class BlockDiagAdj:
    """Per-scene adjacency blocks: values [T, sum_s n_s^2] (row-major n_s x n_s per scene)."""
    scene_ptr: np.ndarray
    values: torch.Tensor

    def cuda(self, *a, **k):      # the reference calls .cuda() on what these functions return
        return self

    def __getitem__(self, t):
        return BlockDiagAdj(self.scene_ptr, self.values[t:t + 1])

    def to_dense(self):
        sp = self.scene_ptr
        n = int(sp[-1])
        out = torch.zeros(self.values.shape[0], n, n, dtype=self.values.dtype, device=self.values.device)
        o = 0
        for s in range(len(sp) - 1):
            b, k = int(sp[s]), int(sp[s + 1] - sp[s])
            out[:, b:b + k, b:b + k] = self.values[:, o:o + k * k].reshape(-1, k, k)
            o += k * k
        return out


def _device_of(t, device):
    if device is not None:
        return torch.device(device)
    if isinstance(t, torch.Tensor) and t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise _capi.DagnetHipError("adjacency construction runs on a ROCm device; none is available (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _build(adj_type, sigma, k, seq_start_end, pos_tnc, T, compact, device):
    sp = scene_ptr_from_sse(seq_start_end)
    dev = _device_of(pos_tnc if pos_tnc is not None else seq_start_end, device)
    plan = _capi.HipPlan(None, sp, 1, device=dev)
    if pos_tnc is not None:
        pos_tnc = pos_tnc.detach().to(dev, torch.float32).contiguous()
    vals = _capi.adjacency(plan, adj_type, sigma, k, pos_tnc, T, dense=not compact)
    return BlockDiagAdj(sp, vals) if compact else vals


def adjs_fully_connected_pred(seq_start_end, compact=False, device=None):
    out = _build(0, 1.0, 0, seq_start_end, None, 1, compact, device)
    return out if compact else out[0]


def adjs_distance_sim_pred(sigma, seq_start_end, pred_traj, compact=False, device=None):
    out = _build(1, sigma, 0, seq_start_end, pred_traj[None], 1, compact, device)
    return out if compact else out[0]


def adjs_knn_sim_pred(top_k_neigh, seq_start_end, pred_traj, compact=False, device=None):
    out = _build(2, 1.0, top_k_neigh, seq_start_end, pred_traj[None], 1, compact, device)
    return out if compact else out[0]


def compute_adjs(args, seq_start_end, compact=False, device=None):
    T = args.obs_len + args.pred_len
    one = _build(0, 1.0, 0, seq_start_end, None, 1, compact, device)
    if compact:
        return BlockDiagAdj(one.scene_ptr, one.values.expand(T, -1).contiguous())
    return one.expand(T, -1, -1).contiguous()


def compute_adjs_knnsim(args, seq_start_end, obs_traj, pred_traj_gt, compact=False, device=None):
    traj = torch.cat((obs_traj, pred_traj_gt))[: args.obs_len + args.pred_len]
    return _build(2, 1.0, args.top_k_neigh, seq_start_end, traj, traj.shape[0], compact, device)


def compute_adjs_distsim(args, seq_start_end, obs_traj, pred_traj_gt, compact=False, device=None):
    traj = torch.cat((obs_traj, pred_traj_gt))[: args.obs_len + args.pred_len]
    return _build(1, args.sigma, 0, seq_start_end, traj, traj.shape[0], compact, device)
