"""Dense-signature GAT with per-scene (block-diagonal) evaluation on the HIP library.

Same classes, constructor arguments, parameter names and forward signatures as the reference
(models/dagnet/gat/gat_layers.py:5-40, gat_model.py:6-22; models/avrnn/gat/* is identical).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _capi
from ._runtime import HipBacked, infer_scene_ptr_from_adj, scene_ptr_from_sse


class GraphAttentionLayer(nn.Module):
    """One attention head (W [in,out], a [2*out,1]; gat_layers.py:5-40).

    ``forward(input, adj)`` evaluates the head on its own: ``h = input W`` (fp32 MFMA GEMM), then per scene of the
    block-diagonal ``adj`` the masked LeakyReLU logits, row softmax and ``attention @ h`` (+ ELU when ``concat``) in one
    kernel - the reference's [N,N,2F] ``a_input`` tensor is never built.  Differentiable (the training operators of _ops).
    Inside ``GAT`` all heads + out_att + BatchNorm + tanh run fused instead."""

    def __init__(self, in_features, out_features, alpha, concat=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.alpha = alpha
        self.concat = concat
        self.W = nn.Parameter(torch.zeros(in_features, out_features))
        nn.init.xavier_uniform_(self.W.data, gain=1.414)
        self.a = nn.Parameter(torch.zeros(2 * out_features, 1))
        nn.init.xavier_uniform_(self.a.data, gain=1.414)

    def forward(self, input, adj, seq_start_end=None):
        """input [N,in], adj dense [N,N] block diagonal (or BlockDiagAdj) -> [N,out]."""
        from . import _ops
        plan, a = _layer_plan(input, adj, seq_start_end)
        h = _ops.MatmulFn.apply(input.float(), self.W)
        return _ops.GatAttnFn.apply(h, _ops.StackFn.apply(True, self.a), a, plan, 1, self.out_features, float(self.alpha),
                                    bool(self.concat))


def _layer_plan(x, adj, seq_start_end):
    """Scene tables + adjacency operand (compact values or the dense matrix) for the stand-alone layers."""
    from .adjacency_matrix import BlockDiagAdj
    from .training import table_plan
    if not x.is_cuda:
        raise _capi.DagnetHipError(f"input lives on {x.device}; the HIP path needs a ROCm device and has no CPU fallback")
    if isinstance(adj, BlockDiagAdj):
        sp, a = adj.scene_ptr, adj.values.reshape(-1).float()
    else:
        sp = scene_ptr_from_sse(seq_start_end, x.shape[0]) if seq_start_end is not None else infer_scene_ptr_from_adj(adj)
        a = adj.float().contiguous()
    return table_plan(sp, x.device), a


class GAT(nn.Module, HipBacked):
    """Dense version of GAT: nheads x head(concat) -> out_att -> BatchNorm1d -> tanh."""

    _has_batchnorm = True

    def __init__(self, nin, nhid, nout, alpha, nheads):
        super().__init__()
        self.nin, self.nhid, self.nout, self.alpha, self.nheads = nin, nhid, nout, alpha, nheads
        self.attentions = [GraphAttentionLayer(nin, nhid, alpha=alpha, concat=True) for _ in range(nheads)]
        for i, attention in enumerate(self.attentions):
            self.add_module("attention_{}".format(i), attention)
        self.out_att = GraphAttentionLayer(nhid * nheads, nout, alpha=alpha, concat=False)
        self.bn1 = nn.BatchNorm1d(nout)

    def _hip_cfg(self):
        return dict(model=_capi.MODEL_GRAPH, x_dim=1, h_dim=self.nout, z_dim=1, rnn_dim=self.nin, n_layers=1,
                    g_dim=0, n_max_agents=0, graph_model=_capi.GRAPH_GAT, graph_hid=self.nhid, n_heads=self.nheads,
                    adjacency_type=0, top_k_neigh=0, sigma=1.0, alpha=float(self.alpha))

    def _hip_state(self):
        return {"graph." + k: v for k, v in self.state_dict().items()}

    def forward(self, x, adj, seq_start_end=None):
        """x [N,nin], adj dense [N,N] (block diagonal) or compact BlockDiagAdj -> [N,nout].

        ``seq_start_end`` (optional, not in the reference signature) names the blocks; when it is
        omitted they are inferred from the support of ``adj``."""
        self._require_eval("GAT.forward")
        return _graph_module_forward(self, x, adj, seq_start_end)


def _graph_module_forward(mod, x, adj, seq_start_end):
    from .adjacency_matrix import BlockDiagAdj
    if isinstance(adj, BlockDiagAdj):
        sp, a, dense = adj.scene_ptr, adj.values.reshape(-1), False
    else:
        sp = scene_ptr_from_sse(seq_start_end, x.shape[0]) if seq_start_end is not None else infer_scene_ptr_from_adj(adj)
        a, dense = adj, True
    plan = mod.hip_plan(x.device, sp, 1)
    return _capi.graph_forward(mod.hip_model(x.device), plan, 0, x.float(), a.float(), dense)
