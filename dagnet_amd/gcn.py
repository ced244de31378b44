"""Two-layer GCN with un-normalised adjacency weights, evaluated per scene on the HIP library.

Same classes / parameter names / signatures as the reference
(models/dagnet/gcn/gcn_layers.py:6-34, gcn_model.py:6-19)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _capi
from ._runtime import HipBacked
from .gat import _graph_module_forward


class GraphConvolution(nn.Module):
    """One graph convolution (weight [in,out], bias [out]; gcn_layers.py:6-34): ``adj @ (input @ weight) + bias`` with the
    adjacency VALUES as weights, per scene of the block-diagonal ``adj``.  Differentiable.  Inside ``GCN`` both layers and
    their BatchNorms run fused instead."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = nn.Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, input, adj, seq_start_end=None):
        """input [N,in], adj dense [N,N] block diagonal (or BlockDiagAdj) -> [N,out]."""
        from . import _ops
        from .gat import _layer_plan
        plan, a = _layer_plan(input, adj, seq_start_end)
        return _ops.AdjMatmulFn.apply(_ops.MatmulFn.apply(input.float(), self.weight), self.bias, a, plan)


class GCN(nn.Module, HipBacked):
    _has_batchnorm = True

    def __init__(self, nin, nhid, nout):
        super().__init__()
        self.nin, self.nhid, self.nout = nin, nhid, nout
        self.gc1 = GraphConvolution(nin, nhid)
        self.bn1 = nn.BatchNorm1d(nhid)
        self.gc2 = GraphConvolution(nhid, nout)
        self.bn2 = nn.BatchNorm1d(nout)

    def _hip_cfg(self):
        return dict(model=_capi.MODEL_GRAPH, x_dim=1, h_dim=self.nout, z_dim=1, rnn_dim=self.nin, n_layers=1,
                    g_dim=0, n_max_agents=0, graph_model=_capi.GRAPH_GCN, graph_hid=self.nhid, n_heads=0,
                    adjacency_type=0, top_k_neigh=0, sigma=1.0, alpha=0.0)

    def _hip_state(self):
        return {"graph." + k: v for k, v in self.state_dict().items()}

    def forward(self, x, adj, seq_start_end=None):
        self._require_eval("GCN.forward")
        return _graph_module_forward(self, x, adj, seq_start_end)
