"""A-VRNN with the reference's constructor, state-dict keys and forward/sample signatures
(models/avrnn/model.py:11-215)."""
from __future__ import annotations

import torch.nn as nn

from . import _capi
from ._runtime import _GRAPH, scene_ptr_from_sse
from ._vrnn_common import RolloutModule, make_graph, mlp2, resolve_adj


class AVRNN(RolloutModule):
    _has_batchnorm = True

    def __init__(self, args):
        super().__init__()
        self.x_dim, self.z_dim, self.h_dim = args.x_dim, args.z_dim, args.h_dim
        self.rnn_dim, self.n_layers = args.rnn_dim, args.n_layers
        self.graph_model = args.graph_model
        self.adjacency_type = args.adjacency_type
        self.top_k_neigh = args.top_k_neigh
        self.sigma = args.sigma
        self.graph_hid = args.graph_hid
        self.alpha = args.alpha
        self.n_heads = args.n_heads
        self.phi_x = mlp2(self.x_dim, self.h_dim)
        self.phi_z = mlp2(self.z_dim, self.h_dim)
        self.enc = mlp2(self.h_dim + self.rnn_dim, self.h_dim)
        self.enc_mean = nn.Linear(self.h_dim, self.z_dim)
        self.enc_logvar = nn.Linear(self.h_dim, self.z_dim)
        self.prior = mlp2(self.rnn_dim, self.h_dim)
        self.prior_mean = nn.Linear(self.h_dim, self.z_dim)
        self.prior_logvar = nn.Linear(self.h_dim, self.z_dim)
        self.dec = mlp2(self.h_dim + self.rnn_dim, self.h_dim)
        self.dec_mean = nn.Linear(self.h_dim, self.x_dim)
        self.dec_logvar = nn.Linear(self.h_dim, self.x_dim)
        self.rnn = nn.GRU(self.h_dim + self.h_dim, self.rnn_dim, self.n_layers)
        self.graph = make_graph(self.graph_model, self.rnn_dim, self.graph_hid, self.rnn_dim, self.alpha, self.n_heads)
        self.lg = nn.Linear(self.rnn_dim + self.rnn_dim, self.rnn_dim)

    def _hip_cfg(self):
        return dict(model=_capi.MODEL_AVRNN, x_dim=self.x_dim, h_dim=self.h_dim, z_dim=self.z_dim, rnn_dim=self.rnn_dim,
                    n_layers=self.n_layers, g_dim=0, n_max_agents=0, graph_model=_GRAPH[self.graph_model],
                    graph_hid=self.graph_hid, n_heads=self.n_heads or 0, adjacency_type=self.adjacency_type,
                    top_k_neigh=self.top_k_neigh or 0, sigma=float(self.sigma), alpha=float(self.alpha or 0.0))

    def forward(self, x, adj, seq_start_end=None, eps=None, seed=None):
        """x [T,N,2] relative coords, adj dense [T_adj,N,N] or BlockDiagAdj -> (KLD, NLL, h).
        The reference signature has no seq_start_end; without it the scenes are read off adj's support."""
        sp, a, dense = resolve_adj(adj, x.shape[1], seq_start_end)
        losses, h = self._run_forward(sp, x, adj=a, adj_dense=dense, eps=eps, seed=seed)
        if isinstance(losses, tuple):
            return losses[0], losses[1], h
        return losses[0:1], losses[1:2], h

    def sample(self, samples_seq_len, h, x_abs_start, seq_start_end, eps=None, seed=None):
        e = None if eps is None else eps[None]
        return self._run_sample(scene_ptr_from_sse(seq_start_end, h.shape[1]), 1, samples_seq_len, h, x0=x_abs_start,
                                eps=e, seed=seed)[0]

    def sample_k(self, n_samples, samples_seq_len, h, x_abs_start, seq_start_end, eps=None, seed=None):
        return self._run_sample(scene_ptr_from_sse(seq_start_end, h.shape[1]), n_samples, samples_seq_len, h,
                                x0=x_abs_start, eps=eps, seed=seed)
