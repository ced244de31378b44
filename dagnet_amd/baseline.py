"""VRNN baseline with the reference's constructor, state-dict keys and forward/sample signatures
(models/baseline/model.py:7-170); the per-timestep loop runs in the HIP rollout kernel."""
from __future__ import annotations

import torch.nn as nn

from . import _capi
from ._vrnn_common import RolloutModule, mlp2


class VRNN(RolloutModule):
    def __init__(self, args):
        super().__init__()
        self.x_dim, self.z_dim, self.h_dim = args.x_dim, args.z_dim, args.h_dim
        self.rnn_dim, self.n_layers = args.rnn_dim, args.n_layers
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

    def _hip_cfg(self):
        return dict(model=_capi.MODEL_VRNN, x_dim=self.x_dim, h_dim=self.h_dim, z_dim=self.z_dim, rnn_dim=self.rnn_dim,
                    n_layers=self.n_layers, g_dim=0, n_max_agents=0, graph_model=0, graph_hid=0, n_heads=0,
                    adjacency_type=0, top_k_neigh=0, sigma=1.0, alpha=0.0)

    def forward(self, x, eps=None, seed=None):
        """x [T,N,x_dim] relative coords -> (KLD[1], NLL[1], h[L,N,R]).  eps [T-1,N,Z] optional injected noise."""
        losses, h = self._run_forward(self._single_agent_ptr(x.shape[1]), x, eps=eps, seed=seed)
        if isinstance(losses, tuple):
            return losses[0], losses[1], h
        return losses[0:1], losses[1:2], h

    def sample(self, samples_seq_len, h, eps=None, seed=None):
        """-> samples [T,N,x_dim].  eps [T,N,Z] optional injected noise."""
        e = None if eps is None else eps[None]
        return self._run_sample(self._single_agent_ptr(h.shape[1]), 1, samples_seq_len, h, eps=e, seed=seed)[0]

    def sample_k(self, n_samples, samples_seq_len, h, eps=None, seed=None):
        """All K best-of-K rollouts in one launch -> [K,T,N,x_dim]."""
        return self._run_sample(self._single_agent_ptr(h.shape[1]), n_samples, samples_seq_len, h, eps=eps, seed=seed)
