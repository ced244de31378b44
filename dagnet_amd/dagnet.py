"""DAG-Net with the reference's constructor, state-dict keys and forward/sample signatures
(models/dagnet/model.py:10-270)."""
from __future__ import annotations

import torch.nn as nn

from . import _capi
from ._runtime import _GRAPH, as_goal_index, scene_ptr_from_sse
from ._vrnn_common import RolloutModule, make_graph, mlp2, resolve_adj


class DAGNet(RolloutModule):
    _has_batchnorm = True

    def __init__(self, args, n_max_agents):
        super().__init__()
        self.n_max_agents = n_max_agents
        self.n_layers = args.n_layers
        self.x_dim, self.h_dim, self.z_dim = args.x_dim, args.h_dim, args.z_dim
        self.d_dim = n_max_agents * 2
        self.g_dim = args.g_dim
        self.rnn_dim = args.rnn_dim
        self.graph_model = args.graph_model
        self.graph_hid = args.graph_hid
        self.adjacency_type = args.adjacency_type
        self.top_k_neigh = args.top_k_neigh
        self.sigma = args.sigma
        self.alpha = args.alpha
        self.n_heads = args.n_heads

        # goal generator: Linear -> ReLU -> Linear -> Softmax (indices 0 and 2 hold parameters)
        self.dec_goal = nn.Sequential(nn.Linear(self.d_dim + self.g_dim + self.rnn_dim, self.h_dim), nn.ReLU(),
                                      nn.Linear(self.h_dim, self.g_dim), nn.Softmax(dim=-1))
        if self.adjacency_type == 2 and self.top_k_neigh is None:
            raise Exception('Using KNN-similarity but top_k_neigh is not specified')
        self.graph_goals = make_graph(self.graph_model, self.g_dim, self.graph_hid, self.g_dim, self.alpha, self.n_heads)
        self.graph_hiddens = make_graph(self.graph_model, self.rnn_dim, self.graph_hid, self.rnn_dim, self.alpha, self.n_heads)
        self.lg_goals = nn.Sequential(nn.Linear(self.g_dim + self.g_dim, self.g_dim), nn.Softmax(dim=-1))
        self.lg_hiddens = nn.Linear(self.rnn_dim + self.rnn_dim, self.rnn_dim)
        self.phi_x = mlp2(self.x_dim, self.h_dim)
        self.phi_z = mlp2(self.z_dim, self.h_dim)
        self.enc = mlp2(self.h_dim + self.g_dim + self.rnn_dim, self.h_dim)
        self.enc_mean = nn.Linear(self.h_dim, self.z_dim)
        self.enc_logvar = nn.Linear(self.h_dim, self.z_dim)
        self.prior = mlp2(self.g_dim + self.rnn_dim, self.h_dim)
        self.prior_mean = nn.Linear(self.h_dim, self.z_dim)
        self.prior_logvar = nn.Linear(self.h_dim, self.z_dim)
        self.dec = mlp2(self.d_dim + self.g_dim + self.h_dim + self.rnn_dim, self.h_dim)
        self.dec_mean = nn.Linear(self.h_dim, self.x_dim)
        self.dec_logvar = nn.Linear(self.h_dim, self.x_dim)
        self.rnn = nn.GRU(self.h_dim + self.h_dim, self.rnn_dim, self.n_layers)

    def _hip_cfg(self):
        return dict(model=_capi.MODEL_DAGNET, x_dim=self.x_dim, h_dim=self.h_dim, z_dim=self.z_dim, rnn_dim=self.rnn_dim,
                    n_layers=self.n_layers, g_dim=self.g_dim, n_max_agents=self.n_max_agents,
                    graph_model=_GRAPH[self.graph_model], graph_hid=self.graph_hid, n_heads=self.n_heads or 0,
                    adjacency_type=self.adjacency_type, top_k_neigh=self.top_k_neigh or 0, sigma=float(self.sigma),
                    alpha=float(self.alpha or 0.0))

    def forward(self, traj, traj_rel, goals_ohe, seq_start_end, adj_out, eps=None, seed=None):
        """-> (KLD[1], NLL[1], cross_entropy[1], h[L,N,R]).  goals_ohe: one-hot [T,N,G] (or int indices [T,N]);
        adj_out: dense [T_adj,N,N] or BlockDiagAdj."""
        sp, a, dense = resolve_adj(adj_out, traj.shape[1], seq_start_end)
        goals = as_goal_index(goals_ohe, self.g_dim)
        losses, h = self._run_forward(sp, traj_rel, traj=traj, goals=goals, adj=a, adj_dense=dense, eps=eps, seed=seed)
        if isinstance(losses, tuple):
            return losses[0], losses[1], losses[2], h
        return losses[0:1], losses[1:2], losses[2:3], h

    def sample(self, samples_seq_len, h, x_abs_start, g_start, seq_start_end, eps=None, q=None, seed=None):
        e = None if eps is None else eps[None]
        qq = None if q is None else q[None]
        return self._run_sample(scene_ptr_from_sse(seq_start_end, h.shape[1]), 1, samples_seq_len, h, x0=x_abs_start,
                                g0=as_goal_index(g_start, self.g_dim), eps=e, q=qq, seed=seed)[0]

    def sample_k(self, n_samples, samples_seq_len, h, x_abs_start, g_start, seq_start_end, eps=None, q=None, seed=None,
                 force_goals=None, return_goals=False):
        return self._run_sample(scene_ptr_from_sse(seq_start_end, h.shape[1]), n_samples, samples_seq_len, h,
                                x0=x_abs_start, g0=as_goal_index(g_start, self.g_dim), eps=eps, q=q,
                                force_goals=force_goals, seed=seed, return_goals=return_goals)
