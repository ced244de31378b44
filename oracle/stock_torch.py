"""CPU baseline stand-in (TEST / BENCH INFRASTRUCTURE - never imported by dagnet_amd/): the reference's ``sample()`` call
pattern assembled from STOCK ``torch.nn`` layers, op for op as the reference issues them on its PyTorch CPU path -
SURVEY 8(d) "a stock-PyTorch-CPU model assembled from torch.nn layers with the reference's layer shapes".

Unlike oracle/dagnet_oracle.py (which computes the attention logits by broadcast) this keeps what makes the reference
slow, because that is what a user of the reference pays today:
  * one attention head materialises the [N, N, 2F] pair tensor over the WHOLE flattened batch and multiplies it by ``a``
    (models/dagnet/gat/gat_layers.py:25-40, the repeat / cat of line 29);
  * the adjacency is rebuilt every step, scene by scene, with scipy's fp64 ``distance_matrix`` and assembled into one dense
    [N, N] fp32 block-diagonal tensor in a Python loop (models/utils/adjacency_matrix.py:21-28,49-58,86-103,
    models/utils/utils.py:290-302);
  * ``nn.GRU`` is called with a length-1 sequence per step, goals are drawn with ``torch.multinomial`` on
    ``exp(probabilities)`` and the latent noise with ``torch.rand_like`` (models/dagnet/model.py:196-270,
    models/avrnn/model.py:163-215, models/baseline/model.py:135-170).
Sub-module names equal the reference's state-dict keys, so the same ``state_dict`` loads.  Pinned by
tests/test_stock_baseline.py against the golden ``samples`` captured from the reference itself (noise injected by wrapping
torch.rand_like / torch.multinomial, as oracle/make_golden.py does)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.spatial import distance_matrix


class DenseGatHead(nn.Module):
    def __init__(self, n_in, n_out, alpha, concat):
        super().__init__()
        self.W = nn.Parameter(torch.zeros(n_in, n_out))
        self.a = nn.Parameter(torch.zeros(2 * n_out, 1))
        self.alpha, self.concat, self.n_out = alpha, concat, n_out

    def forward(self, x, adj):
        h = x.mm(self.W)
        n = h.shape[0]
        pairs = torch.cat([h.repeat(1, n).view(n * n, -1), h.repeat(n, 1)], dim=1).view(n, n, 2 * self.n_out)      # [N,N,2F]
        e = F.leaky_relu(pairs.matmul(self.a).squeeze(2), self.alpha)
        att = F.softmax(torch.where(adj > 0, e, torch.full_like(e, -9e15)), dim=1)
        out = att.matmul(h)
        return F.elu(out) if self.concat else out


class DenseGat(nn.Module):
    def __init__(self, n_in, n_hid, n_out, alpha, n_heads):
        super().__init__()
        self.n_heads = n_heads
        for i in range(n_heads):
            self.add_module(f"attention_{i}", DenseGatHead(n_in, n_hid, alpha, True))
        self.out_att = DenseGatHead(n_hid * n_heads, n_out, alpha, False)
        self.bn1 = nn.BatchNorm1d(n_out)

    def forward(self, x, adj):
        x = torch.cat([getattr(self, f"attention_{i}")(x, adj) for i in range(self.n_heads)], dim=1)
        return torch.tanh(self.bn1(self.out_att(x, adj)))


class DenseGcnLayer(nn.Module):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(n_in, n_out))
        self.bias = nn.Parameter(torch.zeros(n_out))

    def forward(self, x, adj):
        return adj.mm(x.mm(self.weight)) + self.bias


class DenseGcn(nn.Module):
    def __init__(self, n_in, n_hid, n_out):
        super().__init__()
        self.gc1, self.bn1 = DenseGcnLayer(n_in, n_hid), nn.BatchNorm1d(n_hid)
        self.gc2, self.bn2 = DenseGcnLayer(n_hid, n_out), nn.BatchNorm1d(n_out)

    def forward(self, x, adj):
        return torch.tanh(self.bn2(self.gc2(self.bn1(self.gc1(x, adj)), adj)))


def _mlp(n_in, n_hid):
    return nn.Sequential(nn.Linear(n_in, n_hid), nn.LeakyReLU(), nn.Linear(n_hid, n_hid), nn.LeakyReLU())


def dense_adjacency(kind, sigma, top_k, sse, pos):
    """Per scene: fp64 distances (scipy) -> weights -> one dense fp32 [N,N] block-diagonal tensor."""
    n_rows = int(sse[-1][1])
    out = torch.zeros(n_rows, n_rows)
    p = pos.detach().cpu().numpy()
    for a, b in sse:
        a, b = int(a), int(b)
        if kind == 0:
            blk = np.ones((b - a, b - a))
        else:
            d = distance_matrix(p[a:b], p[a:b])
            if kind == 1:
                blk = np.exp(-d / sigma)
            else:
                blk = np.zeros_like(d)
                nn_idx = np.argsort(d, axis=1)[:, :min(top_k, b - a)]
                np.put_along_axis(blk, nn_idx, 1.0, axis=1)
        out[a:b, a:b] = torch.from_numpy(blk)
    return out


class StockRollout(nn.Module):
    """kind in {'vrnn', 'avrnn', 'dagnet'}; ``args`` as the reference constructors read it."""

    def __init__(self, kind, args, n_max_agents=0):
        super().__init__()
        self.kind, self.a = kind, args
        H, Z, R, X = args.h_dim, args.z_dim, args.rnn_dim, args.x_dim
        G = args.g_dim if kind == "dagnet" else 0
        D = 2 * n_max_agents if kind == "dagnet" else 0
        self.G, self.D = G, D
        self.phi_x, self.phi_z = _mlp(X, H), _mlp(Z, H)
        self.enc, self.enc_mean, self.enc_logvar = _mlp(H + G + R, H), nn.Linear(H, Z), nn.Linear(H, Z)
        self.prior, self.prior_mean, self.prior_logvar = _mlp(G + R, H), nn.Linear(H, Z), nn.Linear(H, Z)
        self.dec, self.dec_mean, self.dec_logvar = _mlp(D + G + H + R, H), nn.Linear(H, X), nn.Linear(H, X)
        self.rnn = nn.GRU(2 * H, R, args.n_layers)

        def graph(n):
            if args.graph_model == "gcn":
                return DenseGcn(n, args.graph_hid, n)
            return DenseGat(n, args.graph_hid, n, args.alpha, args.n_heads)
        if kind == "avrnn":
            self.graph, self.lg = graph(R), nn.Linear(2 * R, R)
        if kind == "dagnet":
            self.dec_goal = nn.Sequential(nn.Linear(D + R + G, H), nn.ReLU(), nn.Linear(H, G), nn.Softmax(dim=-1))
            self.graph_goals, self.graph_hiddens = graph(G), graph(R)
            self.lg_goals = nn.Sequential(nn.Linear(2 * G, G), nn.Softmax(dim=-1))
            self.lg_hiddens = nn.Linear(2 * R, R)

    def _adj(self, sse, pos):
        a = self.a
        return dense_adjacency(a.adjacency_type, a.sigma, a.top_k_neigh, sse, pos)

    @torch.no_grad()
    def sample(self, T, h, x_abs=None, g_ohe=None, sse=None):
        kind, N = self.kind, h.shape[1]
        out = torch.zeros(T, N, self.a.x_dim)
        if kind == "dagnet":
            d_t = torch.zeros(N, self.D)
            for a, b in sse:
                a, b = int(a), int(b)
                d_t[a:b, :2 * (b - a)] = x_abs[a:b].reshape(-1).repeat(b - a, 1)
            g_t = g_ohe
        for t in range(T):
            if kind == "dagnet":
                adj = self._adj(sse, x_abs)
                dec_g = self.dec_goal(torch.cat([d_t, h[-1], g_t], 1))
                g_comb = self.lg_goals(torch.cat((dec_g, self.graph_goals(dec_g, adj)), dim=-1))
                idx = torch.multinomial(torch.exp(g_comb), 1).squeeze()
                g_t = torch.zeros(N, self.G).scatter_(1, idx.view(-1, 1), 1.0)
                pr = self.prior(torch.cat([g_t, h[-1]], 1))
            else:
                pr = self.prior(h[-1])
            mean, logvar = self.prior_mean(pr), self.prior_logvar(pr)
            z = torch.rand_like(mean).mul(logvar.mul(0.5).exp()).add(mean)
            phi_z = self.phi_z(z)
            if kind == "dagnet":
                dec = self.dec(torch.cat([d_t, g_t, phi_z, h[-1]], 1))
            else:
                dec = self.dec(torch.cat([phi_z, h[-1]], 1))
            x_t = self.dec_mean(dec)
            self.dec_logvar(dec)                          # computed (and discarded) by the reference too
            out[t] = x_t
            _, h = self.rnn(torch.cat([self.phi_x(x_t), phi_z], 1).unsqueeze(0), h)
            if kind == "avrnn":
                x_abs = x_abs + x_t                       # A-VRNN: adjacency from the UPDATED positions
                h[-1] = self.lg(torch.cat((self.graph(h[-1].clone(), self._adj(sse, x_abs)), h[-1]), dim=-1))
            elif kind == "dagnet":
                adj = self._adj(sse, x_abs)               # DAG-Net: still the positions before this step's displacement
                h[-1] = self.lg_hiddens(torch.cat((self.graph_hiddens(h[-1].clone(), adj), h[-1]), dim=-1))
                x_abs = x_abs + x_t
                step = torch.zeros(N, self.D)
                for a, b in sse:
                    a, b = int(a), int(b)
                    step[a:b, :2 * (b - a)] = x_t[a:b].reshape(-1).repeat(b - a, 1)
                d_t = d_t + step
        return out
