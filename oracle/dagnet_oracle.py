"""CPU oracle for the DAG-Net / A-VRNN / VRNN rollout hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; nothing under ``dagnet_amd/`` does.  The product path is the HIP library
(``dagnet_amd/csrc``) and it fails loudly when that library is missing.

What it is: a functional restatement (torch CPU, fp32) of the reference's
per-timestep algorithm with every random draw turned into an explicit input:

* ``eps``  -- the U[0,1) noise the reference draws with ``torch.rand_like`` in
  ``_reparameterize`` (models/dagnet/model.py:118-121, models/avrnn/model.py:87-90,
  models/baseline/model.py:65-68),
* ``q``    -- Exp(1) variates; ``torch.multinomial(p, 1)`` is ``argmax(p / q)``
  (SURVEY 8c'), used by the goal draw (models/utils/utils.py:200-207).

It keeps the reference's *dense* formulation of the graph layers
(``[N, N]`` attention over the whole flattened batch, block-diagonal adjacency)
so that timing it on host cores is a fair stand-in for the reference's CPU path.

Pinning: the reference ships no tests / golden vectors for this path
(SURVEY 4), so the oracle is pinned by outputs of the reference itself, run in
the build container by ``oracle/make_golden.py`` (noise injected by wrapping
``torch.rand_like`` / ``torch.multinomial``) and committed under
``tests/golden/``.  ``tests/test_oracle_golden.py`` checks this file against
them.

Weights are passed as the reference's ``state_dict`` (same key names).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    "make_cfg", "gat_forward", "gat_layer_forward", "gcn_forward", "gcn_layer_forward",
    "adj_fully_connected", "adj_distance_sim", "adj_knn_sim", "adjacency_pred", "adjacency_seq",
    "block_diag_dense", "one_hot", "sample_goal",
    "vrnn_sample", "vrnn_forward", "avrnn_sample", "avrnn_forward",
    "dagnet_sample", "dagnet_forward",
    "relative_to_abs", "ade_raw", "fde_raw", "min_over_k_sum", "evaluate_best_of_k",
]


def make_cfg(**kw):
    """Config namespace with the attribute names the reference's ``args`` uses."""
    d = dict(x_dim=2, h_dim=64, z_dim=32, rnn_dim=64, n_layers=2, g_dim=90,
             graph_model="gat", graph_hid=8, adjacency_type=1, top_k_neigh=None,
             sigma=1.2, alpha=0.2, n_heads=4, n_max_agents=10)
    d.update(kw)
    return SimpleNamespace(**d)


# --------------------------------------------------------------------------
# small building blocks
# --------------------------------------------------------------------------
def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def _mlp2(sd, name, x, slope=0.01):
    """Linear -> LeakyReLU -> Linear -> LeakyReLU (nn.Sequential indices 0 and 2)."""
    x = F.leaky_relu(_lin(sd, name + ".0", x), slope)
    return F.leaky_relu(_lin(sd, name + ".2", x), slope)


def _reparam(mean, logvar, eps):
    # models/dagnet/model.py:118-121 -- note eps is UNIFORM [0,1)
    return eps * torch.exp(0.5 * logvar) + mean


def _gru_step(sd, x, h, n_layers):
    """One seq_len=1 call of nn.GRU (gate order r,z,n); h is [L,N,R]."""
    out = []
    inp = x
    for l in range(n_layers):
        gi = F.linear(inp, sd[f"rnn.weight_ih_l{l}"], sd[f"rnn.bias_ih_l{l}"])
        gh = F.linear(h[l], sd[f"rnn.weight_hh_l{l}"], sd[f"rnn.bias_hh_l{l}"])
        i_r, i_z, i_n = gi.chunk(3, 1)
        h_r, h_z, h_n = gh.chunk(3, 1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        hn = (1.0 - z) * n + z * h[l]
        out.append(hn)
        inp = hn
    return torch.stack(out, 0)


_BN_HOOK = []      # train_step() records the BatchNorm inputs here (running-statistics update)


def _bn_eval(sd, prefix, x, training=False, eps=1e-5):
    if training:
        for hook in _BN_HOOK:
            hook(prefix, x)
        mu = x.mean(0)
        var = x.var(0, unbiased=False)
    else:
        mu = sd[prefix + ".running_mean"]
        var = sd[prefix + ".running_var"]
    return (x - mu) / torch.sqrt(var + eps) * sd[prefix + ".weight"] + sd[prefix + ".bias"]


# --------------------------------------------------------------------------
# graph layers (dense formulation, models/dagnet/gat/gat_layers.py:25-40,
# gat_model.py:18-22, gcn/gcn_layers.py:20-26, gcn_model.py:14-19)
# --------------------------------------------------------------------------
def gat_layer_forward(W, a, x, adj, alpha, concat):
    Wh = x @ W
    Fo = W.shape[1]
    s_src = Wh @ a[:Fo, 0]          # a[:F] . Wh_i
    s_dst = Wh @ a[Fo:, 0]          # a[F:] . Wh_j
    e = F.leaky_relu(s_src[:, None] + s_dst[None, :], alpha)
    e = torch.where(adj > 0, e, torch.full_like(e, -9e15))
    att = torch.softmax(e, dim=1)
    out = att @ Wh
    return F.elu(out) if concat else out


def gat_forward(sd, prefix, x, adj, alpha, n_heads, training=False):
    heads = [gat_layer_forward(sd[f"{prefix}.attention_{i}.W"], sd[f"{prefix}.attention_{i}.a"],
                               x, adj, alpha, True) for i in range(n_heads)]
    x = torch.cat(heads, 1)
    x = gat_layer_forward(sd[f"{prefix}.out_att.W"], sd[f"{prefix}.out_att.a"], x, adj, alpha, False)
    return torch.tanh(_bn_eval(sd, prefix + ".bn1", x, training))


def gcn_layer_forward(W, b, x, adj):
    out = adj @ (x @ W)
    return out + b if b is not None else out


def gcn_forward(sd, prefix, x, adj, training=False):
    x = gcn_layer_forward(sd[f"{prefix}.gc1.weight"], sd.get(f"{prefix}.gc1.bias"), x, adj)
    x = _bn_eval(sd, prefix + ".bn1", x, training)
    x = gcn_layer_forward(sd[f"{prefix}.gc2.weight"], sd.get(f"{prefix}.gc2.bias"), x, adj)
    x = _bn_eval(sd, prefix + ".bn2", x, training)
    return torch.tanh(x)


def _graph(sd, cfg, prefix, x, adj, training=False):
    if cfg.graph_model == "gat":
        return gat_forward(sd, prefix, x, adj, cfg.alpha, cfg.n_heads, training)
    return gcn_forward(sd, prefix, x, adj, training)


# --------------------------------------------------------------------------
# adjacency (models/utils/adjacency_matrix.py:21-28,49-58,86-103,108-152;
# block assembly models/utils/utils.py:290-302)
# --------------------------------------------------------------------------
def _sse_list(sse):
    return [(int(s), int(e)) for s, e in np.asarray(sse).reshape(-1, 2)]


def block_diag_dense(blocks):
    """fp64/any blocks -> one fp32 [N,N] matrix, zeros off the diagonal blocks."""
    n = sum(b.shape[0] for b in blocks)
    out = torch.zeros(n, n, dtype=torch.float32)
    o = 0
    for b in blocks:
        k = b.shape[0]
        out[o:o + k, o:o + k] = torch.as_tensor(b).to(torch.float32)
        o += k
    return out


def _pairwise_dist64(p):
    p = np.asarray(p, dtype=np.float64)
    d = p[:, None, :] - p[None, :, :]
    return np.sqrt((d * d).sum(-1))


def adj_fully_connected(sse):
    return block_diag_dense([np.ones((e - s, e - s)) for s, e in _sse_list(sse)])


def adj_distance_sim(sigma, sse, pos):
    pos = np.asarray(pos)
    return block_diag_dense([np.exp(-_pairwise_dist64(pos[s:e]) / sigma) for s, e in _sse_list(sse)])


def adj_knn_sim(k, sse, pos):
    pos = np.asarray(pos)
    blocks = []
    for s, e in _sse_list(sse):
        d = _pairwise_dist64(pos[s:e])
        nn_idx = np.argsort(d, axis=1, kind="stable")[:, :min(k, e - s)]
        m = np.zeros_like(d)
        np.put_along_axis(m, nn_idx, 1.0, axis=1)
        blocks.append(m)
    return block_diag_dense(blocks)


def adjacency_pred(cfg, sse, pos):
    if cfg.adjacency_type == 0:
        return adj_fully_connected(sse)
    if cfg.adjacency_type == 1:
        return adj_distance_sim(cfg.sigma, sse, pos)
    return adj_knn_sim(cfg.top_k_neigh, sse, pos)


def adjacency_seq(cfg, sse, traj):
    """[T,N,N] adjacency for every timestep of ``traj`` [T,N,2] (obs+pred GT)."""
    traj = np.asarray(traj)
    return torch.stack([adjacency_pred(cfg, sse, traj[t]) for t in range(traj.shape[0])], 0)


# --------------------------------------------------------------------------
# goals (models/utils/utils.py:182-207)
# --------------------------------------------------------------------------
def one_hot(idx, n):
    return F.one_hot(torch.as_tensor(idx).long(), n).to(torch.float32)


def sample_goal(weights, q):
    """Categorical draw as torch.multinomial does it: argmax(w / q), q ~ Exp(1)."""
    return torch.argmax(weights / q, dim=-1)


def _disposition(cfg, sse, pos):
    """Each agent row holds the flattened xy of all agents of its scene, zero padded to 2*n_max."""
    N = pos.shape[0]
    d = torch.zeros(N, 2 * cfg.n_max_agents, dtype=pos.dtype)
    for s, e in _sse_list(sse):
        d[s:e, : 2 * (e - s)] = pos[s:e].reshape(1, -1)
    return d


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------
def _kld(mu_e, lv_e, mu_p, lv_p):
    el = (lv_p - lv_e).sum(1) - mu_e.shape[1] + torch.exp(lv_e - lv_p).sum(1) \
        + ((mu_e - mu_p) ** 2 / torch.exp(lv_p)).sum(1)
    return (0.5 * el).mean()


def _nll(mu, lv, x):
    el = ((x - mu) ** 2 / torch.exp(lv)).sum(1) + x.shape[1] * math.log(2 * math.pi) + lv.sum(1)
    return (0.5 * el).mean()


# --------------------------------------------------------------------------
# VRNN (models/baseline/model.py:70-115,135-170)
# --------------------------------------------------------------------------
def vrnn_sample(sd, cfg, T, h, eps, trace=None):
    """eps: [T,N,Z]. Returns displacements [T,N,2]."""
    out = []
    for t in range(T):
        pr = _mlp2(sd, "prior", h[-1])
        z = _reparam(_lin(sd, "prior_mean", pr), _lin(sd, "prior_logvar", pr), eps[t])
        phi_z = _mlp2(sd, "phi_z", z)
        dec_mean = _lin(sd, "dec_mean", _mlp2(sd, "dec", torch.cat([phi_z, h[-1]], 1)))
        phi_x = _mlp2(sd, "phi_x", dec_mean)
        h = _gru_step(sd, torch.cat([phi_x, phi_z], 1), h, cfg.n_layers)
        out.append(dec_mean)
        if trace is not None:
            trace.append(dict(z=z, dec_mean=dec_mean, h=h.clone()))
    return torch.stack(out, 0)


def _posterior_step(sd, cfg, x_t, h, eps_t, g_t=None, d_t=None):
    """Shared teacher-forced step. Returns (kld_t, nll_t, phi_x, phi_z)."""
    hl = h[-1]
    phi_x = _mlp2(sd, "phi_x", x_t)
    enc_in = [phi_x, hl] if g_t is None else [phi_x, g_t, hl]
    en = _mlp2(sd, "enc", torch.cat(enc_in, 1))
    mu_e, lv_e = _lin(sd, "enc_mean", en), _lin(sd, "enc_logvar", en)
    pr = _mlp2(sd, "prior", hl if g_t is None else torch.cat([g_t, hl], 1))
    mu_p, lv_p = _lin(sd, "prior_mean", pr), _lin(sd, "prior_logvar", pr)
    z = _reparam(mu_e, lv_e, eps_t)
    phi_z = _mlp2(sd, "phi_z", z)
    dec_in = [phi_z, hl] if g_t is None else [d_t, g_t, phi_z, hl]
    de = _mlp2(sd, "dec", torch.cat(dec_in, 1))
    mu_d, lv_d = _lin(sd, "dec_mean", de), _lin(sd, "dec_logvar", de)
    return _kld(mu_e, lv_e, mu_p, lv_p), _nll(mu_d, lv_d, x_t), phi_x, phi_z


def vrnn_forward(sd, cfg, x, eps):
    """x: [T,N,2] relative coords, eps: [T-1,N,Z] (step t uses eps[t-1])."""
    T, N, _ = x.shape
    h = torch.zeros(cfg.n_layers, N, cfg.rnn_dim)
    kld = torch.zeros(1)
    nll = torch.zeros(1)
    for t in range(1, T):
        k, n, phi_x, phi_z = _posterior_step(sd, cfg, x[t], h, eps[t - 1])
        h = _gru_step(sd, torch.cat([phi_x, phi_z], 1), h, cfg.n_layers)
        kld += k
        nll += n
    return kld, nll, h


# --------------------------------------------------------------------------
# A-VRNN (models/avrnn/model.py:93-143,163-215)
# --------------------------------------------------------------------------
def _refine_last(sd, cfg, graph_prefix, blend_prefix, h, adj, training=False):
    h_g = _graph(sd, cfg, graph_prefix, h[-1], adj, training)
    new_last = _lin(sd, blend_prefix, torch.cat([h_g, h[-1]], 1))
    return torch.cat([h[:-1], new_last[None]], 0)


def avrnn_sample(sd, cfg, T, h, x_abs_start, sse, eps, trace=None):
    x_abs = x_abs_start
    out = []
    for t in range(T):
        pr = _mlp2(sd, "prior", h[-1])
        z = _reparam(_lin(sd, "prior_mean", pr), _lin(sd, "prior_logvar", pr), eps[t])
        phi_z = _mlp2(sd, "phi_z", z)
        dec_mean = _lin(sd, "dec_mean", _mlp2(sd, "dec", torch.cat([phi_z, h[-1]], 1)))
        phi_x = _mlp2(sd, "phi_x", dec_mean)
        h = _gru_step(sd, torch.cat([phi_x, phi_z], 1), h, cfg.n_layers)
        out.append(dec_mean)
        x_abs = x_abs + dec_mean                      # adjacency uses the UPDATED positions
        adj = adjacency_pred(cfg, sse, x_abs)
        h = _refine_last(sd, cfg, "graph", "lg", h, adj)
        if trace is not None:
            trace.append(dict(z=z, dec_mean=dec_mean, h=h.clone(), adj=adj))
    return torch.stack(out, 0)


def avrnn_forward(sd, cfg, x, adj, eps, training=False):
    T, N, _ = x.shape
    h = torch.zeros(cfg.n_layers, N, cfg.rnn_dim)
    kld = torch.zeros(1)
    nll = torch.zeros(1)
    for t in range(1, T):
        k, n, phi_x, phi_z = _posterior_step(sd, cfg, x[t], h, eps[t - 1])
        h = _gru_step(sd, torch.cat([phi_x, phi_z], 1), h, cfg.n_layers)
        h = _refine_last(sd, cfg, "graph", "lg", h, adj[t], training)
        kld += k
        nll += n
    return kld, nll, h


# --------------------------------------------------------------------------
# DAG-Net (models/dagnet/model.py:138-194,196-270)
# --------------------------------------------------------------------------
def _goal_head(sd, cfg, d_t, h_last, g_prev, adj, training=False):
    hid = F.relu(_lin(sd, "dec_goal.0", torch.cat([d_t, h_last, g_prev], 1)))   # ReLU here, not LeakyReLU
    dec_g = torch.softmax(_lin(sd, "dec_goal.2", hid), -1)
    g_graph = _graph(sd, cfg, "graph_goals", dec_g, adj, training)
    g_comb = torch.softmax(_lin(sd, "lg_goals.0", torch.cat([dec_g, g_graph], -1)), -1)
    return dec_g, g_graph, g_comb


def dagnet_sample(sd, cfg, T, h, x_abs_start, g_start, sse, eps, q, trace=None, force_goals=None):
    """g_start: one-hot [N,G]; eps [T,N,Z]; q [T,N,G].  ``force_goals`` ([T,N] int)
    overrides the categorical draw (used to test the continuous path in isolation)."""
    g_t = g_start
    x_abs = x_abs_start
    out = []
    for t in range(T):
        d_t = _disposition(cfg, sse, x_abs)
        adj = adjacency_pred(cfg, sse, x_abs)          # PRE-update positions, used for both graphs
        dec_g, g_graph, g_comb = _goal_head(sd, cfg, d_t, h[-1], g_t, adj)
        w = torch.exp(g_comb)                          # exp of probabilities (reference quirk)
        idx = sample_goal(w, q[t]) if force_goals is None else torch.as_tensor(force_goals[t]).long()
        g_t = one_hot(idx, cfg.g_dim)
        pr = _mlp2(sd, "prior", torch.cat([g_t, h[-1]], 1))
        z = _reparam(_lin(sd, "prior_mean", pr), _lin(sd, "prior_logvar", pr), eps[t])
        phi_z = _mlp2(sd, "phi_z", z)
        dec_mean = _lin(sd, "dec_mean", _mlp2(sd, "dec", torch.cat([d_t, g_t, phi_z, h[-1]], 1)))
        phi_x = _mlp2(sd, "phi_x", dec_mean)
        h = _gru_step(sd, torch.cat([phi_x, phi_z], 1), h, cfg.n_layers)
        h = _refine_last(sd, cfg, "graph_hiddens", "lg_hiddens", h, adj)
        x_abs = x_abs + dec_mean
        out.append(dec_mean)
        if trace is not None:
            ratio = w / q[t]
            top2 = torch.topk(ratio, 2, dim=-1).values
            trace.append(dict(dec_g=dec_g, g_graph=g_graph, g_comb=g_comb, goal=idx,
                              margin=(top2[:, 0] - top2[:, 1]) / top2[:, 0],
                              z=z, dec_mean=dec_mean, h=h.clone(), adj=adj))
    return torch.stack(out, 0)


def dagnet_forward(sd, cfg, traj, traj_rel, goals_ohe, sse, adj, eps, training=False):
    T, N, _ = traj.shape
    h = torch.zeros(cfg.n_layers, N, cfg.rnn_dim)
    kld = torch.zeros(1)
    nll = torch.zeros(1)
    ce = torch.zeros(1)
    for t in range(1, T):
        d_t = _disposition(cfg, sse, traj[t])
        g_t = goals_ohe[t]
        _, _, g_comb = _goal_head(sd, cfg, d_t, h[-1], goals_ohe[t - 1], adj[t], training)
        ce -= (g_t * g_comb).sum()
        k, n, phi_x, phi_z = _posterior_step(sd, cfg, traj_rel[t], h, eps[t - 1], g_t=g_t, d_t=d_t)
        h = _gru_step(sd, torch.cat([phi_x, phi_z], 1), h, cfg.n_layers)
        kld += k
        nll += n
        h = _refine_last(sd, cfg, "graph_hiddens", "lg_hiddens", h, adj[t], training)
    return kld, nll, ce, h


# --------------------------------------------------------------------------
# metrics (models/utils/utils.py:8-21,76-111; evaluate_helper/evaluate
# models/dagnet/evaluate_bsk.py:71-129)
# --------------------------------------------------------------------------
def relative_to_abs(rel, start):
    return torch.cumsum(rel, dim=0) + start[None]


def ade_raw(pred, gt):
    """[T,N,2] -> [N]: SUM over time of the per-step L2 error (not the mean)."""
    return torch.sqrt(((gt - pred) ** 2).sum(2)).sum(0)


def fde_raw(pred_last, gt_last):
    return torch.sqrt(((pred_last - gt_last) ** 2).sum(1))


def min_over_k_sum(err_kn, sse):
    """err_kn [K,N] -> sum over scenes of min over K of the scene's summed error."""
    tot = torch.zeros((), dtype=err_kn.dtype)
    for s, e in _sse_list(sse):
        tot = tot + err_kn[:, s:e].sum(1).min()
    return tot


def evaluate_best_of_k(samples_rel_k, start, gt, sse):
    """samples_rel_k [K,T,N,2] -> (ade_sum, fde_sum): the per-batch sums of ``evaluate``."""
    ades, fdes = [], []
    for k in range(samples_rel_k.shape[0]):
        a = relative_to_abs(samples_rel_k[k], start)
        ades.append(ade_raw(a, gt))
        fdes.append(fde_raw(a[-1], gt[-1]))
    return min_over_k_sum(torch.stack(ades), sse), min_over_k_sum(torch.stack(fdes), sse)


# --------------------------------------------------------------------------
# one training step (row T): models/dagnet/train_bsk.py:259-291, models/avrnn/train_bsk.py:231-262,
# models/baseline/train_bsk.py (same three lines without the graph)
# --------------------------------------------------------------------------
def train_step(sd, cfg, kind, batch, eps, kld_weight=1.0, ce_weight=1.0, lr=1e-3, clip=10.0, adam_state=None):
    """Train-mode forward (BatchNorm batch statistics), loss = kld_weight KLD + NLL (+ ce_weight CE), backward,
    clip_grad_norm_(clip), one Adam step (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay).

    batch: VRNN (x_rel,), A-VRNN (x_rel, adj[T,N,N]), DAG-Net (traj, traj_rel, goals_ohe, sse, adj).
    -> dict(kld, nll, ce, loss, grads{name}, total_norm, params{name} after the step, buffers{name} after the step)
    BatchNorm running statistics follow nn.BatchNorm1d (momentum 0.1, unbiased variance, one update per call)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    work = dict(sd)
    work.update(params)
    bn_inputs = []

    def bn_train_hook(prefix, x):
        bn_inputs.append((prefix, x.detach()))
    _BN_HOOK.append(bn_train_hook)
    try:
        if kind == "vrnn":
            kld, nll, _ = vrnn_forward(work, cfg, batch[0], eps)
            ce = torch.zeros(1)
            loss = kld_weight * kld + nll
        elif kind == "avrnn":
            kld, nll, _ = avrnn_forward(work, cfg, batch[0], batch[1], eps, training=True)
            ce = torch.zeros(1)
            loss = kld_weight * kld + nll
        else:
            kld, nll, ce, _ = dagnet_forward(work, cfg, *batch, eps, training=True)
            loss = kld_weight * kld + nll + ce_weight * ce
    finally:
        _BN_HOOK.pop()
    names = list(params)
    grads = torch.autograd.grad(loss.sum(), [params[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(params[k])) for k, g in zip(names, grads)}
    total_norm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(clip / (total_norm + 1e-6), max=1.0)
    b1, b2, aeps = 0.9, 0.999, 1e-8
    step = 1 if adam_state is None else adam_state["step"] + 1
    new_params, m_out, v_out = {}, {}, {}
    for k in names:
        g = grads[k] * coef
        m = (adam_state["m"][k] if adam_state else torch.zeros_like(g)) * b1 + (1 - b1) * g
        v = (adam_state["v"][k] if adam_state else torch.zeros_like(g)) * b2 + (1 - b2) * g * g
        denom = v.sqrt() / math.sqrt(1 - b2 ** step) + aeps
        new_params[k] = params[k].detach() - (lr / (1 - b1 ** step)) * m / denom
        m_out[k], v_out[k] = m, v
    buffers = {k: v.clone() for k, v in sd.items() if k not in params}
    for prefix, x in bn_inputs:
        n = x.shape[0]
        buffers[prefix + ".running_mean"] = 0.9 * buffers[prefix + ".running_mean"] + 0.1 * x.mean(0)
        buffers[prefix + ".running_var"] = 0.9 * buffers[prefix + ".running_var"] + 0.1 * x.var(0, unbiased=True) if n > 1 \
            else buffers[prefix + ".running_var"]
        if prefix + ".num_batches_tracked" in buffers:
            buffers[prefix + ".num_batches_tracked"] = buffers[prefix + ".num_batches_tracked"] + 1
    return dict(kld=kld.detach(), nll=nll.detach(), ce=ce.detach(), loss=loss.detach(), grads=grads, total_norm=total_norm,
                params=new_params, buffers=buffers, adam=dict(step=step, m=m_out, v=v_out))
