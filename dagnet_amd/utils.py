"""Hot utilities with the reference's names and semantics (models/utils/utils.py:8-21,76-111,
182-207,290-302).  Numeric work runs in libdagnet_hip.so; one-hot / block-diag are pure layout."""
from __future__ import annotations

import torch

from . import _capi
from ._runtime import draw_seed


def relative_to_abs(traj_rel, start_pos):
    """(seq_len, batch, 2) displacements + (batch, 2) start -> absolute coords (seq_len, batch, 2)."""
    return _capi.relative_to_abs(traj_rel.float(), start_pos.float())


def average_displacement_error(pred_traj, pred_traj_gt, mode='sum'):
    """Sum over time of the per-step L2 error, per agent ('raw') or summed over the batch ('sum')."""
    r = _capi.displacement_errors(None, pred_traj.float(), pred_traj_gt.float())
    if mode == 'sum':
        return r["ade"].sum()
    elif mode == 'raw':
        return r["ade"]


def final_displacement_error(pred_pos, pred_pos_gt, mode='sum'):
    r = _capi.displacement_errors(None, pred_pos.float()[None], pred_pos_gt.float()[None])
    if mode == 'raw':
        return r["fde"]
    else:
        return r["fde"].sum()


def one_hot_encode(inds, N):
    dims = list(inds.shape) + [N]
    ret = torch.zeros(dims, dtype=torch.float32, device=inds.device)
    ret.scatter_(-1, inds.unsqueeze(-1).long(), 1)
    return ret


def to_goals_one_hot(original_goal, ohe_dim):
    return one_hot_encode(original_goal[:, :].data, ohe_dim)


def sample_multinomial(probs, q=None, seed=None):
    """One categorical draw per row of probs [batch, g_dim] -> one-hot [batch, g_dim].
    Same draw rule as torch.multinomial (argmax probs/q, q ~ Exp(1)); q may be injected."""
    if q is None and seed is None:
        seed = draw_seed()
    inds = _capi.categorical(probs.float(), q, seed or 0)
    return one_hot_encode(inds.long(), probs.size(-1))


def block_diag_irregular(matrices):
    """List of [n_s,n_s] (or [T,n_s,n_s]) blocks -> fp32 [N,N] (or [T,N,N]) with zeros elsewhere."""
    ns = [m.shape[-1] for m in matrices]
    n = int(sum(ns))
    lead = tuple(matrices[0].shape[:-2])
    dev = matrices[0].device if isinstance(matrices[0], torch.Tensor) else "cpu"
    v = torch.zeros(lead + (n, n), dtype=torch.float32, device=dev)
    o = 0
    for m, k in zip(matrices, ns):
        v[..., o:o + k, o:o + k] = torch.as_tensor(m).to(torch.float32)
        o += k
    return v
