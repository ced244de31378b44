"""CPU: the oracle's training step (oracle.train_step) against the per-parameter gradients, total norm, Adam-updated
parameters and BatchNorm buffers captured from the reference in train() mode (tests/golden/train_*.npz)."""
import numpy as np
import pytest
import torch

from oracle import dagnet_oracle as orc
from tests import cases


def oracle_train_step(meta, sd):
    batch, eps = cases.build_train_inputs(meta)
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = batch
    a = meta["args"]
    cfg = orc.make_cfg(n_max_agents=meta["n_max_agents"], x_dim=2, **{k: v for k, v in a.items()})
    kind = meta["kind"]
    if kind == "vrnn":
        b = (obs_rel,)
    elif kind == "avrnn":
        b = (obs_rel, orc.adjacency_seq(cfg, sse, torch.cat((obs, gt), 0)))
    else:
        traj, rel = torch.cat((obs, gt), 0), torch.cat((obs_rel, gt_rel), 0)
        goals = torch.cat((obs_goals, gt_goals), 0).long()
        b = (traj, rel, orc.one_hot(goals, cfg.g_dim), sse, orc.adjacency_seq(cfg, sse, traj))
    return orc.train_step(sd, cfg, kind, b, eps, kld_weight=meta["kld_weight"], ce_weight=meta["ce_weight"], lr=meta["lr"],
                          clip=meta["clip"])


def check_step(out, gold, lr=1e-3, clip=10.0, grad_rtol=cases.GRAD_RTOL):
    for k in ("kld", "nll", "ce", "loss"):
        np.testing.assert_allclose(np.asarray(out[k]).reshape(-1), gold[k].reshape(-1), rtol=cases.LOSS_RTOL, atol=1e-5, err_msg=k)
    for key, want in gold.items():
        if key.startswith("grad/"):
            got = np.asarray(out["grads"][key[5:]])
            # analytically-zero gradients (a bias in front of a train-mode BatchNorm) are rounding noise that scales
            # with the whole step's gradient: allow 1e-6 of the total norm on top of the per-tensor bound
            tol = grad_rtol * np.abs(want).max() + cases.GRAD_ATOL + 1e-6 * float(gold["total_norm"])
            assert np.abs(got - want).max() <= tol, f"{key}: max abs diff {np.abs(got - want).max():.3e} > {tol:.3e}"
    np.testing.assert_allclose(float(out["total_norm"]), float(gold["total_norm"]), rtol=1e-4)
    check_after(out, gold, lr, clip)


def check_after(out, gold, lr, clip):
    """Parameters and BatchNorm buffers after the step.  The first Adam step moves a weight by lr * g / (|g| + 1e-8):
    where the clipped gradient is far above 1e-8 the update is +-lr and must match closely; where it is not, the
    update is only bounded by lr (its value is decided by rounding noise in g)."""
    coef = min(1.0, clip / (float(gold["total_norm"]) + 1e-6))
    for key, want in gold.items():
        if not key.startswith("after/"):
            continue
        name = key[6:]
        got = out["params"].get(name)
        if got is None:
            got = out["buffers"][name]
        got, want = np.asarray(got, dtype=np.float64), want.astype(np.float64)
        if "grad/" + name in gold:
            g = np.abs(gold["grad/" + name]) * coef
            tol = np.where(g > 1e-5, 2e-6 + 1e-5 * np.abs(want), 2.0 * lr + 1e-6)
        else:
            tol = 2e-5 + 1e-4 * np.abs(want)          # BatchNorm running statistics (fp32 column means of O(1) values)
        assert (np.abs(got - want) <= tol).all(), f"{key}: max abs diff {np.abs(got - want).max():.3e}"


@pytest.mark.parametrize("name", cases.list_train_cases())
def test_oracle_train_step_matches_reference(name):
    meta, gold = cases.load_golden(name)
    _, _, sd = cases.build_model(meta)
    out = oracle_train_step(meta, sd)
    check_step(out, gold, meta["lr"], meta["clip"])
