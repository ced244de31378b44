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


def _thin(key, gold, got):
    """Fixtures of models with millions of parameters keep every 61st element of a large tensor (key + '@thin') and its L2 norm
    (key + '@norm'): -> (what to compare, the stored values), after checking the norm of the whole tensor."""
    if key in gold:
        return np.asarray(got), gold[key]
    got = np.asarray(got)
    want_norm = float(gold[key + "@norm"])
    got_norm = float(np.sqrt((got.astype(np.float64) ** 2).sum()))
    assert abs(got_norm - want_norm) <= 1e-3 * want_norm + 1e-6, f"{key}: norm {got_norm:.6e} against {want_norm:.6e}"
    return got.reshape(-1)[::61], gold[key + "@thin"]


def _names(gold, prefix):
    return sorted({k.split("@")[0] for k in gold if k.startswith(prefix)})


def check_step(out, gold, lr=1e-3, clip=10.0, grad_rtol=cases.GRAD_RTOL):
    for k in ("kld", "nll", "ce", "loss"):
        np.testing.assert_allclose(np.asarray(out[k]).reshape(-1), gold[k].reshape(-1), rtol=cases.LOSS_RTOL, atol=1e-5, err_msg=k)
    for key in _names(gold, "grad/"):
        got, want = _thin(key, gold, out["grads"][key[5:]])
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
    for key in _names(gold, "after/"):
        name = key[6:]
        got = out["params"].get(name)
        if got is None:
            got = out["buffers"][name]
        got, want = _thin(key, gold, got)
        got, want = np.asarray(got, dtype=np.float64), want.astype(np.float64)
        gkey = "grad/" + name
        if gkey in gold or gkey + "@thin" in gold:
            g = np.abs(gold[gkey] if gkey in gold else gold[gkey + "@thin"]).reshape(want.shape) * coef
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


# ---- three consecutive steps ((f)3: Adam bias correction, moment carry-over, BatchNorm running statistics) ----------------
MULTI_GRAD_RTOL = 3e-3      # steps 2 and 3 start from parameters that already differ from the reference's by rounding noise


def multistep_inputs(meta, i):
    """Batch and noise of step i of a multistep_* fixture (seed + 1000 i), checked against the stored checksum."""
    s_i = meta["seed"] + 1000 * i
    batch = cases.syn.make_batch(meta["sizes"], meta["obs_len"], meta["pred_len"], s_i, kind=meta["data_kind"], g_cells=tuple(meta["g_cells"]))
    N = batch[0].shape[1]
    T = meta["obs_len"] + meta["pred_len"] if meta["kind"] == "dagnet" else meta["obs_len"]
    eps = cases.syn.uniform_noise(s_i, "eps_train", (T - 1, N, meta["args"]["z_dim"]))
    want = meta["checks"]["steps"][i]
    assert abs(cases._sum(batch[0]) + cases._sum(eps) - want) <= 1e-9 * max(1.0, abs(want)), "input generator drifted"
    return batch, eps


def check_multistep(steps, final, gold, meta):
    """steps: per step dict(kld, nll, ce, loss, grads{name}, total_norm); final: dict(params{name}, buffers{name}, m{name}, v{name})."""
    lr, n = meta["lr"], meta["steps"]
    names = [k[len("after/"):] for k in gold if k.startswith("after/")]
    gtol = {}
    for i, out in enumerate(steps):
        for k in ("kld", "nll", "ce", "loss"):
            np.testing.assert_allclose(np.asarray(out[k]).reshape(-1), gold[f"step{i}/{k}"].reshape(-1), rtol=3e-4, atol=1e-5, err_msg=f"step {i} {k}")
        tn = float(gold[f"step{i}/total_norm"])
        np.testing.assert_allclose(float(out["total_norm"]), tn, rtol=3e-4, err_msg=f"step {i} total_norm")
        for name in names:
            key = f"step{i}/grad/{name}"
            if key not in gold:
                continue
            want = gold[key]
            tol = MULTI_GRAD_RTOL * np.abs(want).max() + cases.GRAD_ATOL + 1e-6 * tn
            gtol[(i, name)] = tol
            d = np.abs(np.asarray(out["grads"][name]) - want).max()
            assert d <= tol, f"{key}: max abs diff {d:.3e} > {tol:.3e}"
    for name in names:
        want = gold["after/" + name].astype(np.float64)
        got = final["params"].get(name)
        if got is None:
            got = final["buffers"][name]
        got = np.asarray(got, dtype=np.float64)
        if f"step0/grad/{name}" not in gold:                  # BatchNorm running statistics / num_batches_tracked
            # a bias in front of a train-mode BatchNorm (GCN layers) has an analytically zero gradient: Adam turns its rounding
            # noise into +-lr moves, the pre-BatchNorm mean follows that bias, and so does running_mean (not running_var)
            drift = 2.0 * lr * (n - 1) if name.endswith("running_mean") else 0.0
            assert (np.abs(got - want) <= 5e-5 + 3e-4 * np.abs(want) + drift).all(), f"after/{name}: {np.abs(got - want).max():.3e}"
            continue
        # Adam moves a weight by lr * m_hat / (sqrt(v_hat) + eps): a gradient error d changes one step's update by at most
        # 2 lr d / |g| (and never by more than 2 lr), so the bound per entry follows from the gradient tolerances above
        coefs = [min(1.0, meta["clip"] / (float(gold[f"step{i}/total_norm"]) + 1e-6)) for i in range(n)]
        gmin = np.min([np.abs(gold[f"step{i}/grad/{name}"]) * coefs[i] for i in range(n)], axis=0)
        tol = 2e-6 + 1e-5 * np.abs(want)
        for i in range(n):
            tol = tol + np.minimum(2.0 * lr, 2.0 * lr * gtol[(i, name)] * coefs[i] / np.maximum(gmin, 1e-30))
        assert (np.abs(got - want) <= tol).all(), f"after/{name}: max abs diff {np.abs(got - want).max():.3e}"
        m_want, v_want = gold["exp_avg/" + name], gold["exp_avg_sq/" + name]
        mt = sum(gtol[(i, name)] * coefs[i] for i in range(n))
        assert np.abs(np.asarray(final["m"][name]) - m_want).max() <= 0.1 * mt + 1e-9, f"exp_avg/{name}"
        gmax = max(np.abs(gold[f"step{i}/grad/{name}"]).max() * coefs[i] for i in range(n))
        assert np.abs(np.asarray(final["v"][name]) - v_want).max() <= 1e-3 * 2.0 * gmax * mt + 1e-12, f"exp_avg_sq/{name}"


def list_multistep_cases():
    return cases.list_cases(("multistep_",))


@pytest.mark.parametrize("name", list_multistep_cases())
def test_oracle_three_steps_match_reference(name):
    meta, gold = cases.load_golden(name)
    _, _, sd = cases.build_model(meta)
    a = meta["args"]
    cfg = orc.make_cfg(n_max_agents=meta["n_max_agents"], x_dim=2, **{k: v for k, v in a.items()})
    kind, state, steps = meta["kind"], None, []
    for i in range(meta["steps"]):
        (obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse), eps = multistep_inputs(meta, i)
        if kind == "vrnn":
            b = (obs_rel,)
        elif kind == "avrnn":
            b = (obs_rel, orc.adjacency_seq(cfg, sse, torch.cat((obs, gt), 0)))
        else:
            traj, rel = torch.cat((obs, gt), 0), torch.cat((obs_rel, gt_rel), 0)
            b = (traj, rel, orc.one_hot(torch.cat((obs_goals, gt_goals), 0).long(), cfg.g_dim), sse, orc.adjacency_seq(cfg, sse, traj))
        out = orc.train_step(sd, cfg, kind, b, eps, kld_weight=meta["kld_weights"][i], ce_weight=meta["ce_weight"], lr=meta["lr"],
                             clip=meta["clip"], adam_state=state)
        state = out["adam"]
        sd = dict(out["buffers"])
        sd.update(out["params"])
        steps.append(out)
    check_multistep(steps, dict(params=out["params"], buffers=out["buffers"], m=state["m"], v=state["v"]), gold, meta)
