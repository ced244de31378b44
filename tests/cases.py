"""Rebuild the inputs of a golden case from its seeds (weights, batch, injected noise)."""
from __future__ import annotations

import glob
import json
import os

import numpy as np
import torch

from dagnet_amd import synthetic as syn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def list_cases(prefixes=("vrnn", "avrnn", "dagnet")):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
    return [n for n in names if n.startswith(tuple(prefixes))]


def list_graph_cases():
    return list_cases(("gat_", "gcn_"))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return meta, {k: z[k] for k in z.files if k != "meta"}


def _sum(t):
    return float(np.asarray(t, dtype=np.float64).sum())


def build_model(meta):
    import dagnet_amd
    args = syn.make_args(obs_len=meta["obs_len"], pred_len=meta["pred_len"], **meta["args"])
    if meta["kind"] == "vrnn":
        m = dagnet_amd.VRNN(args)
    elif meta["kind"] == "avrnn":
        m = dagnet_amd.AVRNN(args)
    else:
        m = dagnet_amd.DAGNet(args, meta["n_max_agents"])
    sd = syn.random_state_dict(m, meta["seed"])
    m.load_state_dict(sd)
    m.eval()
    w = float(sum(_sum(v) for v in sd.values() if v.dtype.is_floating_point))
    assert abs(w - meta["checks"]["weights"]) <= 1e-6 * max(1.0, abs(w)), "weight generator drifted"
    return args, m, sd


def build_inputs(meta):
    seed, K = meta["seed"], meta["K"]
    batch = syn.make_batch(meta["sizes"], meta["obs_len"], meta["pred_len"], seed, kind=meta["data_kind"],
                           g_cells=tuple(meta["g_cells"]))
    N = batch[0].shape[1]
    Z, G = meta["args"]["z_dim"], meta["args"]["g_dim"]
    eps_f = syn.uniform_noise(seed, "eps_forward", (meta["obs_len"] - 1, N, Z))
    eps_s = syn.uniform_noise(seed, "eps_sample", (K, meta["pred_len"], N, Z))
    q_s = syn.exp1_noise(seed, "q_sample", (K, meta["pred_len"], N, G)) if meta["kind"] == "dagnet" else None
    c = meta["checks"]
    for got, want in ((_sum(batch[0]), c["obs"]), (_sum(batch[1]), c["gt"]), (_sum(eps_f), c["eps_f"]),
                      (_sum(eps_s), c["eps_s"]), (_sum(q_s) if q_s is not None else 0.0, c["q_s"])):
        assert abs(got - want) <= 1e-9 * max(1.0, abs(want)), "input generator drifted from the golden fixture"
    return batch, eps_f, eps_s, q_s


def graph_inputs(meta):
    n = int(sum(meta["sizes"]))
    x = syn.uniform_noise(meta["seed"], "graph_x", (n, meta["nin"])) * 2 - 1
    pos = syn.uniform_noise(meta["seed"], "graph_pos", (n, 2)) * 10
    assert abs(_sum(x) - meta["checks"]["x"]) < 1e-6 and abs(_sum(pos) - meta["checks"]["pos"]) < 1e-6
    return x, pos, syn.seq_start_end(meta["sizes"])


# tolerances (SURVEY 7 "Hard parts", calibrated against the reference-vs-reference floor of ~2e-6):
DISP_ATOL, DISP_RTOL = 1e-4, 1e-4        # per-step displacements
H_ATOL = 2e-4                             # hidden state after the observed steps
LOSS_RTOL = 1e-4                          # KLD / NLL / CE, ADE / FDE sums
MARGIN_MIN = 1e-4                         # rows whose top-2 goal margin is below this are excluded


# ---- training-step fixtures (row T) ---------------------------------------------------------------
def list_train_cases():
    return list_cases(("train_",))


# per-parameter gradient: |got - want| <= GRAD_RTOL * max|want| + GRAD_ATOL (+ 1e-6 of the total norm).
# Calibration: on train_dagnet_bsk_gat_n5 the reference's own fp32 gradients sit 1.5e-4 .. 4e-4 (relative to max|g|) away from
# an fp64 evaluation of the same step (train-mode BatchNorm over 15 rows is ill-conditioned), so 1e-3 is ~3x the fixture's
# own rounding noise; a wrong formula (slope, gate order, a missing 1/N) shows up at >= 1e-2.
GRAD_RTOL = 1e-3
GRAD_ATOL = 2e-6


def build_train_inputs(meta):
    """-> (batch tuple as seq_collate emits it, eps [T-1,N,Z]); T = full sequence for DAG-Net, observed steps otherwise."""
    seed = meta["seed"]
    batch = syn.make_batch(meta["sizes"], meta["obs_len"], meta["pred_len"], seed, kind=meta["data_kind"], g_cells=tuple(meta["g_cells"]))
    N = batch[0].shape[1]
    T = meta["obs_len"] + meta["pred_len"] if meta["kind"] == "dagnet" else meta["obs_len"]
    eps = syn.uniform_noise(seed, "eps_train", (T - 1, N, meta["args"]["z_dim"]))
    c = meta["checks"]
    assert abs(_sum(batch[0]) - c["obs"]) <= 1e-9 * max(1.0, abs(c["obs"])) and abs(_sum(eps) - c["eps"]) <= 1e-9 * max(1.0, abs(c["eps"]))
    return batch, eps
