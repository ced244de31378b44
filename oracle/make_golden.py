"""Generate tests/golden/*.npz by running the REFERENCE's own code (imported from /root/reference)
with injected noise.  Runs only in the build container; the GPU box never sees the reference.

    python oracle/make_golden.py            # writes tests/golden/

How noise is injected (the reference's algorithm is not changed):
  * torch.rand_like(x)            -> next pre-generated U[0,1) tensor of x's shape
  * torch.multinomial(p, 1)       -> argmax(p / q) with the next pre-generated Exp(1) tensor
    (bit-identical to what torch.multinomial does internally, SURVEY 8c')
``.cuda()`` is hard-coded throughout the reference, so Tensor.cuda / Module.cuda are made no-ops.

Inputs and weights are NOT stored: they are regenerated from seeds by dagnet_amd.synthetic in the
tests (a float64 checksum of each is stored to catch generator drift).  Stored: reference outputs.
"""
from __future__ import annotations

import json
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
sys.path.insert(0, "/root/reference")

from models.avrnn.model import AVRNN as RefAVRNN            # noqa: E402
from models.baseline.model import VRNN as RefVRNN            # noqa: E402
from models.dagnet.model import DAGNet as RefDAGNet          # noqa: E402
from models.dagnet.gat.gat_model import GAT as RefGAT        # noqa: E402
from models.dagnet.gcn.gcn_model import GCN as RefGCN        # noqa: E402
from models.dagnet.gat.gat_layers import GraphAttentionLayer as RefGatLayer     # noqa: E402
from models.dagnet.gcn.gcn_layers import GraphConvolution as RefGcnLayer        # noqa: E402
from models.utils import adjacency_matrix as ref_adj         # noqa: E402
from models.utils import utils as ref_utils                  # noqa: E402

from dagnet_amd import synthetic as syn                       # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class Injector:
    """Feeds pre-generated noise to torch.rand_like / torch.multinomial, in call order."""

    def __init__(self):
        self.eps, self.q = [], []
        self.margins, self.goals = [], []
        self._rand_like, self._multinomial = torch.rand_like, torch.multinomial

    def __enter__(self):
        def rand_like(x, *a, **k):
            e = self.eps.pop(0)
            assert e.shape == x.shape, (e.shape, x.shape)
            return e.clone()

        def multinomial(p, n, *a, **k):
            assert n == 1
            q = self.q.pop(0)
            ratio = p / q
            top2 = torch.topk(ratio, 2, dim=-1).values
            self.margins.append(((top2[:, 0] - top2[:, 1]) / top2[:, 0]).clone())
            idx = torch.argmax(ratio, dim=-1, keepdim=True)
            self.goals.append(idx[:, 0].clone())
            return idx

        torch.rand_like, torch.multinomial = rand_like, multinomial
        return self

    def __exit__(self, *a):
        torch.rand_like, torch.multinomial = self._rand_like, self._multinomial


def checksum(t):
    return float(np.asarray(t, dtype=np.float64).sum())


def build_case(name, kind, args_kw, sizes, obs_len, pred_len, seed, K, data_kind="bsk", g_cells=(9, 10)):
    args = syn.make_args(obs_len=obs_len, pred_len=pred_len, **args_kw)
    n_max = int(max(sizes))
    if kind == "vrnn":
        ref = RefVRNN(args)
    elif kind == "avrnn":
        ref = RefAVRNN(args)
    else:
        ref = RefDAGNet(args, n_max)
    sd = syn.random_state_dict(ref, seed)
    ref.load_state_dict(sd)
    ref.eval()
    batch = syn.make_batch(sizes, obs_len, pred_len, seed, kind=data_kind, g_cells=g_cells)
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = batch
    N = obs.shape[1]
    Z, G = args.z_dim, args.g_dim
    eps_f = syn.uniform_noise(seed, "eps_forward", (obs_len - 1, N, Z))
    eps_s = syn.uniform_noise(seed, "eps_sample", (K, pred_len, N, Z))
    q_s = syn.exp1_noise(seed, "q_sample", (K, pred_len, N, G)) if kind == "dagnet" else None

    out = {}
    # ---- teacher-forced pass over the observed steps
    with torch.no_grad(), Injector() as inj:
        inj.eps = [eps_f[t] for t in range(obs_len - 1)]
        if kind == "vrnn":
            kld, nll, h = ref(obs_rel)
            ce = torch.zeros(1)
        elif kind == "avrnn":
            adj = _ref_obs_adj(args, sse, obs, gt)
            kld, nll, h = ref(obs_rel, adj)
            ce = torch.zeros(1)
        else:
            adj = _ref_obs_adj(args, sse, obs, gt)
            goals_ohe = ref_utils.to_goals_one_hot(obs_goals, G)
            kld, nll, ce, h = ref(obs, obs_rel, goals_ohe, sse, adj)
        assert not inj.eps
    out.update(fwd_kld=kld.numpy(), fwd_nll=nll.numpy(), fwd_ce=ce.numpy(), fwd_h=h.numpy())
    # ---- K rollouts from h
    samples, goals, margins = [], [], []
    for k in range(K):
        with torch.no_grad(), Injector() as inj:
            inj.eps = [eps_s[k, t] for t in range(pred_len)]
            if kind == "vrnn":
                s = ref.sample(pred_len, h)
            elif kind == "avrnn":
                s = ref.sample(pred_len, h, obs[-1], sse)
            else:
                inj.q = [q_s[k, t] for t in range(pred_len)]
                s = ref.sample(pred_len, h, obs[-1], goals_ohe[-1], sse)
                goals.append(torch.stack(inj.goals).numpy())
                margins.append(torch.stack(inj.margins).numpy())
            assert not inj.eps and not inj.q
        samples.append(s.numpy())
    samples = np.stack(samples)
    out["samples"] = samples
    if goals:
        out["goals"] = np.stack(goals).astype(np.int32)
        out["margins"] = np.stack(margins).astype(np.float32)
    # ---- metrics exactly as evaluate() computes them
    ade, fde = [], []
    for k in range(K):
        a = ref_utils.relative_to_abs(torch.from_numpy(samples[k]), obs[-1])
        ade.append(ref_utils.average_displacement_error(a, gt, mode="raw"))
        fde.append(ref_utils.final_displacement_error(a[-1], gt[-1], mode="raw"))
    out["ade_raw"] = torch.stack(ade).numpy()
    out["fde_raw"] = torch.stack(fde).numpy()
    out["ade_sum"] = np.float32(_ref_evaluate_helper(ade, sse))
    out["fde_sum"] = np.float32(_ref_evaluate_helper(fde, sse))
    meta = dict(name=name, kind=kind, args=args_kw, sizes=[int(s) for s in sizes], obs_len=obs_len, pred_len=pred_len,
                seed=seed, K=K, data_kind=data_kind, g_cells=list(g_cells), n_max_agents=n_max,
                checks=dict(obs=checksum(obs), gt=checksum(gt), eps_f=checksum(eps_f), eps_s=checksum(eps_s),
                            q_s=checksum(q_s) if q_s is not None else 0.0,
                            weights=float(sum(checksum(v) for k_, v in sd.items() if v.dtype.is_floating_point))))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: N={N} samples{samples.shape} kld={float(kld):.5f} nll={float(nll):.5f} ce={float(ce):.5f} "
          f"ade_sum={float(out['ade_sum']):.4f}" + (f" min_margin={float(np.min(out['margins'])):.2e}" if goals else ""))


def train_case(name, kind, args_kw, sizes, obs_len, pred_len, seed, kld_weight=0.5, ce_weight=1e-2, lr=1e-3, clip=10.0,
               data_kind="bsk", g_cells=(9, 10), thin=False):
    """One training step exactly as the reference issues it (models/dagnet/train_bsk.py:259-291 and the avrnn /
    baseline analogues): model.train(), forward on the full sequence (DAG-Net) or the observed steps (others),
    loss = w_kld KLD + NLL (+ CE_weight CE), backward, clip_grad_norm_(params, clip), Adam step.
    Stored: losses, every parameter's .grad, the total norm, the parameters and BatchNorm buffers after the step."""
    args = syn.make_args(obs_len=obs_len, pred_len=pred_len, **args_kw)
    n_max = int(max(sizes))
    ref = RefVRNN(args) if kind == "vrnn" else RefAVRNN(args) if kind == "avrnn" else RefDAGNet(args, n_max)
    sd = syn.random_state_dict(ref, seed)
    ref.load_state_dict(sd)
    ref.train()
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = syn.make_batch(sizes, obs_len, pred_len, seed, kind=data_kind, g_cells=g_cells)
    N = obs.shape[1]
    T = obs_len + pred_len if kind == "dagnet" else obs_len
    eps = syn.uniform_noise(seed, "eps_train", (T - 1, N, args.z_dim))
    opt = torch.optim.Adam(ref.parameters(), lr=lr)
    opt.zero_grad()
    with Injector() as inj:
        inj.eps = [eps[t] for t in range(T - 1)]
        if kind == "vrnn":
            kld, nll, _ = ref(obs_rel)
            ce = torch.zeros(1)
            loss = kld_weight * kld + nll
        elif kind == "avrnn":
            kld, nll, _ = ref(obs_rel, _ref_obs_adj(args, sse, obs, gt))
            ce = torch.zeros(1)
            loss = kld_weight * kld + nll
        else:
            adj = _ref_obs_adj(args, sse, obs, gt)
            traj, rel = torch.cat((obs, gt), 0), torch.cat((obs_rel, gt_rel), 0)
            g_ohe = torch.cat((ref_utils.to_goals_one_hot(obs_goals, args.g_dim), ref_utils.to_goals_one_hot(gt_goals, args.g_dim)), 0)
            kld, nll, ce, _ = ref(traj, rel, g_ohe, sse, adj)
            loss = kld_weight * kld + nll + ce * ce_weight
        assert not inj.eps
    loss.backward()
    out = dict(kld=kld.detach().numpy(), nll=nll.detach().numpy(), ce=ce.detach().numpy(), loss=loss.detach().numpy())
    for k_, p_ in ref.named_parameters():
        out["grad/" + k_] = (p_.grad if p_.grad is not None else torch.zeros_like(p_)).numpy().copy()
    total_norm = torch.nn.utils.clip_grad_norm_(ref.parameters(), clip)
    out["total_norm"] = np.float32(total_norm)
    opt.step()
    for k_, v in ref.state_dict().items():
        out["after/" + k_] = v.detach().numpy().copy()
    meta = dict(name=name, kind=kind, args=args_kw, sizes=[int(s_) for s_ in sizes], obs_len=obs_len, pred_len=pred_len, seed=seed,
                data_kind=data_kind, g_cells=list(g_cells), n_max_agents=n_max, kld_weight=kld_weight, ce_weight=ce_weight, lr=lr,
                clip=clip, checks=dict(obs=checksum(obs), eps=checksum(eps),
                                       weights=float(sum(checksum(v) for v in sd.values() if v.dtype.is_floating_point))))
    if thin:        # a model with millions of parameters: of every large tensor keep each 61st element and the L2 norm
        for k_ in [k_ for k_, v in out.items() if isinstance(v, np.ndarray) and v.size > 65536]:
            v = out.pop(k_)
            out[k_ + "@thin"] = v.reshape(-1)[::61].copy()
            out[k_ + "@norm"] = np.float64(np.sqrt((v.astype(np.float64) ** 2).sum()))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: N={N} T={T} kld={float(kld):.5f} nll={float(nll):.5f} ce={float(ce):.5f} |g|={float(total_norm):.4f}")


def _ref_obs_adj(args, sse, obs, gt):
    if args.adjacency_type == 0:
        return ref_adj.compute_adjs(args, sse)
    if args.adjacency_type == 1:
        return ref_adj.compute_adjs_distsim(args, sse, obs, gt)
    return ref_adj.compute_adjs_knnsim(args, sse, obs, gt)


def _ref_evaluate_helper(error, sse):
    """The reference's own evaluate_helper (models/dagnet/evaluate_bsk.py:71-82)."""
    from models.dagnet.evaluate_bsk import evaluate_helper      # import only builds an argparse parser
    return float(evaluate_helper(list(error), sse))


def graph_case(name, kind, nin, nhid, nout, sizes, seed, nheads=4, alpha=0.2):
    ref = RefGAT(nin, nhid, nout, alpha, nheads) if kind == "gat" else RefGCN(nin, nhid, nout)
    sd = syn.random_state_dict(ref, seed)
    ref.load_state_dict(sd)
    ref.eval()
    n = int(sum(sizes))
    x = syn.uniform_noise(seed, "graph_x", (n, nin)) * 2 - 1
    pos = syn.uniform_noise(seed, "graph_pos", (n, 2)) * 10
    sse = syn.seq_start_end(sizes)
    outs = {}
    for at, fn in ((0, lambda: ref_adj.adjs_fully_connected_pred(sse)),
                   (1, lambda: ref_adj.adjs_distance_sim_pred(1.2, sse, pos)),
                   (2, lambda: ref_adj.adjs_knn_sim_pred(3, sse, pos))):
        adj = fn()
        assert adj.dtype == torch.float32
        with torch.no_grad():
            outs[f"adj{at}"] = adj.numpy()
            outs[f"out{at}"] = ref(x, adj).numpy()
    meta = dict(name=name, kind=kind, nin=nin, nhid=nhid, nout=nout, sizes=[int(s) for s in sizes], seed=seed,
                nheads=nheads, alpha=alpha, checks=dict(x=checksum(x), pos=checksum(pos)))
    outs["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **outs)
    print(f"{name}: out {outs['out1'].shape}")


def state_dict_layouts():
    """Key names and shapes of the reference modules' state dicts (the drop-in contract)."""
    out = {}
    for tag, kw in (("real_gat", dict(REAL, n_layers=2)), ("real_gcn", dict(REAL, n_layers=1, graph_model="gcn"))):
        a = syn.make_args(**kw)
        for name, mod in (("vrnn", RefVRNN(a)), ("avrnn", RefAVRNN(a)), ("dagnet", RefDAGNet(a, 10))):
            out[f"{name}_{tag}"] = {k: list(v.shape) for k, v in mod.state_dict().items()}
    with open(os.path.join(OUT, "state_dict_layouts.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("state_dict_layouts.json:", {k: len(v) for k, v in out.items()})


TINY = dict(h_dim=16, z_dim=8, rnn_dim=16, g_dim=12)
REAL = dict(h_dim=64, z_dim=32, rnn_dim=64, g_dim=90)


def train_cases():
    # --- one training step (row T): tiny dims for every model / graph kind, one real-dims DAG-Net step
    train_case("train_vrnn_tiny", "vrnn", dict(TINY, n_layers=2), [2, 5, 3], 4, 5, 60)
    train_case("train_avrnn_tiny_gat", "avrnn", dict(TINY, n_layers=1, graph_model="gat", adjacency_type=1), [2, 5, 3], 4, 5, 61)
    train_case("train_avrnn_tiny_gcn", "avrnn", dict(TINY, n_layers=2, graph_model="gcn", adjacency_type=2, top_k_neigh=3), [2, 5, 3], 4, 5, 62)
    train_case("train_dagnet_tiny_gat", "dagnet", dict(TINY, n_layers=2, graph_model="gat", adjacency_type=1), [2, 5, 3], 4, 5, 63,
               g_cells=(3, 4), data_kind="sdd")
    train_case("train_dagnet_tiny_gcn", "dagnet", dict(TINY, n_layers=1, graph_model="gcn", adjacency_type=0), [2, 5, 3], 4, 5, 64,
               g_cells=(3, 4), data_kind="sdd")
    train_case("train_dagnet_bsk_gat_n5", "dagnet", dict(REAL, n_layers=2), [5, 5, 5], 3, 4, 65)
    train_sdd_case()


def train_sdd_case():
    # --- one training step at the SDD shape: g_dim = 32 x 30 = 960, ragged scenes (models/dagnet/train_sdd.py:29-30,334)
    train_case("train_dagnet_sdd_g960", "dagnet", dict(REAL, g_dim=960, n_layers=2), [2, 7, 4], 3, 4, 66, data_kind="sdd", g_cells=(32, 30),
               thin=True)


def data_case():
    """Row L / 8(f) rank 2: the reference's collate functions and goal grid (models/dagnet/data/loader_bsk.py:12-35,
    loader_sdd.py:16-42, models/utils/utils.py:210-233) on seeded inputs.  Inputs are stored with the outputs (small)."""
    from models.dagnet.data import loader_bsk as ref_bsk
    from models.dagnet.data import loader_sdd as ref_sdd
    rng = np.random.default_rng(77)
    out = {}
    # goal grid: positions inside the rectangle, on grid lines, on the upper border (index = n_cells: the reference's quirk)
    lim = (-3.0, 57.0, 2.0, 62.0)
    tj = rng.uniform([lim[0], lim[2]], [lim[1], lim[3]], size=(6, 20, 2))
    tj[0, 3] = [lim[0], lim[2]]
    tj[1, 7] = [lim[1], lim[3]]
    tj[2, 5] = [lim[0] + 60.0 / 32 * 4, lim[2] + 60.0 / 30 * 7]
    out["goal_tj"] = tj
    out["goal_limits"] = np.asarray(lim)
    for w in (1, 3, 4, 7):
        out[f"goals_w{w}"] = np.stack([ref_utils.compute_goals_fixed(t, *lim, 32, 30, window=w) for t in tj])
    out["get_goal"] = np.asarray([ref_utils.get_goal(p, *lim, 32, 30) for p in tj[:, 0]])
    # basketball collate: 3 scenes of 5 agents
    To, Tp, n = 4, 6, 5
    items = [[torch.from_numpy(rng.normal(size=s)).float() for s in ((To, n, 2), (Tp, n, 2), (To, n, 2), (Tp, n, 2), (To, n), (Tp, n))]
             for _ in range(3)]
    for i, it in enumerate(items):
        for j, t in enumerate(it):
            out[f"bsk_in_{i}_{j}"] = t.numpy()
    for j, t in enumerate(ref_bsk.seq_collate(items)):
        out[f"bsk_out_{j}"] = t.numpy()
    # SDD collate: ragged scenes, per-agent-major items
    sizes = [2, 5, 3]
    items = [[torch.from_numpy(rng.normal(size=s)).float()
              for s in ((m, 2, To), (m, 2, Tp), (m, 2, To), (m, 2, Tp), (m, To), (m, Tp), (m,), (m, To + Tp))] for m in sizes]
    for i, it in enumerate(items):
        for j, t in enumerate(it):
            out[f"sdd_in_{i}_{j}"] = t.numpy()
    for j, t in enumerate(ref_sdd.seq_collate(items)):
        out[f"sdd_out_{j}"] = t.numpy()
    # BasketDataset: the arrays it derives (relative coordinates, obs / pred split) for 4 scenes of 5 agents, read from files
    import tempfile
    traj = rng.uniform(0, 45, size=(To + Tp, 4 * n, 2))
    goals = rng.integers(0, 90, size=(To + Tp, 4 * n)).astype(np.float64)
    out["ds_traj"], out["ds_goals"] = traj, goals
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "trajectories.npy"), traj)
        np.save(os.path.join(d, "goals.npy"), goals)
        ds = ref_bsk.BasketDataset(d, n_agents=n, obs_len=To, pred_len=Tp)
    for j, t in enumerate(ds[2]):
        out[f"ds_item2_{j}"] = t.numpy()
    out["ds_len"] = np.asarray(len(ds))
    np.savez_compressed(os.path.join(OUT, "data_utils.npz"), **out)
    print("data_utils.npz:", len(out), "arrays")


def sdd_tracks(seed=91):
    """Synthetic Stanford-Drone-style tracks for the window extraction: per scene rows (frame, pedestrian, x, y), frames 12
    apart, pedestrians entering and leaving at different frames, positions a random walk with more than 4 decimals, rows of a
    frame in shuffled pedestrian order.  -> (list of (scene_name, data), limits dictionary)"""
    rng = np.random.default_rng(seed)
    files, limits = [], {}
    for name, n_frames, spans in (("alpha_0", 30, [(0, 30), (2, 17), (5, 30), (0, 9), (11, 24), (3, 29), (20, 30)]),
                                  ("beta_1", 25, [(0, 25), (0, 14), (6, 25), (8, 19)])):
        rows = []
        for ped, (a, b) in enumerate(spans):
            xy = np.cumsum(rng.normal(scale=0.7, size=(b - a, 2)), axis=0) + rng.uniform(10, 40, size=2)
            for k in range(b - a):
                rows.append([12.0 * (a + k), 100.0 + 3 * ped, xy[k, 0], xy[k, 1]])
        rows = np.asarray(rows)
        rows = rows[np.lexsort((rng.permutation(len(rows)), rows[:, 0]))]
        files.append((name, rows))
        limits[name] = {"x_min": float(rows[:, 2].min() - 1), "x_max": float(rows[:, 2].max() + 1),
                        "y_min": float(rows[:, 3].min() - 1), "y_max": float(rows[:, 3].max() + 1)}
    return files, limits


def sdd_windows_case():
    """8(f) rank 2, the SDD window extraction (models/dagnet/data/loader_sdd.py:65-227) on synthetic tracks written in the
    reference's own file layout (<root>/sdd_npy/train.npy + <root>/limits.json); skip 1 and 3.  Inputs stored with the outputs."""
    import json
    import tempfile
    from models.dagnet.data import loader_sdd as ref_sdd
    files, limits = sdd_tracks()
    out = {"n_files": np.asarray(len(files))}
    for i, (name, rows) in enumerate(files):
        out[f"file{i}_rows"] = rows
        out[f"file{i}_limits"] = np.asarray(list(limits[name].values()))
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "sdd_npy"))
        with open(os.path.join(d, "limits.json"), "w") as fh:
            json.dump(limits, fh)
        arr = np.empty(len(files), dtype=object)
        for i, f in enumerate(files):
            arr[i] = f
        path = os.path.join(d, "sdd_npy", "train.npy")
        np.save(path, arr, allow_pickle=True)
        for skip in (1, 3):
            ds = ref_sdd.TrajectoryDataset(path, obs_len=4, pred_len=6, set="train", skip=skip, n_cells_x=5, n_cells_y=4,
                                           goals_window=3, disable=True)
            for k in ("obs_traj", "pred_traj", "obs_traj_rel", "pred_traj_rel", "obs_goals", "pred_goals", "loss_mask", "non_linear_ped"):
                out[f"skip{skip}_{k}"] = getattr(ds, k).numpy()
            out[f"skip{skip}_seq_start_end"] = np.asarray(ds.seq_start_end)
            out[f"skip{skip}_len"] = np.asarray([len(ds), ds.__max_agents__()])
            for j, t in enumerate(ds[3]):
                out[f"skip{skip}_item3_{j}"] = t.numpy()
    np.savez_compressed(os.path.join(OUT, "sdd_windows.npz"), **out)
    print("sdd_windows.npz:", len(out), "arrays;", int(out["skip1_len"][0]), "sequences at skip 1")


def layer_case():
    """Rows D1 / D3 / E3: the reference's stand-alone layers (gat_layers.py:25-40 with concat True / False,
    gcn_layers.py:20-26) on block-diagonal adjacencies of types 1 and 2, outputs plus the gradients of sum(out * probe)
    w.r.t. the input and the layer's parameters; and block_diag_irregular (models/utils/utils.py:290-302) on fp64 [n,n]
    and [T,n,n] blocks.  -> tests/golden/layers.npz"""
    sizes, seed = [2, 5, 3, 10], 70
    n = int(sum(sizes))
    sse = syn.seq_start_end(sizes)
    pos = syn.uniform_noise(seed, "layer_pos", (n, 2)) * 10
    adjs = {1: ref_adj.adjs_distance_sim_pred(1.2, sse, pos), 2: ref_adj.adjs_knn_sim_pred(3, sse, pos)}
    out = {}
    specs = (("gat_concat", lambda: RefGatLayer(64, 8, 0.2, concat=True), 64, 8),
             ("gat_out", lambda: RefGatLayer(32, 64, 0.2, concat=False), 32, 64),
             ("gat_wide", lambda: RefGatLayer(32, 90, 0.2, concat=False), 32, 90),
             ("gcn", lambda: RefGcnLayer(64, 8), 64, 8),
             ("gcn_nobias", lambda: RefGcnLayer(8, 64, bias=False), 8, 64))
    checks = {}
    for tag, mk, nin, nout in specs:
        ref = mk()
        ref.load_state_dict(syn.random_state_dict(ref, seed))
        x0 = syn.uniform_noise(seed, "layer_x_" + tag, (n, nin)) * 2 - 1
        probe = syn.uniform_noise(seed, "layer_probe_" + tag, (n, nout)) * 2 - 1
        checks[tag] = checksum(x0) + checksum(probe)
        for at, adj in adjs.items():
            x = x0.clone().requires_grad_(True)
            ref.zero_grad()
            y = ref(x, adj)
            (y * probe).sum().backward()
            out[f"{tag}_adj{at}_out"] = y.detach().numpy()
            out[f"{tag}_adj{at}_dx"] = x.grad.numpy().copy()
            for k_, p_ in ref.named_parameters():
                out[f"{tag}_adj{at}_d{k_}"] = p_.grad.numpy().copy()
    # block_diag_irregular: fp64 blocks in, fp32 out (implicit cast on assignment into torch.zeros)
    rng = np.random.default_rng(71)
    b2 = [rng.normal(size=(m, m)) for m in (1, 4, 2, 7)]
    b3 = [rng.normal(size=(3, m, m)) for m in (2, 5, 3)]
    for i, b in enumerate(b2):
        out[f"bd2_in_{i}"] = b
    for i, b in enumerate(b3):
        out[f"bd3_in_{i}"] = b
    o2 = ref_utils.block_diag_irregular([torch.from_numpy(b) for b in b2])
    o3 = ref_utils.block_diag_irregular([torch.from_numpy(b) for b in b3])
    assert o2.dtype == torch.float32 and o3.dtype == torch.float32
    out["bd2_out"], out["bd3_out"] = o2.numpy(), o3.numpy()
    meta = dict(sizes=sizes, seed=seed, checks=dict(pos=checksum(pos), **checks))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "layers.npz"), **out)
    print("layers.npz:", len(out), "arrays; bd2", o2.shape, "bd3", o3.shape)


MULTI_STEPS = 3
MULTI_KLD = (0.0, 0.3, 1.0)      # three different KLD warm-up weights (train_bsk.py:377-380 ramps the weight per epoch)


def multistep_case(name, kind, args_kw, sizes, obs_len, pred_len, seed, ce_weight=1e-2, lr=1e-3, clip=10.0, data_kind="bsk",
                   g_cells=(9, 10)):
    """(f)3: three consecutive optimizer steps of the reference loop body (train_bsk.py:286-291) on three different
    batches with three KLD weights: Adam's bias correction, moment carry-over and the BatchNorm running statistics over
    several steps.  Step i uses the batch / noise of seed + 1000 i.  Stored: losses and total norm per step, every
    gradient per step, and after the last step the state dict, exp_avg and exp_avg_sq."""
    args = syn.make_args(obs_len=obs_len, pred_len=pred_len, **args_kw)
    n_max = int(max(sizes))
    ref = RefVRNN(args) if kind == "vrnn" else RefAVRNN(args) if kind == "avrnn" else RefDAGNet(args, n_max)
    sd = syn.random_state_dict(ref, seed)
    ref.load_state_dict(sd)
    ref.train()
    opt = torch.optim.Adam(ref.parameters(), lr=lr)
    out, checks = {}, []
    for i in range(MULTI_STEPS):
        s_i = seed + 1000 * i
        obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = syn.make_batch(sizes, obs_len, pred_len, s_i, kind=data_kind, g_cells=g_cells)
        N = obs.shape[1]
        T = obs_len + pred_len if kind == "dagnet" else obs_len
        eps = syn.uniform_noise(s_i, "eps_train", (T - 1, N, args.z_dim))
        checks.append(checksum(obs) + checksum(eps))
        opt.zero_grad()
        with Injector() as inj:
            inj.eps = [eps[t] for t in range(T - 1)]
            if kind == "vrnn":
                kld, nll, _ = ref(obs_rel)
                ce = torch.zeros(1)
                loss = MULTI_KLD[i] * kld + nll
            elif kind == "avrnn":
                kld, nll, _ = ref(obs_rel, _ref_obs_adj(args, sse, obs, gt))
                ce = torch.zeros(1)
                loss = MULTI_KLD[i] * kld + nll
            else:
                adj = _ref_obs_adj(args, sse, obs, gt)
                traj, rel = torch.cat((obs, gt), 0), torch.cat((obs_rel, gt_rel), 0)
                g_ohe = torch.cat((ref_utils.to_goals_one_hot(obs_goals, args.g_dim), ref_utils.to_goals_one_hot(gt_goals, args.g_dim)), 0)
                kld, nll, ce, _ = ref(traj, rel, g_ohe, sse, adj)
                loss = MULTI_KLD[i] * kld + nll + ce * ce_weight
            assert not inj.eps
        loss.backward()
        for k_, v in (("kld", kld), ("nll", nll), ("ce", ce), ("loss", loss)):
            out[f"step{i}/{k_}"] = v.detach().numpy().copy()
        for k_, p_ in ref.named_parameters():
            out[f"step{i}/grad/{k_}"] = (p_.grad if p_.grad is not None else torch.zeros_like(p_)).numpy().copy()
        out[f"step{i}/total_norm"] = np.float32(torch.nn.utils.clip_grad_norm_(ref.parameters(), clip))
        opt.step()
    for k_, v in ref.state_dict().items():
        out["after/" + k_] = v.detach().numpy().copy()
    names = [k_ for k_, _ in ref.named_parameters()]
    st = opt.state_dict()["state"]
    for i, k_ in enumerate(names):
        out["exp_avg/" + k_] = st[i]["exp_avg"].numpy().copy()
        out["exp_avg_sq/" + k_] = st[i]["exp_avg_sq"].numpy().copy()
        assert int(st[i]["step"]) == MULTI_STEPS
    meta = dict(name=name, kind=kind, args=args_kw, sizes=[int(s_) for s_ in sizes], obs_len=obs_len, pred_len=pred_len, seed=seed,
                data_kind=data_kind, g_cells=list(g_cells), n_max_agents=n_max, kld_weights=list(MULTI_KLD), ce_weight=ce_weight, lr=lr,
                clip=clip, steps=MULTI_STEPS, checks=dict(steps=checks,
                                                         weights=float(sum(checksum(v) for v in sd.values() if v.dtype.is_floating_point))))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: {MULTI_STEPS} steps, losses", [float(out[f'step{i}/loss']) for i in range(MULTI_STEPS)],
          "norms", [float(out[f'step{i}/total_norm']) for i in range(MULTI_STEPS)])


def multistep_cases():
    # lr = 2e-4: parameters whose gradient is pure rounding noise (the out head's attention vector, a bias in front of a
    # train-mode BatchNorm) still move by +-lr per Adam step, in a direction the noise decides; a small lr keeps the effect
    # of that wandering on the later steps' gradients below the comparison tolerance
    multistep_case("multistep_dagnet_tiny_gat", "dagnet", dict(TINY, n_layers=2, graph_model="gat", adjacency_type=1), [2, 5, 3], 4, 5, 163,
                   g_cells=(3, 4), data_kind="sdd", lr=2e-4)
    multistep_case("multistep_avrnn_tiny_gcn", "avrnn", dict(TINY, n_layers=2, graph_model="gcn", adjacency_type=2, top_k_neigh=3), [2, 5, 3], 4, 5, 162,
                   lr=2e-4)


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "train":
        return train_cases()
    if len(sys.argv) > 1 and sys.argv[1] == "train_sdd":
        return train_sdd_case()
    if len(sys.argv) > 1 and sys.argv[1] == "layers":
        return layer_case()
    if len(sys.argv) > 1 and sys.argv[1] == "train3":
        return multistep_cases()
    if len(sys.argv) > 1 and sys.argv[1] == "data":
        return data_case()
    if len(sys.argv) > 1 and sys.argv[1] == "sdd_windows":
        return sdd_windows_case()
    data_case()
    sdd_windows_case()
    state_dict_layouts()
    train_cases()
    layer_case()
    multistep_cases()
    # --- tiny dims, ragged scenes, every graph x adjacency combination
    for L in (1, 2):
        build_case(f"vrnn_tiny_L{L}", "vrnn", dict(TINY, n_layers=L), [2, 5, 3], 4, 5, 10 + L, 2)
    for gm in ("gat", "gcn"):
        for at in (0, 1, 2):
            kw = dict(TINY, n_layers=2 if at != 2 else 1, graph_model=gm, adjacency_type=at, top_k_neigh=3 if at == 2 else None)
            build_case(f"avrnn_tiny_{gm}_adj{at}", "avrnn", kw, [2, 5, 3], 4, 5, 20 + at, 2)
            build_case(f"dagnet_tiny_{gm}_adj{at}", "dagnet", kw, [2, 5, 3], 4, 5, 30 + at, 2, g_cells=(3, 4), data_kind="sdd")
    # --- real basketball dims (5 and 10 agents per scene), obs 10 / pred 40
    build_case("vrnn_bsk", "vrnn", dict(REAL, n_layers=2), [10, 10], 10, 40, 41, 2)
    build_case("avrnn_bsk_gat", "avrnn", dict(REAL, n_layers=1), [5, 5, 5], 10, 40, 42, 2)
    build_case("avrnn_bsk_gat_L2_n10", "avrnn", dict(REAL, n_layers=2), [10, 10], 10, 40, 43, 2)
    build_case("dagnet_bsk_gat_n5", "dagnet", dict(REAL, n_layers=2), [5, 5, 5], 10, 40, 44, 2)
    build_case("dagnet_bsk_gat_n10", "dagnet", dict(REAL, n_layers=2), [10, 10, 10], 10, 40, 45, 3)
    build_case("dagnet_bsk_gcn_n10", "dagnet", dict(REAL, n_layers=2, graph_model="gcn"), [10, 10], 10, 40, 46, 2)
    # --- SDD-shaped: obs 8 / pred 12, ragged, z=16 (shipped A-VRNN/VRNN SDD checkpoints)
    build_case("avrnn_sdd_gat", "avrnn", dict(REAL, z_dim=16, n_layers=1), [2, 7, 4, 12], 8, 12, 47, 2, data_kind="sdd")
    build_case("vrnn_sdd", "vrnn", dict(REAL, z_dim=16, n_layers=2), [2, 7, 4], 8, 12, 48, 2, data_kind="sdd")
    # --- SDD-shaped DAG-Net: g_dim = 32 x 30 = 960 (models/dagnet/train_sdd.py:29-30,334), ragged scenes
    build_case("dagnet_sdd_gat_g960", "dagnet", dict(REAL, g_dim=960, n_layers=2), [2, 7, 4], 8, 12, 55, 2, data_kind="sdd", g_cells=(32, 30))
    build_case("dagnet_sdd_gcn_g960", "dagnet", dict(REAL, g_dim=960, n_layers=1, graph_model="gcn", adjacency_type=2, top_k_neigh=3), [3, 9], 8, 12, 56, 1,
               data_kind="sdd", g_cells=(32, 30))
    # --- edge cases: a one-agent scene, a 32-agent scene (largest register-resident tile), a 33-agent scene
    build_case("dagnet_bsk_gat_ragged_1_32", "dagnet", dict(REAL, n_layers=2), [1, 32, 7], 4, 6, 49, 2, data_kind="sdd", g_cells=(9, 10))
    build_case("avrnn_bsk_gat_n33", "avrnn", dict(REAL, n_layers=1), [33, 2], 4, 5, 50, 1, data_kind="sdd")
    # --- stand-alone graph modules
    graph_case("gat_hid", "gat", 64, 8, 64, [2, 5, 3, 10], 51)
    graph_case("gat_goal", "gat", 90, 8, 90, [3, 10], 52)
    graph_case("gcn_hid", "gcn", 64, 8, 64, [2, 5, 3, 10], 53)
    graph_case("gat_tiny", "gat", 12, 4, 20, [1, 4, 2], 54, nheads=2)


if __name__ == "__main__":
    main()
