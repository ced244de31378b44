#!/usr/bin/env python
"""Benchmark of the hot path: one best-of-K rollout (K sample() rollouts + min-over-K ADE/FDE) per step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

Metric (BASELINE.json): sampled trajectories / second = scenes x agents x K / t_rollout, inputs
(h after the observed steps, last positions, last goals, GT future) already resident in HBM.
N > 1: one process per GPU (torch.distributed / RCCL), scenes sharded, one all-reduce of (ade_sum, fde_sum, n_agents) per
step; value = all ranks' trajectories / max time over ranks.  Either the driver starts the ranks (torch.distributed.run:
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or, when --gpus N > 1 is given WITHOUT WORLD_SIZE, this
script starts N child processes itself - before anything in the parent touches the GPU; the parent only waits and relays
rank 0's JSON line.  Workloads are weak scaling (fixed scenes per rank) except dagnet_bsk_1Mx20 (BASELINE configs[4]:
1 048 576 scenes in total, sharded over the ranks, walked in chunks of 4096 scenes: strong scaling).
--rehearse runs the launcher, the process group (gloo, CPU) and the reductions with no GPU work (a CPU test uses it).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import dagnet_amd                                    # noqa: E402
from dagnet_amd import _capi, synthetic as syn       # noqa: E402
from dagnet_amd._runtime import scene_ptr_from_sse   # noqa: E402

PEAK_FP32_MATRIX_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
# Split products on the 16-bit matrix pipe, fp32 in / fp32 out / fp32 accumulation.  Their bound in fp32-equivalent
# FLOP/s is the dense 16-bit peak (2.5 PFLOP/s) over the 16-bit products one fp32 product costs:
#   register-resident rollout kernels (csrc/v2_common.h mm<..., BF>): two fp16 terms per value, three products (hh, hl, lh);
#   wide layers of the g_dim = 960 goal head (csrc/gemm_f32.hip, SPLIT = 2): the same since round 3; three bf16 terms per value and
#   six products (SPLIT = 3) in dagnet_op_gemm's wide layers and under DAGNET_WIDE_BF16X3.
PEAK_SPLIT_FP16X2_TFLOPS = 2500.0 / 3.0
PEAK_SPLIT_BF16X3_TFLOPS = 2500.0 / 6.0


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*_pmc.json,
    written by tools/profile_round.sh + tools/pmc_summary.py: 2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction applied).
    None when no counter pass of this workload/kernel has been committed."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
        try:
            ks = json.load(open(path)).get(workload, {}).get("kernels", {})
        except (OSError, ValueError):
            continue
        # a section that lists a rollout kernel of ANOTHER dispatch path was recorded before the path changed: stale
        family = {"rollout_v2_kernel": ("rollout_v2_kernel",),
                  "gemm_nt_kernel": ("gemm_nt_kernel", "gemm_nt_pipe_kernel", "rollout_v2_kernel", "wide::"),
                  "dagnet_rollout_big_kernel": ("dagnet_rollout_big_kernel",), "dagnet_rollout_kernel": ("dagnet_rollout_kernel",)}
        ok = family.get(kernel.split(" ")[0], ())
        if not ks or any(not any(f in name for f in ok) for name in ks):
            continue
        stem = kernel.split(" ")[0]
        stems = (stem + "<", "gemm_nt_pipe_kernel<") if stem == "gemm_nt_kernel" else (stem + "<",)
        hits = [float(v["traffic_bytes_per_launch"]) for name, v in ks.items()
                if any(x in name for x in stems) and "traffic_bytes_per_launch" in v]
        if hits:
            return max(hits)        # several instantiations of the kernel: the dominant (largest) one
    return None

WORKLOADS = {
    # BASELINE.json configs[2]: DAG-Net full, basketball-shaped, 4096 scenes x 20 samples, 1 GPU
    "dagnet_bsk_4096x20": dict(kind="dagnet", scenes=4096, agents=10, K=20, obs=10, pred=40, args=dict(n_layers=2)),
    # configs[1]: A-VRNN (single GAT on hidden states), 512 scenes
    "avrnn_bsk_512x20": dict(kind="avrnn", scenes=512, agents=10, K=20, obs=10, pred=40, args=dict(n_layers=1)),
    # the same model on 4096 scenes: what the kernel reaches when the tile count does not quantise
    "avrnn_bsk_4096x20": dict(kind="avrnn", scenes=4096, agents=10, K=20, obs=10, pred=40, args=dict(n_layers=1)),
    # configs[3]: DAG-Net SDD-shaped: obs 8 / pred 12, ragged scenes (2..32 agents, mean 6), g_dim = 32 x 30 = 960
    "dagnet_sdd_1024x20": dict(kind="dagnet", scenes=1024, agents=0, K=20, obs=8, pred=12, args=dict(n_layers=2, g_dim=960),
                               data="sdd", g_cells=(32, 30)),
    # configs[3] at the size SURVEY 8(d) states (cfg4: S = 4096 ragged scenes, K = 20)
    "dagnet_sdd_4096x20": dict(kind="dagnet", scenes=4096, agents=0, K=20, obs=8, pred=12, args=dict(n_layers=2, g_dim=960),
                               data="sdd", g_cells=(32, 30)),
    # configs[4], evaluation leg: 1 048 576 scenes x 20 samples in total, sharded over the ranks (strong scaling), every rank
    # walks its share in chunks of 4096 scenes (one sample_k launch + one metric launch per chunk, sums added on the device)
    "dagnet_bsk_1Mx20": dict(kind="dagnet", scenes=0, total_scenes=1 << 20, chunk=4096, agents=10, K=20, obs=10, pred=40,
                             args=dict(n_layers=2)),
    # configs[0] shape on the GPU: VRNN baseline
    "vrnn_bsk_4096x20": dict(kind="vrnn", scenes=4096, agents=10, K=20, obs=10, pred=40, args=dict(n_layers=2)),
    # configs[4], training leg (row T): one optimizer step of DAG-Net on a chunk of scenes, full 50-step sequence
    "train_dagnet_bsk_1024": dict(kind="dagnet", scenes=1024, agents=10, K=1, obs=10, pred=40, args=dict(n_layers=2), train=True),
    "train_dagnet_bsk_8192": dict(kind="dagnet", scenes=8192, agents=10, K=1, obs=10, pred=40, args=dict(n_layers=2), train=True),
    # twice the rows per optimizer step: the step's kernels are latency-bound chains, larger ones amortise better (823 k against
    # 715 k training trajectories/s); 32768 scenes no longer fit the 288 GB (the stacked tape of 16384 takes about half of it)
    "train_dagnet_bsk_16384": dict(kind="dagnet", scenes=16384, agents=10, K=1, obs=10, pred=40, args=dict(n_layers=2), train=True),
    # the same step at the SDD shape (models/dagnet/train_sdd.py:29-30,334): g_dim = 960, ragged scenes, obs 8 / pred 12
    "train_dagnet_sdd_4096": dict(kind="dagnet", scenes=4096, agents=0, K=1, obs=8, pred=12, args=dict(n_layers=2, g_dim=960),
                                  data="sdd", g_cells=(32, 30), train=True),
}


def flops_per_row_step(kind, a, n, d_dim):
    """Algorithmic MACs per row and timestep of sample(); 1 MAC = 2 FLOP.
    One-hot inputs are gathers (0 MAC), the dead dec_logvar of sample() is excluded."""
    X, H, Z, R, L, G, F, P = a.x_dim, a.h_dim, a.z_dim, a.rnn_dim, a.n_layers, a.g_dim, a.graph_hid, a.n_heads

    def gat(nin, nout):
        return nin * F * P + 2 * F * P + n * F * P + F * P * nout + 2 * nout + n * nout
    prior = R * H + H * H + 2 * H * Z
    phi_z = Z * H + H * H
    dec = ((d_dim + H + R) if kind == "dagnet" else (H + R)) * H + H * H + H * X
    phi_x = X * H + H * H
    gru = 3 * R * 2 * H + 3 * R * R + (L - 1) * 6 * R * R
    mac = prior + phi_z + dec + phi_x + gru
    if kind != "vrnn":
        mac += gat(R, R) + 2 * R * R
    if kind == "dagnet":
        mac += (d_dim + R) * H + H * G + gat(G, G) + 2 * G * G
    return 2 * mac


def scene_sizes(wl, seed=0):
    if wl["agents"]:
        return [wl["agents"]] * wl["scenes"]
    return [int(v) for v in syn.ragged_sizes(wl["scenes"], seed=seed)]


def build(wl, device, seed=0):
    a = syn.make_args(obs_len=wl["obs"], pred_len=wl["pred"], **wl["args"])
    if wl["kind"] == "dagnet":
        model = dagnet_amd.DAGNet(a, wl["agents"] or 32)
    elif wl["kind"] == "avrnn":
        model = dagnet_amd.AVRNN(a)
    else:
        model = dagnet_amd.VRNN(a)
    model.load_state_dict(syn.random_state_dict(model, 128))
    return a, model.to(device).eval()


def cpu_baseline(wl, budget_s=20.0):
    """The oracle (CPU restatement with the reference's dense graph formulation) on a bounded sample:
    32 scenes (the reference's eval batch, models/dagnet/evaluate_bsk.py:32), as many of the K samples
    as fit the time budget."""
    from oracle import dagnet_oracle as orc
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    cores = torch.get_num_threads()
    scenes, n, T = 32, wl["agents"] or 6, wl["pred"]
    a = syn.make_args(obs_len=wl["obs"], pred_len=T, **wl["args"])
    kind = wl["kind"]
    ref_mod = {"dagnet": lambda: dagnet_amd.DAGNet(a, wl["agents"] or 32), "avrnn": lambda: dagnet_amd.AVRNN(a), "vrnn": lambda: dagnet_amd.VRNN(a)}[kind]()
    sd = syn.random_state_dict(ref_mod, 128)
    cfg = orc.make_cfg(n_max_agents=wl["agents"] or 32, **{k: getattr(a, k) for k in ("x_dim", "h_dim", "z_dim", "rnn_dim", "n_layers", "g_dim",
                                                                      "graph_model", "graph_hid", "adjacency_type",
                                                                      "top_k_neigh", "sigma", "alpha", "n_heads")})
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = syn.make_batch([n] * scenes, wl["obs"], T, 1, kind=wl.get("data", "bsk"),
                                                                        g_cells=wl.get("g_cells", (9, 10)))
    N = obs.shape[1]
    h = syn.uniform_noise(1, "h0", (a.n_layers, N, a.rnn_dim)) - 0.5

    def one(k):
        eps = syn.uniform_noise(k, "eps", (T, N, a.z_dim))
        with torch.no_grad():
            if kind == "vrnn":
                orc.vrnn_sample(sd, cfg, T, h, eps)
            elif kind == "avrnn":
                orc.avrnn_sample(sd, cfg, T, h, obs[-1], sse, eps)
            else:
                q = syn.exp1_noise(k, "q", (T, N, a.g_dim))
                orc.dagnet_sample(sd, cfg, T, h, obs[-1], orc.one_hot(obs_goals[-1], a.g_dim), sse, eps, q)
    one(0)   # warm-up
    t0 = time.perf_counter()
    done = 0
    while done < wl["K"] and (time.perf_counter() - t0) < budget_s:
        one(done + 1)
        done += 1
    dt = time.perf_counter() - t0
    return dict(value=N * done / dt, unit="trajectories/s", cores=cores, kind="port",
                sample=f"{scenes} scenes x {n} agents x {done} samples x {T} steps, dense-graph oracle, torch CPU fp32")


def cpu_baseline_stock(wl, kind=None, scenes=32, budget_s=10.0):
    """The reference's sample() call pattern on stock torch.nn layers (oracle/stock_torch.py: dense [N,N,2F] attention over
    the whole flattened batch, per-scene scipy adjacency, nn.GRU per step) - what a user of the reference pays on the host
    cores today.  32 scenes = the reference's eval batch (models/dagnet/evaluate_bsk.py:32).  Pinned against the reference's
    own outputs by tests/test_stock_baseline.py."""
    from oracle.stock_torch import StockRollout
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    kind = kind or wl["kind"]
    n, T = wl["agents"] or 6, wl["pred"]
    a = syn.make_args(obs_len=wl["obs"], pred_len=T, **{k: v for k, v in wl["args"].items() if kind == "dagnet" or k != "g_dim"})
    n_max = wl["agents"] or 32
    m = StockRollout(kind, a, n_max)
    ours = {"dagnet": lambda: dagnet_amd.DAGNet(a, n_max), "avrnn": lambda: dagnet_amd.AVRNN(a), "vrnn": lambda: dagnet_amd.VRNN(a)}[kind]()
    m.load_state_dict(syn.random_state_dict(ours, 128))
    m.eval()
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = syn.make_batch([n] * scenes, wl["obs"], T, 1, kind=wl.get("data", "bsk"),
                                                                        g_cells=wl.get("g_cells", (9, 10)))
    N = obs.shape[1]
    h = syn.uniform_noise(1, "h0", (a.n_layers, N, a.rnn_dim)) - 0.5
    g0 = torch.nn.functional.one_hot(obs_goals[-1].long(), a.g_dim).float() if kind == "dagnet" else None
    torch.manual_seed(0)
    t0 = time.perf_counter()
    done = 0
    while done < wl["K"] and (done == 0 or time.perf_counter() - t0 < budget_s):
        m.sample(T, h.clone(), obs[-1] if kind != "vrnn" else None, g0, sse)
        done += 1
    dt = time.perf_counter() - t0
    return dict(value=N * done / dt, unit="trajectories/s", cores=torch.get_num_threads(), kind="torch.nn dense-GAT",
                sample=f"{kind}: {scenes} scenes x {n} agents x {done} samples x {T} steps, stock torch.nn layers with the reference's dense "
                       "[N,N,2F] graph attention and per-scene scipy adjacency (oracle/stock_torch.py), torch CPU fp32")


def train_flops_per_row_step(a, n, d_dim):
    """Algorithmic FLOPs of one teacher-forced row-step of DAG-Net, forward + backward (backward = 2x the forward MACs:
    input and weight gradients).  Forward = the sample() count of flops_per_row_step plus the encoder and dec_logvar."""
    X, H, Z, R, G = a.x_dim, a.h_dim, a.z_dim, a.rnn_dim, a.g_dim
    fwd = flops_per_row_step("dagnet", a, n, d_dim) + 2 * ((H + R) * H + H * H + 2 * H * Z + H * X)
    return 3 * fwd


def bench_train(opt, wl, rank, world, dev, dist):
    """One training step = zero_grad, train-mode forward over the full sequence, backward, clip + Adam (row T)."""
    from dagnet_amd import _ops, training
    from dagnet_amd.adjacency_matrix import compute_adjs_distsim
    from dagnet_amd.utils import to_goals_one_hot
    a, model = build(wl, dev)
    model.train()
    S, T = wl["scenes"], wl["obs"] + wl["pred"]
    sizes = scene_sizes(wl, seed=rank)
    n_rows = int(sum(sizes))
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = [t.to(dev) for t in syn.make_batch(sizes, wl["obs"], wl["pred"], 1000 + rank,
                                                                                            kind=wl.get("data", "bsk"),
                                                                                            g_cells=wl.get("g_cells", (9, 10)))]
    traj, rel = torch.cat((obs, gt), 0), torch.cat((obs_rel, gt_rel), 0)
    goals = torch.cat((obs_goals, gt_goals), 0).to(torch.int32)
    n_params = sum(p.numel() for p in model.parameters())
    adj = compute_adjs_distsim(a, sse, obs, gt, compact=True)
    optim = _ops.ClipAdam(list(model.parameters()), lr=1e-3, clip=10.0)
    training.attach_grad_arena(model, optim)

    frac = 1.0 / world                          # KLD / NLL are means over the rows of the shard, CE is a sum
    if opt.sync_bn:
        _ops.set_sync_batchnorm("dist")
        _ops.BN_SYNC.rows_all = n_rows * world
        if world > 1:
            opt.no_graph = True
    eps_all = torch.empty((T - 1, n_rows, a.z_dim), dtype=torch.float32, device=dev)

    def loss_fn(m, eps):
        kld, nll, ce, _ = m(traj, rel, goals, sse, adj, eps=eps)
        return frac * (kld + nll) + 1e-2 * ce, (kld, nll, ce)

    gfb = None if opt.no_graph else training.GraphedForwardBackward(model, loss_fn, [eps_all])

    def step():
        eps_all.uniform_()                      # torch.rand_like of the reference's _reparameterize, drawn for all steps at once
        if gfb is not None:
            loss, _ = gfb(eps_all)
        else:
            optim.zero_grad()
            loss, _ = loss_fn(model, eps_all)
            loss.backward()
        training.allreduce_gradients(optim)
        optim.step()
        return loss

    for _ in range(opt.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(opt.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank != 0:
        return
    flop = train_flops_per_row_step(a, wl["agents"] or n_rows / S, model.d_dim) * n_rows * (T - 1)
    achieved = flop / (dt / opt.steps) / 1e12
    # HBM-side bytes of one optimizer step, summed over all its kernels, from the newest committed counter passes of this workload
    # (tools/train_pmc.sh -> profiles/*_pmc.json); the tape the backward pass cannot avoid is ~150 GB at this size
    traffic = None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
        try:
            traffic = json.load(open(path))[opt.workload]["step"]["traffic_bytes_per_step"]
            break
        except (OSError, ValueError, KeyError):
            continue
    hbm = None if traffic is None else traffic / (dt / opt.steps) / 1e9
    line = {
        "metric": f"training trajectories/sec (one optimizer step, teacher-forced {T}-step sequences)",
        "value": n_rows * world * opt.steps / dt, "unit": "trajectories/s", "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup,
        "ms_per_step": dt / opt.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        # layers with more than 96 outputs (g_dim = 960 only) multiply as three bf16 terms / six products (dagnet_op_gemm)
        "dtype": "f32 (bf16x3 products in the wide layers)" if a.g_dim > 96 and (_ops.set_fused(-1) & 8) else "f32",
        "data": "synthetic",
        "config": {"workload": opt.workload, "model": "dagnet", "scenes_per_gpu": S, "agents_per_scene": wl["agents"] or n_rows / S, "seq_len": T,
                   "rows_per_gpu": n_rows, "batchnorm": "synchronised over all ranks" if opt.sync_bn else "per-rank statistics",
                   "hip_graph": not opt.no_graph,
                   "parallelism": f"scenes sharded x{world}, one all-reduce of {format(n_params, ',').replace(',', ' ')} fp32 gradients per step"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": traffic,
                     "hbm": None if hbm is None else {"achieved": hbm, "peak": 8000.0, "unit": "GB/s", "frac": hbm / 8000.0,
                                                      "note": "all kernels of the step: (2 x FETCH_SIZE + WRITE_SIZE) / step time"},
                     "kernel": "layer-by-layer operators (gemm_f32.hip + train_ops.hip): bound by streaming activations, see DESIGN.md 4.5"},
        "loss": float(loss.detach()), "hbm_peak_allocated_gb": torch.cuda.max_memory_allocated(dev) / 1e9,
    }
    if not opt.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_train_baseline(wl)
    print(json.dumps(line), flush=True)


def cpu_train_baseline(wl):
    """The oracle's training step (torch CPU autograd over the dense-graph restatement) on 32 scenes."""
    from oracle import dagnet_oracle as orc
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    scenes, n = 32, wl["agents"] or 6
    a = syn.make_args(obs_len=wl["obs"], pred_len=wl["pred"], **wl["args"])
    sd = syn.random_state_dict(dagnet_amd.DAGNet(a, n), 128)
    cfg = orc.make_cfg(n_max_agents=n, **{k: getattr(a, k) for k in ("x_dim", "h_dim", "z_dim", "rnn_dim", "n_layers", "g_dim", "graph_model",
                                                                        "graph_hid", "adjacency_type", "top_k_neigh", "sigma", "alpha",
                                                                        "n_heads")})
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = syn.make_batch([n] * scenes, wl["obs"], wl["pred"], 1, kind=wl.get("data", "bsk"),
                                                                        g_cells=wl.get("g_cells", (9, 10)))
    traj, rel = torch.cat((obs, gt), 0), torch.cat((obs_rel, gt_rel), 0)
    goals = orc.one_hot(torch.cat((obs_goals, gt_goals), 0).long(), a.g_dim)
    adj = orc.adjacency_seq(cfg, sse, traj)
    eps = syn.uniform_noise(1, "eps", (traj.shape[0] - 1, traj.shape[1], a.z_dim))
    t0 = time.perf_counter()
    orc.train_step(sd, cfg, "dagnet", (traj, rel, goals, sse, adj), eps)
    dt = time.perf_counter() - t0
    return dict(value=scenes * n / dt, unit="trajectories/s", cores=torch.get_num_threads(), kind="port",
                sample=f"one training step on {scenes} scenes x {n} agents x {traj.shape[0]} steps, dense-graph oracle, torch CPU autograd")


def device_batch(S, n, obs_len, pred_len, seed, dev):
    """Basketball-shaped batch generated ON the device (the 1M-scene workload: numpy would spend minutes on it): same
    distributions and tuple layout as synthetic.make_batch (half court 45 x 50 ft, N(0, 0.5 ft) steps, 9 x 10 goal grid)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    N, T = S * n, obs_len + pred_len
    start = torch.rand((N, 2), generator=g, device=dev) * torch.tensor([45.0, 50.0], device=dev)
    steps = torch.randn((T, N, 2), generator=g, device=dev) * 0.5
    steps[0] = 0.0
    traj = start[None] + torch.cumsum(steps, 0)
    rel = torch.zeros_like(traj)
    rel[1:] = traj[1:] - traj[:-1]
    goals = (torch.clamp((traj[..., 0] // 5).long(), 0, 8) * 10 + torch.clamp((traj[..., 1] // 5).long(), 0, 9)).float()
    st = torch.arange(0, N, n, dtype=torch.int64)
    return (traj[:obs_len].contiguous(), traj[obs_len:].contiguous(), rel[:obs_len].contiguous(), rel[obs_len:].contiguous(),
            goals[:obs_len].contiguous(), goals[obs_len:].contiguous(), torch.stack((st, st + n), 1))


def visible_gpu_count(sysfs="/sys/class/kfd/kfd/topology/nodes", env=None):
    """GPUs this process's children can use, counted WITHOUT any HIP call (torch.cuda.device_count() falls back to
    hipGetDeviceCount when its amdsmi discovery fails, and that initialises HIP in a parent that is about to start children):
    KFD topology nodes with SIMDs (CPU nodes report simd_count 0), cut down by HIP_ / ROCR_ / CUDA_VISIBLE_DEVICES.
    None when the topology cannot be read (no amdgpu driver): the ranks then fail loudly by themselves in set_device()."""
    env = os.environ if env is None else env
    try:
        n = 0
        for node in sorted(os.listdir(sysfs)):
            with open(os.path.join(sysfs, node, "properties")) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
    except (OSError, ValueError):
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if var in env:
            n = min(n, len([v for v in env[var].split(",") if v.strip()]))
    return n


def self_launch(opt, argv):
    """--gpus N > 1 without a launcher: start N ranks of this script (one per GPU) and relay rank 0's output.  Runs BEFORE
    anything touches the GPU in this process, which only waits (never exec from a process that initialised the GPU) and
    makes no HIP call of its own, not even to count devices."""
    import socket
    import subprocess
    have = None if opt.rehearse else visible_gpu_count()
    if have is not None and have < opt.gpus:
        raise SystemExit(f"--gpus {opt.gpus} but only {have} visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(opt.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(opt.gpus), LOCAL_WORLD_SIZE=str(opt.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0's stdout is relayed (its JSON line and nothing else reaches our stdout); the other ranks' goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    for ln in procs[0].stdout:
        (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln)
        sys.stdout.flush()
    rc = 0
    for pr in procs:
        rc = pr.wait() or rc
        if rc:                                                           # a rank died: do not leave the others hanging
            for other in procs:
                if other.poll() is None:
                    other.kill()
    raise SystemExit(rc)


def walk_cap(n_chunks, steps, warmup, budget=1024):
    """(steps, warmup) of a chunked workload so that (steps + warmup) * n_chunks stays near ``budget`` chunk launches:
    at least one timed walk, at most one warm-up walk when the request does not fit."""
    walks = max(2, budget // max(1, n_chunks))
    if steps + warmup <= walks:
        return steps, warmup
    w = min(warmup, 1)
    return max(1, min(steps, walks - w)), w


def rehearse(opt, wl, rank, world):
    """Host logic of an N-rank run without a GPU: process group (gloo), scene sharding, the step's 3-double all-reduce, the
    max-over-ranks timing, rank 0's JSON line.  The per-rank numbers are made up (rows of the shard); nothing is measured."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = wl["scenes"] or wl["total_scenes"] // world
    rows = S * (wl["agents"] or 6)
    t0 = time.perf_counter()
    for _ in range(opt.steps):
        tot = torch.tensor([1.0 * rows, 2.0 * rows, float(rows)], dtype=torch.float64)
        dist.all_reduce(tot)
    dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    traj = torch.tensor([float(rows * wl["K"])], dtype=torch.float64)
    dist.all_reduce(traj)
    if rank == 0:
        print(json.dumps({"metric": "rehearsal (no GPU work, nothing measured)", "rehearsal": True, "value": None,
                          "n_gpus": dist.get_world_size(), "steps": opt.steps, "warmup": opt.warmup,
                          "scaling": "strong" if wl.get("total_scenes") else "weak",
                          "config": {"workload": opt.workload, "scenes_per_gpu": S, "rows_all_ranks": float(tot[2]),
                                     "trajectories_per_step": float(traj)}}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dagnet_bsk_4096x20", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-loop", action="store_true",
                    help="skip the untimed reference_loop_ms leg (K separate sample() calls): profiling runs, so that the per-kernel "
                         "averages cover only the launches of the timed region")
    ap.add_argument("--no-graph", action="store_true", help="training workloads: issue every launch from the host instead of replaying a hipGraph")
    ap.add_argument("--sync-bn", action="store_true",
                    help="training workloads: BatchNorm statistics over the rows of all ranks (the reference's whole-batch BatchNorm); "
                         "with more than one rank the step runs eagerly (its collectives are not captured into the hipGraph)")
    ap.add_argument("--rehearse", action="store_true", help="launcher + process group + reductions on the CPU (gloo), no GPU work")
    opt = ap.parse_args()
    wl = WORKLOADS[opt.workload]

    if opt.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(opt, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if opt.gpus != world:
        raise SystemExit(f"--gpus {opt.gpus} but WORLD_SIZE={world}")
    if opt.rehearse:
        return rehearse(opt, wl, rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("DAGNET_BENCH_FORCE_DIST"):      # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        # RCCL writes its version banner to the process's stdout (fd 1) at init; the contract is ONE JSON line there.  From here
        # on fd 1 is the old stderr; print() of the JSON line goes to a duplicate of the real stdout.
        sys.stdout.flush()
        real_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        sys.stdout = real_out
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        world_seen = dist.get_world_size()
    else:
        world_seen = 1

    if wl.get("train"):
        bench_train(opt, wl, rank, world, dev, dist)
        if dist is not None:
            dist.destroy_process_group()
        return
    a, model = build(wl, dev)
    K, T = wl["K"], wl["pred"]
    strong = bool(wl.get("total_scenes"))
    S = wl["total_scenes"] // world if strong else wl["scenes"]
    chunk = wl.get("chunk", S)
    from dagnet_amd.evaluate import _obs_adjacency, best_of_k_chunked
    if strong:
        sizes = [wl["agents"]] * S
        batch = device_batch(S, wl["agents"], wl["obs"], T, 1000 + rank, dev)
    else:
        sizes = scene_sizes(wl, seed=rank)      # weak scaling: every rank owns S scenes of its own (seeded by rank)
        batch = [t.to(dev) for t in syn.make_batch(sizes, wl["obs"], T, 1000 + rank, kind=wl.get("data", "bsk"),
                                                   g_cells=wl.get("g_cells", (9, 10)))]
    n_rows = int(sum(sizes))
    n = n_rows / S                      # agents per scene (mean, for ragged workloads)
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = batch
    goals = obs_goals.to(torch.int32)
    sp = scene_ptr_from_sse(sse)

    # observed steps (outside the timed region; reported separately: second call, weights packed and plan built)
    def fwd_chunk(c0, c1):
        a0, a1 = int(sp[c0]), int(sp[c1])
        o, r_, g_, gt_c = obs[:, a0:a1], obs_rel[:, a0:a1], goals[:, a0:a1], gt[:, a0:a1]
        sse_c = sse[c0:c1] - a0
        adj_obs = _obs_adjacency(a, sse_c, o, gt_c) if wl["kind"] != "vrnn" else None
        if wl["kind"] == "dagnet":
            return model(o.contiguous(), r_.contiguous(), g_.contiguous(), sse_c, adj_obs, seed=1 + c0)[3]
        if wl["kind"] == "avrnn":
            return model(r_.contiguous(), adj_obs, seq_start_end=sse_c, seed=1 + c0)[2]
        return model(r_.contiguous(), seed=1 + c0)[2]

    def fwd():
        return torch.cat([fwd_chunk(c0, min(S, c0 + chunk)) for c0 in range(0, S, chunk)], 1)
    h = fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = fwd()
    torch.cuda.synchronize()
    fwd_ms = (time.perf_counter() - t0) * 1e3
    x0, g0 = obs[-1].contiguous(), goals[-1].contiguous()
    n_chunks = (S + chunk - 1) // chunk
    steps_requested, warmup_requested = opt.steps, opt.warmup
    if n_chunks > 1:
        # a step of the chunked workload is a WALK of n_chunks launches (256 on one GPU: ~20 s): the driver's --steps 20 --warmup 5
        # would not fit its time limit, so walks are capped at ~1024 chunk launches per run (8 ranks: 32 chunks per walk, no cap
        # at 20 + 5); the line reports the steps actually timed and the request
        opt.steps, opt.warmup = walk_cap(n_chunks, opt.steps, opt.warmup)
    n_dev = torch.tensor([float(n_rows)], dtype=torch.float64, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(opt.steps)]
    plan_sp = sp[:chunk + 1] if n_chunks > 1 else sp

    if n_chunks > 1:
        def step(i, timed):
            # chunked walk (configs[4]): events bracket the whole walk; kernel_ms below is per chunk launch
            if timed:
                ev[i][0].record()
            tot = best_of_k_chunked(model, a, K, h, x0, g0, gt, sse, chunk, seed=(i + 1) * 100003)
            if timed:
                ev[i][1].record()
            if dist is not None:
                dist.all_reduce(tot)
            return tot
    else:
        metric_plan = _capi.HipPlan(None, sp, 1, device=dev)

        def step(i, timed):
            if timed:
                ev[i][0].record()
            if wl["kind"] == "dagnet":
                s = model.sample_k(K, T, h, x0, g0, sse, seed=i + 1)
            elif wl["kind"] == "avrnn":
                s = model.sample_k(K, T, h, x0, sse, seed=i + 1)
            else:
                s = model.sample_k(K, T, h, seed=i + 1)
            if timed:
                ev[i][1].record()
            r = _capi.best_of_k(metric_plan, s, x0, gt)
            tot = torch.cat([r["sums"], n_dev])
            if dist is not None:
                dist.all_reduce(tot)
            return tot

    # Launches that do not fill the chip several times over (avrnn_bsk_512x20: 3 414 tiles) would be diverted to side streams and
    # overlap with the NEXT step's launch (_capi._overlap_stream): fine for throughput, but the events below would no longer
    # bracket one launch.  The timed region keeps every launch on this stream.
    os.environ["DAGNET_NO_OVERLAP"] = "1"
    for i in range(opt.warmup):
        step(i, False)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(opt.steps):
        tot = step(i, True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    del os.environ["DAGNET_NO_OVERLAP"]
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    kernel_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])) if opt.steps else float("nan")
    tot = tot.cpu()
    traj_local = torch.tensor([float(n_rows * K)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(traj_local)
    traj_per_step = float(traj_local.item())
    value = traj_per_step * opt.steps / dt

    if rank == 0:
        d_dim = 2 * (wl["agents"] or 32)
        launch_flop = float(sum(flops_per_row_step(wl["kind"], a, ns, d_dim) * ns for ns in sizes)) * T * K   # this rank's rows
        fl = launch_flop / (n_rows * K)                                # FLOP per sampled trajectory (mean)
        if n_chunks > 1:
            # the events bracket n_chunks (sample + metric) launch pairs; the metric kernels are ~0.1 % of it
            kernel_ms /= n_chunks
            launch_flop /= n_chunks
        achieved = launch_flop / (kernel_ms * 1e-3) / 1e12
        info = model.hip_plan(dev, np.ascontiguousarray(plan_sp - plan_sp[0]), K).info()
        # the dense products of the register-resident kernels and of the wide goal head (fp16 x2) run on the 16-bit matrix pipe
        # unless switched off; the generic kernels always use fp32 MFMAs
        from dagnet_amd import _ops
        split_on = bool(_ops.set_fused(-1) & 8)
        split = None
        if split_on and "wide goal head" in info["sample_kernel"] and os.environ.get("DAGNET_WIDE_BF16X3"):
            split = ("bf16x3", PEAK_SPLIT_BF16X3_TFLOPS, "bf16 dense / 6 products (three bf16 terms per value; wide layers)")
        elif split_on and "wide goal head" in info["sample_kernel"]:
            split = ("fp16x2", PEAK_SPLIT_FP16X2_TFLOPS, "fp16 dense / 3 products (two fp16 terms per value; wide layers and step kernel)")
        elif split_on and "register-resident" in info["sample_kernel"]:
            split = ("fp16x2", PEAK_SPLIT_FP16X2_TFLOPS, "fp16 dense / 3 products (two fp16 terms per value)")
        split_bf16 = split is not None
        peak = split[1] if split else PEAK_FP32_MATRIX_TFLOPS
        line = {
            "metric": "sampled trajectories/sec (best-of-20 rollout)", "value": value, "unit": "trajectories/s",
            "n_gpus": world_seen, "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": dt / opt.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            # fp32 in, fp32 out, fp32 accumulation; where the products run on the 16-bit matrix pipe as split terms (error of the
            # fp32 MFMA's class, tests/test_gpu_properties.py) the label says so
            "dtype": f"f32 ({split[0]} products)" if split else "f32", "data": "synthetic",
            "config": {"workload": opt.workload, "model": wl["kind"], "scenes_per_gpu": S, "agents_per_scene": n,
                       "samples": K, "obs_len": wl["obs"], "pred_len": T, "rows_per_gpu": n_rows * K,
                       "chunks_per_step": n_chunks, "scenes_per_chunk": chunk,
                       "tile_rows": info["tile_rows"], "tiles": info["n_tiles"], "lds_bytes": info["lds_bytes"],
                       "parallelism": f"scenes sharded x{world_seen}, 1 all-reduce of 3 doubles per step"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak,
                         "peak_kind": split[2] if split else "fp32 matrix",
                         "traffic": pmc_traffic("dagnet_bsk_4096x20" if strong else opt.workload, info["sample_kernel"]),
                         "kernel": info["sample_kernel"], "kernel_ms": kernel_ms,
                         "flop_per_trajectory": fl},
            "forward_obs_ms": fwd_ms, "steps_requested": steps_requested, "warmup_requested": warmup_requested,
            "ade": float(tot[0] / (tot[2] * T)), "fde": float(tot[1] / tot[2]),
        }
        if split and split[0] == "fp16x2" and opt.workload == "dagnet_bsk_4096x20":
            # what bounds the split kernel: the SIMD's vector issue port, not the matrix pipe.  From the committed counter
            # passes of this workload (profiles/*_pmc.json: instructions and cycles of the full-size launches) priced with the issue
            # costs of MI355X_MICROARCH.md: 4 clocks per VALU instruction, 8 per transcendental and per MFMA; cycles, not time -
            # the clock under this kernel varies between boxes (1.88 - 1.99 GHz, profiles/r03_clock_under_load_fp16x2.txt)
            import glob
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
                try:
                    ks = json.load(open(path))[opt.workload]["kernels"]
                    c = next(v for k_, v in ks.items() if "rollout_v2_kernel<2, 2, 32, 1, true>" in k_)
                    ws = float(info["n_tiles"] * T)
                    mfma, vec = c["SQ_INSTS_MFMA"] / ws, (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / ws
                    trans = 908.0            # v_exp / v_rcp / v_log of one wave-step (tools/isa_mix.py on the seed-mode path)
                    need = (vec - trans) * 4 + trans * 8 + mfma * 8
                    have = c["GRBM_GUI_ACTIVE"] / 8 * 1024 / ws
                    line["roofline"]["vector_issue"] = {
                        "vector_instructions_per_wave_step": vec, "mfma_per_wave_step": mfma, "issue_clocks_needed": need,
                        "clocks_measured_per_wave_step_and_simd": have, "frac": need / have, "mfma_pipe_frac": mfma * 32 / have,
                        "source": os.path.basename(path)}
                    break
                except (OSError, ValueError, KeyError, StopIteration):
                    continue
        if split_bf16 and n_chunks == 1 and not opt.no_reference_loop:
            # the same launch with the products on fp32 MFMAs (untimed leg, 2 launches): the fp32-matrix roofline of round 2
            _ops.set_fused(_ops.set_fused(-1) & ~8)
            try:
                step(0, False)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                step(1, False)
                torch.cuda.synchronize()
                f32_ms = (time.perf_counter() - t0) * 1e3
            finally:
                _ops.set_fused(-1)
            line["roofline"]["fp32_mfma"] = {"ms_per_step": f32_ms, "achieved": launch_flop / (f32_ms * 1e-3) / 1e12, "peak": PEAK_FP32_MATRIX_TFLOPS,
                                             "unit": "TFLOP/s", "frac": launch_flop / (f32_ms * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
                                             "note": "the same step with DAGNET_NO_BF16X3 semantics (whole step incl. metrics, wall clock)"}
        if n_chunks == 1 and not opt.no_reference_loop:
            # the reference's evaluation loop AS WRITTEN (models/dagnet/evaluate_bsk.py:113-114): K separate sample() calls
            # (untimed leg, steady state: one untimed pass of the loop first, as a loader's second batch would see it).  The calls overlap on the device through
            # three side streams (_capi._overlap_stream); reference_loop_serial_ms is the same loop on the caller's stream alone.
            def ref_loop(k0, count):
                for k in range(k0, k0 + count):
                    if wl["kind"] == "dagnet":
                        model.sample(T, h, x0, g0, sse, seed=1000 + k)
                    elif wl["kind"] == "avrnn":
                        model.sample(T, h, x0, sse, seed=1000 + k)
                    else:
                        model.sample(T, h, seed=1000 + k)
            for key, serial in (("reference_loop_ms", False), ("reference_loop_serial_ms", True)):
                if serial:
                    os.environ["DAGNET_NO_OVERLAP"] = "1"
                try:
                    ref_loop(999, K)        # one untimed pass: the K = 1 plan, and the allocator's blocks of the side streams (a
                    torch.cuda.synchronize()  # hipMalloc inside the loop synchronises the device and hides the overlap)
                    t0 = time.perf_counter()
                    ref_loop(0, K)
                    torch.cuda.synchronize()
                    line[key] = (time.perf_counter() - t0) * 1e3
                finally:
                    if serial:
                        del os.environ["DAGNET_NO_OVERLAP"]
        if not opt.no_cpu_baseline and world == 1:         # the CPU legs run on rank 0 of the single-GPU run only
            line["cpu_baseline"] = cpu_baseline(wl, budget_s=12.0)
            line["cpu_baseline_torch_nn"] = cpu_baseline_stock(wl, budget_s=8.0)
            # BASELINE configs[0]: VRNN baseline, best-of-20 on the CPU reference path (32 scenes x 10 agents, obs 10 / pred 40)
            line["cpu_baseline_vrnn"] = cpu_baseline_stock(WORKLOADS["vrnn_bsk_4096x20"], budget_s=6.0)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
