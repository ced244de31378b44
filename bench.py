#!/usr/bin/env python
"""Benchmark of the hot path: one best-of-K rollout (K sample() rollouts + min-over-K ADE/FDE) per step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

Metric (BASELINE.json): sampled trajectories / second = scenes x agents x K / t_rollout, inputs
(h after the observed steps, last positions, last goals, GT future) already resident in HBM.
N > 1: one process per GPU (torch.distributed / RCCL), scenes sharded, weak scaling (per-GPU work
fixed), one all-reduce of (ade_sum, fde_sum, n_agents) per step; value = all ranks' trajectories / max time.
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
        hits = [float(v["traffic_bytes_per_launch"]) for name, v in ks.items()
                if kernel.split(" ")[0] + "<" in name and "traffic_bytes_per_launch" in v]
        if hits:
            return max(hits)        # several instantiations of the kernel: the dominant (largest) one
    return None

WORKLOADS = {
    # BASELINE.json configs[2]: DAG-Net full, basketball-shaped, 4096 scenes x 20 samples, 1 GPU
    "dagnet_bsk_4096x20": dict(kind="dagnet", scenes=4096, agents=10, K=20, obs=10, pred=40, args=dict(n_layers=2)),
    # configs[1]: A-VRNN (single GAT on hidden states), 512 scenes
    "avrnn_bsk_512x20": dict(kind="avrnn", scenes=512, agents=10, K=20, obs=10, pred=40, args=dict(n_layers=1)),
    # configs[3]: DAG-Net SDD-shaped: obs 8 / pred 12, ragged scenes (2..32 agents, mean 6), g_dim = 32 x 30 = 960
    "dagnet_sdd_1024x20": dict(kind="dagnet", scenes=1024, agents=0, K=20, obs=8, pred=12, args=dict(n_layers=2, g_dim=960),
                               data="sdd", g_cells=(32, 30)),
    # configs[0] shape on the GPU: VRNN baseline
    "vrnn_bsk_4096x20": dict(kind="vrnn", scenes=4096, agents=10, K=20, obs=10, pred=40, args=dict(n_layers=2)),
    # configs[4], training leg (row T): one optimizer step of DAG-Net on a chunk of scenes, full 50-step sequence
    "train_dagnet_bsk_1024": dict(kind="dagnet", scenes=1024, agents=10, K=1, obs=10, pred=40, args=dict(n_layers=2), train=True),
    "train_dagnet_bsk_8192": dict(kind="dagnet", scenes=8192, agents=10, K=1, obs=10, pred=40, args=dict(n_layers=2), train=True),
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
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = [t.to(dev) for t in syn.make_batch(sizes, wl["obs"], wl["pred"], 1000 + rank)]
    traj, rel = torch.cat((obs, gt), 0), torch.cat((obs_rel, gt_rel), 0)
    goals = torch.cat((obs_goals, gt_goals), 0).to(torch.int32)
    from dagnet_amd.adjacency_matrix import BlockDiagAdj
    adj = compute_adjs_distsim(a, sse, obs, gt, compact=True)
    optim = _ops.ClipAdam(list(model.parameters()), lr=1e-3, clip=10.0)
    training.attach_grad_arena(model, optim)

    frac = 1.0 / world                          # KLD / NLL are means over the rows of the shard, CE is a sum
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
    flop = train_flops_per_row_step(a, wl["agents"], 2 * wl["agents"]) * n_rows * (T - 1)
    achieved = flop / (dt / opt.steps) / 1e12
    line = {
        "metric": "training trajectories/sec (one optimizer step, teacher-forced 50-step sequences)",
        "value": n_rows * world * opt.steps / dt, "unit": "trajectories/s", "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup,
        "ms_per_step": dt / opt.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": opt.workload, "model": "dagnet", "scenes_per_gpu": S, "agents_per_scene": wl["agents"], "seq_len": T,
                   "rows_per_gpu": n_rows, "parallelism": f"scenes sharded x{world}, one all-reduce of 184 350 fp32 gradients per step"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": None,
                     "kernel": "layer-by-layer operators (gemm_f32.hip + train_ops.hip): bound by streaming activations, see DESIGN.md 4.5"},
        "loss": float(loss.detach()),
    }
    if not opt.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_train_baseline(wl)
    print(json.dumps(line), flush=True)


def cpu_train_baseline(wl):
    """The oracle's training step (torch CPU autograd over the dense-graph restatement) on 32 scenes."""
    from oracle import dagnet_oracle as orc
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    scenes, n = 32, wl["agents"]
    a = syn.make_args(obs_len=wl["obs"], pred_len=wl["pred"], **wl["args"])
    sd = syn.random_state_dict(dagnet_amd.DAGNet(a, n), 128)
    cfg = orc.make_cfg(n_max_agents=n, **{k: getattr(a, k) for k in ("x_dim", "h_dim", "z_dim", "rnn_dim", "n_layers", "g_dim", "graph_model",
                                                                        "graph_hid", "adjacency_type", "top_k_neigh", "sigma", "alpha",
                                                                        "n_heads")})
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = syn.make_batch([n] * scenes, wl["obs"], wl["pred"], 1)
    traj, rel = torch.cat((obs, gt), 0), torch.cat((obs_rel, gt_rel), 0)
    goals = orc.one_hot(torch.cat((obs_goals, gt_goals), 0).long(), a.g_dim)
    adj = orc.adjacency_seq(cfg, sse, traj)
    eps = syn.uniform_noise(1, "eps", (traj.shape[0] - 1, traj.shape[1], a.z_dim))
    t0 = time.perf_counter()
    orc.train_step(sd, cfg, "dagnet", (traj, rel, goals, sse, adj), eps)
    dt = time.perf_counter() - t0
    return dict(value=scenes * n / dt, unit="trajectories/s", cores=torch.get_num_threads(), kind="port",
                sample=f"one training step on {scenes} scenes x {n} agents x {traj.shape[0]} steps, dense-graph oracle, torch CPU autograd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dagnet_bsk_4096x20", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="training workloads: issue every launch from the host instead of replaying a hipGraph")
    opt = ap.parse_args()
    wl = WORKLOADS[opt.workload]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if opt.gpus != world and world > 1:
        raise SystemExit(f"--gpus {opt.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("DAGNET_BENCH_FORCE_DIST"):      # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if wl.get("train"):
        bench_train(opt, wl, rank, world, dev, dist)
        if dist is not None:
            dist.destroy_process_group()
        return
    a, model = build(wl, dev)
    S, K, T = wl["scenes"], wl["K"], wl["pred"]
    sizes = scene_sizes(wl, seed=rank)
    n_rows = int(sum(sizes))
    n = n_rows / S                      # agents per scene (mean, for ragged workloads)
    # weak scaling: every rank owns S scenes of its own (seeded by rank)
    batch = [t.to(dev) for t in syn.make_batch(sizes, wl["obs"], T, 1000 + rank, kind=wl.get("data", "bsk"),
                                               g_cells=wl.get("g_cells", (9, 10)))]
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = batch
    from dagnet_amd.evaluate import _obs_adjacency
    goals = obs_goals.to(torch.int32)
    # observed steps (outside the timed region; reported separately: second call, weights packed and plan built)
    adj_obs = _obs_adjacency(a, sse, obs, gt) if wl["kind"] != "vrnn" else None

    def fwd():
        if wl["kind"] == "dagnet":
            return model(obs, obs_rel, goals, sse, adj_obs, seed=1)[3]
        if wl["kind"] == "avrnn":
            return model(obs_rel, adj_obs, seq_start_end=sse, seed=1)[2]
        return model(obs_rel, seed=1)[2]
    h = fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = fwd()
    torch.cuda.synchronize()
    fwd_ms = (time.perf_counter() - t0) * 1e3
    sp = scene_ptr_from_sse(sse)
    metric_plan = _capi.HipPlan(None, sp, 1, device=dev)
    x0, g0 = obs[-1].contiguous(), goals[-1].contiguous()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(opt.steps)]

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
        tot = torch.cat([r["sums"], torch.tensor([float(n_rows)], dtype=torch.float64, device=dev)])
        if dist is not None:
            dist.all_reduce(tot)
        return tot

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
        achieved = launch_flop / (kernel_ms * 1e-3) / 1e12
        info = model.hip_plan(dev, sp, K).info()
        line = {
            "metric": "sampled trajectories/sec (best-of-20 rollout)", "value": value, "unit": "trajectories/s",
            "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": dt / opt.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": opt.workload, "model": wl["kind"], "scenes_per_gpu": S, "agents_per_scene": n,
                       "samples": K, "obs_len": wl["obs"], "pred_len": T, "rows_per_gpu": n_rows * K,
                       "tile_rows": info["tile_rows"], "tiles": info["n_tiles"], "lds_bytes": info["lds_bytes"],
                       "parallelism": f"scenes sharded x{world}, 1 all-reduce of 3 doubles per step"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": pmc_traffic(opt.workload, info["sample_kernel"]),
                         "kernel": info["sample_kernel"], "kernel_ms": kernel_ms,
                         "flop_per_trajectory": fl},
            "forward_obs_ms": fwd_ms,
            "ade": float(tot[0] / (tot[2] * T)), "fde": float(tot[1] / tot[2]),
        }
        if not opt.no_cpu_baseline and world == 1:         # the CPU leg runs on rank 0 of the single-GPU run only
            line["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
