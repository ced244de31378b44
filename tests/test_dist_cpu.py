"""CPU, world_size 2 (gloo): the multi-GPU path's only collective.  Each rank takes its scene shard
(dagnet_amd.sharding), computes the shard's (ade_sum, fde_sum, n_agents) -- here with the CPU oracle
standing in for the HIP rollout, which needs a GPU -- and the all-reduced totals must equal the
single-process evaluation of the whole batch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dagnet_amd import synthetic as syn
from dagnet_amd.evaluate import finalize_metrics
from dagnet_amd.sharding import shard_batch, shard_scenes
from oracle import dagnet_oracle as orc

SIZES = [3, 5, 2, 6, 4, 2, 7]
T_OBS, T_PRED, K, SEED = 4, 6, 3, 5


def _shard_sums(batch, sd, cfg, a, row0, n_total):
    obs, gt, obs_rel, gt_rel, og, pg, sse = batch
    N = obs.shape[1]
    eps_all = syn.uniform_noise(SEED, "eps", (K, T_PRED, n_total, a.z_dim))[:, :, row0:row0 + N]
    h = syn.uniform_noise(SEED, "h0", (a.n_layers, n_total, a.rnn_dim))[:, row0:row0 + N] - 0.5
    with torch.no_grad():
        s = torch.stack([orc.avrnn_sample(sd, cfg, T_PRED, h, obs[-1], sse, eps_all[k]) for k in range(K)])
    a_sum, f_sum = orc.evaluate_best_of_k(s, obs[-1], gt, sse)
    return torch.tensor([float(a_sum), float(f_sum), float(N)], dtype=torch.float64)


def _setup():
    import dagnet_amd
    a = syn.make_args(h_dim=16, z_dim=8, rnn_dim=16, n_layers=1, pred_len=T_PRED, obs_len=T_OBS)
    sd = syn.random_state_dict(dagnet_amd.AVRNN(a), SEED)
    cfg = orc.make_cfg(h_dim=16, z_dim=8, rnn_dim=16, n_layers=1)
    return a, sd, cfg, syn.make_batch(SIZES, T_OBS, T_PRED, SEED)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, sd, cfg, batch = _setup()
    local, local_sizes = shard_batch(batch, SIZES, world, rank)
    row0 = int(sum(SIZES[: shard_scenes(SIZES, world, rank)[0]]))
    tot = _shard_sums(local, sd, cfg, a, row0, sum(SIZES))
    ade, fde = finalize_metrics(tot, T_PRED)          # all-reduces when a process group is up
    if rank == 0:
        out.put((ade, fde))
    dist.destroy_process_group()


def test_sharded_evaluation_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, sd, cfg, batch = _setup()
    tot = _shard_sums(batch, sd, cfg, a, 0, sum(SIZES))
    want = (float(tot[0] / (tot[2] * T_PRED)), float(tot[1] / tot[2]))
    assert abs(got[0] - want[0]) < 1e-9 * max(1, abs(want[0])) + 1e-6 and abs(got[1] - want[1]) < 1e-6


# ---- data-parallel training step (row T): one flat all-reduce of the gradients ----------------------
def _vrnn_grads(x_rel, eps, sd, cfg, weight):
    """Oracle gradients of weight * (KLD + NLL) on a shard (the HIP operators need a GPU; this test covers the reduction)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    kld, nll, _ = orc.vrnn_forward(params, cfg, x_rel, eps)
    grads = torch.autograd.grad((weight * (kld + nll)).sum(), list(params.values()), allow_unused=True)
    return {k: (g if g is not None else torch.zeros_like(params[k])) for k, g in zip(params, grads)}


def _train_setup():
    import dagnet_amd
    a = syn.make_args(h_dim=16, z_dim=8, rnn_dim=16, n_layers=2, pred_len=T_PRED, obs_len=T_OBS)
    model = dagnet_amd.VRNN(a)
    sd = syn.random_state_dict(model, SEED)
    model.load_state_dict(sd)
    cfg = orc.make_cfg(h_dim=16, z_dim=8, rnn_dim=16, n_layers=2)
    batch = syn.make_batch(SIZES, T_OBS, T_PRED, SEED)
    eps = syn.uniform_noise(SEED, "eps_train", (T_OBS - 1, sum(SIZES), a.z_dim))
    return model, sd, cfg, batch, eps


def _train_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dagnet_amd.training import allreduce_gradients
    model, sd, cfg, batch, eps = _train_setup()
    local, local_sizes = shard_batch(batch, SIZES, world, rank)
    row0 = int(sum(SIZES[: shard_scenes(SIZES, world, rank)[0]]))
    n_local = int(sum(local_sizes))
    g = _vrnn_grads(local[2], eps[:, row0:row0 + n_local], sd, cfg, n_local / sum(SIZES))
    for k, p in model.named_parameters():
        p.grad = g[k].clone()
    allreduce_gradients(list(model.parameters()))
    if rank == 0:
        out.put({k: p.grad.numpy() for k, p in model.named_parameters()})
    dist.destroy_process_group()


def test_sharded_training_gradients_equal_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    model, sd, cfg, batch, eps = _train_setup()
    want = _vrnn_grads(batch[2], eps, sd, cfg, 1.0)
    for k, w in want.items():
        np.testing.assert_allclose(got[k], w.numpy(), rtol=2e-4, atol=2e-6, err_msg=k)
