"""CPU, world_size 2 (gloo): the multi-GPU path's only collective.  Each rank takes its scene shard
(dagnet_amd.sharding), computes the shard's (ade_sum, fde_sum, n_agents) -- here with the CPU oracle
standing in for the HIP rollout, which needs a GPU -- and the all-reduced totals must equal the
single-process evaluation of the whole batch."""
import os
import socket

import numpy as np
import pytest
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


# ---- loop.train / loop.validate under data parallelism (host logic; the HIP step is replaced by the oracle) -----------------
class _CpuAdam:
    """The attributes of _ops.ClipAdam that loop / training read, on torch's own CPU optimizer."""

    def __init__(self, params, lr=1e-2, clip=10.0):
        self.params, self.clip, self.arena = list(params), clip, None
        self.opt = torch.optim.Adam(self.params, lr=lr)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self):
        torch.nn.utils.clip_grad_norm_(self.params, self.clip)
        self.opt.step()


def _row_eps(x_rel, z):
    """Noise that is a function of the row's own data, so a scene draws the same numbers whichever rank holds it."""
    u = (x_rel.abs().sum(-1, keepdim=True) * 7.3) % 1.0          # [T,N,1]
    return ((u + torch.linspace(0, 0.9, z)[None, None, :]) % 1.0)[1:]


def _oracle_train_step(cfg):
    def step(model, optimizer, batch, kld_weight=1.0, ce_weight=1.0, eps=None, row_fraction=1.0):
        from dagnet_amd.training import allreduce_gradients
        optimizer.zero_grad()
        params = dict(model.named_parameters())
        x = batch[0]
        kld, nll, _ = orc.vrnn_forward(params, cfg, x, _row_eps(x, cfg.z_dim))
        loss = row_fraction * (kld_weight * kld + nll)
        loss.sum().backward()
        for p in optimizer.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        allreduce_gradients(optimizer.params)
        optimizer.step()
        z = torch.zeros(1)
        return dict(loss=loss.detach().reshape(1), kld=kld.detach().reshape(1), nll=nll.detach().reshape(1), ce=z)
    return step


def _loop_epoch(rank, world):
    import dagnet_amd
    from dagnet_amd import data as D, loop, training
    model, sd, cfg, batch, _ = _train_setup()
    args = syn.make_args(h_dim=16, z_dim=8, rnn_dim=16, n_layers=2, pred_len=T_PRED, obs_len=T_OBS, CE_weight=1.0)
    loader = D.SceneBatcher(list(batch[:6]), syn.scene_ptr(SIZES), batch_size=3, rank=rank, world=world)
    opt = _CpuAdam(model.parameters())
    training.train_step = _oracle_train_step(cfg)            # stand-in for the HIP operators (this process only)
    out = loop.train(args, 1, loader, model, loop.kld_warmup(2, False), opt)
    return out, {k: p.detach().clone() for k, p in model.named_parameters()}, [b[0].shape[1] for b in loader]


def _loop_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res, params, rows = _loop_epoch(rank, world)
    out.put((rank, res, {k: v.numpy() for k, v in params.items()}, rows))
    dist.destroy_process_group()


def test_loop_train_data_parallel_equals_single_process():
    """7 scenes in batches of 3 over 2 ranks: the last batch has ONE scene, so rank 0's share of it is empty and it must
    still join the gradient all-reduce (SceneBatcher never skips a batch); KLD / NLL are weighted by the row share."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][3] == [3, 6, 0] and got[1][3] == [7, 6, 7]              # rows per batch: every rank sees all 3 batches
    want, wparams, _ = _loop_epoch(0, 1)
    for rank, res, params, _ in got:
        for k in ("loss", "kld", "nll"):
            assert res[k] == pytest.approx(want[k], rel=2e-5), (rank, k)
        for k, w in wparams.items():
            np.testing.assert_allclose(params[k], w.numpy(), rtol=0, atol=2e-5, err_msg=f"rank {rank} {k}")
    for k in got[0][2]:
        assert np.array_equal(got[0][2][k], got[1][2][k]), k               # the ranks' weights stay bit-identical


def test_scene_batcher_shares_partition_every_batch():
    from dagnet_amd import data as D
    sizes = [2, 3, 1, 4]
    batch = syn.make_batch(sizes, 2, 2, 1)
    ptr = syn.scene_ptr(sizes)
    for world in (1, 2, 3, 5):
        per_rank = [list(D.SceneBatcher(list(batch[:6]), ptr, batch_size=4, rank=r, world=world)) for r in range(world)]
        assert all(len(b) == 1 for b in per_rank)                           # nobody skips the batch
        rows = [b[0][0].shape[1] for b in per_rank]
        assert sum(rows) == sum(sizes)
        cat = torch.cat([b[0][0] for b in per_rank], 1)
        assert torch.equal(cat, batch[0])                                   # contiguous shares in rank order
        for b in per_rank:
            sse = b[0][-1]
            assert sse.shape[1] == 2 and (sse.shape[0] == 0 or int(sse[-1, 1]) == b[0][0].shape[1])
