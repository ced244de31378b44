"""The reference's epoch functions around the hot path: ``train`` / ``validate`` / ``test``
of models/dagnet/train_bsk.py:103-312 (and the avrnn / baseline analogues), the KLD warm-up (:377-380), the
ReduceLROnPlateau gating (:245-248, :350-356), the checkpoint dictionary (:80-93) and the epoch loop of ``main``
(:382-417) - without its logging, TensorBoard and plotting side effects.

Everything numeric is the HIP path: teacher-forced forward (eval mode: forward_v2 / generic kernel, train mode:
training.py operators), ``sample``, ``relative_to_abs``, the 'sum'-mode ADE / FDE.  Losses and metrics of an epoch are
accumulated in device tensors; the host reads them once per epoch (the reference calls ``.item()`` four times per
batch).
"""
from __future__ import annotations

import numpy as np
import os

import torch

from . import _ops, training
from .adjacency_matrix import compute_adjs, compute_adjs_distsim, compute_adjs_knnsim
from .evaluate import _obs_adjacency, evaluate
from .utils import average_displacement_error, final_displacement_error, relative_to_abs

WARMUP_EPOCHS = 50          # train_bsk.py:378


def kld_warmup(num_epochs, warmup=True, warmup_epochs=WARMUP_EPOCHS):
    """train_bsk.py:377-380: weight of the KLD term per epoch (index epoch - 1): 0 -> 1 linearly over the first 50
    epochs when ``args.warmup``, 1 afterwards."""
    w = np.ones(num_epochs)
    if warmup:
        n = min(warmup_epochs, num_epochs)
        w[:n] = np.linspace(0, 1, num=warmup_epochs)[:n]
    return w


def _kind(model):
    return {"DAGNet": "dagnet", "AVRNN": "avrnn", "VRNN": "vrnn"}[type(model).__name__]


def _split(data, kind):
    """Batch tuple of either loader (7 fields, or 9 with non_linear_ped / loss_mask before seq_start_end)."""
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals = data[:6]
    return obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, data[-1]


_M64 = (1 << 64) - 1


def _mix64(x):
    """splitmix64 finaliser: a bijection of 64-bit integers with good avalanche."""
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def sub_seed(seed, batch_index, phase, rank=None):
    """Philox seed of one call site.  The counter of csrc/rng.h is (local row, t, j, stream) only, so everything else that
    must not share noise goes into the key: the teacher-forced pass (phase 0) and the rollout (phase 1) of one batch, the
    batches of one epoch, the epoch functions (``epoch_seed``) and - data parallel - the RANK, whose shard numbers its rows
    from 0 like every other rank's.  A hash of the tuple, not a sum: (seed, batch b + 1) and (seed + 1, batch b) differ
    (ADVICE r2).  None stays None; the result fits 63 bits."""
    if seed is None:
        return None
    if rank is None:
        rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
    h = _mix64(int(seed) & _M64)
    for v in (batch_index, phase, rank):
        h = _mix64(h ^ (int(v) & _M64))
    return h >> 1


def epoch_seed(seed, epoch, what):
    """Seed of one epoch function (what: 0 validate, 1 test, 2 evaluate(test), 3 evaluate(valid)) - hashed, so that no two of them
    and no two epochs meet whatever the batch count."""
    return None if seed is None else _mix64(_mix64(int(seed) & _M64) ^ ((int(epoch) << 8) | int(what))) >> 1


def _dist_on():
    return torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1


def _row_fraction(loader, n_local, dev):
    """N_local / N_global of the batch in flight (1.0 in a single process).  A SceneBatcher knows the whole batch on every
    rank; any other loader costs one all-reduce of the row count."""
    if not _dist_on():
        return 1.0
    n_global = getattr(loader, "global_rows", 0)
    if not n_global:
        t = torch.tensor([float(n_local)], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t)
        n_global = float(t.item())
    return float(n_local) / float(n_global) if n_global else 0.0


def _eval_pass(args, epch, loader, model, warmup, seed=None):
    """validate() / test() of the reference minus the side effects: per batch the teacher-forced pass over the observed
    steps, the loss, ONE sample() from the last h, 'sum'-mode ADE / FDE.  -> dict of python floats.
    Data parallel: KLD / NLL of a batch are means over its rows, so a rank weighs them by its share of the batch's rows;
    the seven partial sums are all-reduced once per pass, so every rank returns the same numbers (and steps its plateau
    scheduler identically)."""
    kind = _kind(model)
    dev = next(model.parameters()).device
    acc = torch.zeros(7, dtype=torch.float64, device=dev)       # loss, kld, nll, ce, ade, fde, total_traj
    w = float(warmup[epch - 1])
    n_batches = 0
    model.eval()
    with torch.no_grad():
        for data in loader:
            obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = _split([t.to(dev) for t in data], kind)
            s_f, s_s = sub_seed(seed, n_batches, 0), sub_seed(seed, n_batches, 1)
            n_batches += 1
            frac = _row_fraction(loader, obs.shape[1], dev)
            if obs.shape[1] == 0:                     # this rank's share of a short batch is empty
                continue
            if kind == "dagnet":
                adj = _obs_adjacency(args, sse, obs, gt)
                goals = obs_goals.to(torch.int32)
                kld, nll, ce, h = model(obs, obs_rel, goals, sse, adj, seed=s_f)
                samples_rel = model.sample(args.pred_len, h, obs[-1], goals[-1], sse, seed=s_s)
            elif kind == "avrnn":
                adj = _obs_adjacency(args, sse, obs, gt)
                kld, nll, h = model(obs_rel, adj, seq_start_end=sse, seed=s_f)
                ce = torch.zeros_like(kld)
                samples_rel = model.sample(args.pred_len, h, obs[-1], sse, seed=s_s)
            else:
                kld, nll, h = model(obs_rel, seed=s_f)
                ce = torch.zeros_like(kld)
                samples_rel = model.sample(args.pred_len, h, seed=s_s)
            kld, nll = frac * kld, frac * nll
            loss = nll + w * kld + (ce * args.CE_weight if kind == "dagnet" else 0.0)
            samples = relative_to_abs(samples_rel, obs[-1])
            acc[0] += loss[0].double(); acc[1] += kld[0].double(); acc[2] += nll[0].double(); acc[3] += ce[0].double()
            acc[4] += average_displacement_error(samples, gt).double()
            acc[5] += final_displacement_error(samples[-1], gt[-1]).double()
            acc[6] += samples.shape[1]
    if _dist_on():
        torch.distributed.all_reduce(acc)
    a = acc.cpu()
    n_scenes = len(loader.dataset)
    return dict(loss=float(a[0]) / n_scenes, kld=float(a[1]) / n_scenes, nll=float(a[2]) / n_scenes, ce=float(a[3]) / n_scenes,
                ade=float(a[4] / (a[6] * args.pred_len)), fde=float(a[5] / a[6]), total_traj=int(a[6]))


def validate(args, epch, valid_loader, model, warmup, lr_scheduler=None, seed=None):
    """train_bsk.py:177-248 -> (ade_val, fde_val); steps the plateau scheduler on ADE once ``epch`` is past the settling
    period (100 epochs for basketball, ``args.scheduler_after`` overrides: the SDD scripts use 200)."""
    r = _eval_pass(args, epch, valid_loader, model, warmup, seed=seed)
    if getattr(args, "lr_scheduler", False) and lr_scheduler is not None and epch > getattr(args, "scheduler_after", 100):
        lr_scheduler.step(r["ade"])
    validate.last = r
    return r["ade"], r["fde"]


def test(args, epch, test_loader, model, warmup, seed=None):
    """train_bsk.py:103-174 -> dict(loss, kld, nll, ce, ade, fde) (the reference only logs them)."""
    return _eval_pass(args, epch, test_loader, model, warmup, seed=seed)


def _full_adjacency(args, sse, obs, gt):
    if args.adjacency_type == 0:
        return compute_adjs(args, sse, compact=True, device=obs.device)
    if args.adjacency_type == 1:
        return compute_adjs_distsim(args, sse, obs, gt, compact=True)
    return compute_adjs_knnsim(args, sse, obs, gt, compact=True)


def train(args, epch, train_loader, model, warmup, optimizer, sync_batchnorm=None):
    """train_bsk.py:251-312: one epoch of optimizer steps.  DAG-Net trains on the full obs + pred sequence, A-VRNN and
    VRNN on the observed steps only (models/avrnn/train_bsk.py:256, models/baseline/train_bsk.py:218).
    ``optimizer`` is an ``_ops.ClipAdam`` (clip_grad_norm_(args.clip) + Adam in one).  -> mean losses per scene.
    Data parallel (torch.distributed initialised, the loader handing every rank its share of each batch): a rank scales
    its KLD / NLL (means over rows) by its share of the batch's rows so that the all-reduced gradients are the
    single-process ones; a rank whose share is empty still joins the gradient all-reduce and steps the optimizer; the
    epoch's loss sums are all-reduced once.
    ``sync_batchnorm`` (default: the environment variable DAGNET_SYNC_BN): train-mode BatchNorm statistics over the rows of ALL
    ranks, as the reference's single-process BatchNorm sees them (_ops.SyncBnFn: one all-gather + one all-reduce of a few hundred
    bytes per BatchNorm call); off, every rank normalises with its own shard's statistics.  With it a rank
    whose share of a batch is EMPTY cannot skip the forward pass's collectives: use batches of at least one scene per rank."""
    from . import _ops
    kind = _kind(model)
    dev = next(model.parameters()).device
    acc = torch.zeros(4, dtype=torch.float64, device=dev)
    w = float(warmup[epch - 1])
    if sync_batchnorm is None:
        sync_batchnorm = os.environ.get("DAGNET_SYNC_BN", "0") not in ("", "0")
    sync = bool(sync_batchnorm) and _dist_on() and kind != "vrnn"
    for data in train_loader:
        obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = _split([t.to(dev) for t in data], kind)
        frac = _row_fraction(train_loader, obs.shape[1], dev)
        if obs.shape[1] == 0:
            if sync:
                raise RuntimeError("synchronised BatchNorm needs at least one scene per rank in every batch (this rank's share is empty)")
            training.empty_step(model, optimizer)
            continue
        prev_bn = _ops.set_sync_batchnorm("dist") if sync else None
        if sync:
            _ops.BN_SYNC.rows_all = int(round(obs.shape[1] / frac))
        if kind == "dagnet":
            assert len(obs) + len(gt) == args.obs_len + args.pred_len
            adj = _full_adjacency(args, sse, obs, gt)
            batch = (torch.cat((obs, gt), 0), torch.cat((obs_rel, gt_rel), 0),
                     torch.cat((obs_goals, gt_goals), 0).to(torch.int32), sse, adj)
            r = training.train_step(model, optimizer, batch, kld_weight=w, ce_weight=args.CE_weight, row_fraction=frac)
        elif kind == "avrnn":
            r = training.train_step(model, optimizer, (obs_rel, _full_adjacency(args, sse, obs, gt), sse), kld_weight=w, row_fraction=frac)
        else:
            r = training.train_step(model, optimizer, (obs_rel,), kld_weight=w, row_fraction=frac)
        if sync:
            _ops.set_sync_batchnorm(*prev_bn)
            _ops.BN_SYNC.rows_all = None
        acc[0] += r["loss"][0].double(); acc[1] += frac * r["kld"][0].double(); acc[2] += frac * r["nll"][0].double(); acc[3] += r["ce"][0].double()
    if _dist_on():
        torch.distributed.all_reduce(acc)
    a = (acc / len(train_loader.dataset)).cpu()
    return dict(loss=float(a[0]), kld=float(a[1]), nll=float(a[2]), ce=float(a[3]))


class PlateauScheduler:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, threshold=1e-2, patience=10, factor=0.5) (train_bsk.py:350-356;
    mode 'min', relative threshold, no cooldown, min_lr 0, eps 1e-8) for an optimizer that keeps ``lr`` as an attribute
    (``_ops.ClipAdam``).  ``state_dict`` uses torch's key names so checkpoints carry over."""

    def __init__(self, optimizer, threshold=1e-2, patience=10, factor=0.5, eps=1e-8):
        self.optimizer, self.threshold, self.patience, self.factor, self.eps = optimizer, threshold, patience, factor, eps
        self.best, self.num_bad_epochs, self.last_epoch = float("inf"), 0, 0

    def step(self, metric):
        metric = float(metric)
        self.last_epoch += 1
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.num_bad_epochs = metric, 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            new_lr = self.optimizer.lr * self.factor
            if self.optimizer.lr - new_lr > self.eps:
                self.optimizer.lr = new_lr
            self.num_bad_epochs = 0

    def state_dict(self):
        return dict(best=self.best, num_bad_epochs=self.num_bad_epochs, last_epoch=self.last_epoch, threshold=self.threshold,
                    patience=self.patience, factor=self.factor, eps=self.eps, mode="min", threshold_mode="rel", cooldown=0,
                    cooldown_counter=0, min_lrs=[0], _last_lr=[self.optimizer.lr])

    def load_state_dict(self, sd):
        self.best, self.num_bad_epochs, self.last_epoch = sd["best"], sd["num_bad_epochs"], sd["last_epoch"]
        if sd.get("_last_lr"):
            self.optimizer.lr = float(sd["_last_lr"][0])


def optimizer_state_dict(optimizer: "_ops.ClipAdam"):
    """``torch.optim.Adam.state_dict()`` layout (state: {i: step, exp_avg, exp_avg_sq}, param_groups) for a ClipAdam."""
    state = {i: dict(step=torch.tensor(float(optimizer.step_count)), exp_avg=m.detach().cpu().clone(), exp_avg_sq=v.detach().cpu().clone())
             for i, (m, v) in enumerate(zip(optimizer.m, optimizer.v))} if optimizer.step_count else {}
    group = dict(lr=optimizer.lr, betas=tuple(optimizer.betas), eps=optimizer.eps, weight_decay=0, amsgrad=False,
                 params=list(range(len(optimizer.params))))
    return dict(state=state, param_groups=[group])


def load_optimizer_state_dict(optimizer: "_ops.ClipAdam", sd):
    g = sd["param_groups"][0]
    optimizer.lr, optimizer.betas, optimizer.eps = float(g["lr"]), tuple(g["betas"]), float(g["eps"])
    for i, st in sd.get("state", {}).items():
        i = int(i)
        optimizer.m[i].copy_(st["exp_avg"])
        optimizer.v[i].copy_(st["exp_avg_sq"])
        optimizer.step_count = int(float(st["step"]))


def save_checkpoint(fn, args, epoch, model, optimizer, lr_scheduler, best_ade, ade_list_val, fde_list_val, ade_list_test,
                    fde_list_test):
    """train_bsk.py:80-93: the same dictionary (keys and nesting), so either code base resumes from the other's file."""
    torch.save({
        'args': dict(args.__dict__),
        'epoch': epoch,
        'best_ade': best_ade,
        'model_state_dict': {k: v.detach().cpu() for k, v in model.state_dict().items()},
        'optimizer_state_dict': optimizer_state_dict(optimizer),
        'lr_scheduler_state_dict': lr_scheduler.state_dict(),
        'ade_list_val': ade_list_val,
        'fde_list_val': fde_list_val,
        'ade_list_test': ade_list_test,
        'fde_list_test': fde_list_test,
    }, fn)


def load_checkpoint(fn, model, optimizer=None, lr_scheduler=None):
    """-> the checkpoint dict; tensors only (``weights_only=True``), model / optimizer / scheduler restored when given."""
    ck = torch.load(fn, map_location="cpu", weights_only=True)
    model.load_state_dict(ck["model_state_dict"])
    if optimizer is not None and ck.get("optimizer_state_dict"):
        load_optimizer_state_dict(optimizer, ck["optimizer_state_dict"])
    if lr_scheduler is not None and ck.get("lr_scheduler_state_dict"):
        lr_scheduler.load_state_dict(ck["lr_scheduler_state_dict"])
    return ck


class FitContext:
    """What the callbacks of ``fit`` get: the live model / optimizer / scheduler / history and ``save(fn)``, which writes the
    reference's checkpoint dictionary (train_bsk.py:80-93) for the epoch in flight."""

    def __init__(self, args, model, optimizer, lr_scheduler, hist):
        self.args, self.model, self.optimizer, self.lr_scheduler, self.hist, self.epoch = args, model, optimizer, lr_scheduler, hist, 0

    def save(self, fn):
        h = self.hist
        save_checkpoint(fn, self.args, self.epoch, self.model, self.optimizer, self.lr_scheduler, h["best_ade"], h["ade_list_val"],
                        h["fde_list_val"], h["ade_list_test"], h["fde_list_test"])


def fit(args, model, train_loader, valid_loader, test_loader, start_epoch=0, best_ade=float("inf"), on_best=None, on_epoch=None,
        seed=None, optimizer=None, lr_scheduler=None):
    """The epoch loop of ``main`` (train_bsk.py:382-417): train -> validate -> test per epoch, best-ADE tracking,
    best-of-``args.num_samples`` evaluation every ``args.eval_every`` epochs.  File handling is the caller's:
    ``on_best(epoch, ade, ctx)`` is invoked where the reference writes its best checkpoint (:397-405) and
    ``on_epoch(epoch, ctx)`` at the end of every epoch (periodic / evaluation checkpoints, :408-417); ``ctx`` is a
    ``FitContext`` (``ctx.save(fn)`` writes the reference's dictionary with optimizer and scheduler state).
    Resuming (:340-366): build ``optimizer`` / ``lr_scheduler``, restore them with ``load_checkpoint`` and pass them in with
    ``start_epoch`` / ``best_ade`` - Adam moments, step count, learning rate and plateau counters then carry on.
    -> history dict."""
    if optimizer is None:
        optimizer = _ops.ClipAdam(list(model.parameters()), lr=args.learning_rate, clip=args.clip)
    if getattr(optimizer, "arena", None) is None:
        training.attach_grad_arena(model, optimizer)
    sched = lr_scheduler if lr_scheduler is not None else PlateauScheduler(optimizer)
    warmup = kld_warmup(args.num_epochs, getattr(args, "warmup", False))
    hist = dict(train=[], val=[], test=[], ade_list_val=[], fde_list_val=[], ade_list_test=[], fde_list_test=[], best_ade=best_ade,
                optimizer=optimizer, lr_scheduler=sched)
    ctx = FitContext(args, model, optimizer, sched, hist)
    for epoch in range(start_epoch + 1, args.num_epochs + 1):
        ctx.epoch = epoch
        hist["train"].append(train(args, epoch, train_loader, model, warmup, optimizer))
        ade, _ = validate(args, epoch, valid_loader, model, warmup, sched, seed=epoch_seed(seed, epoch, 0))      # fresh noise every epoch
        hist["val"].append(validate.last)
        hist["test"].append(test(args, epoch, test_loader, model, warmup, seed=epoch_seed(seed, epoch, 1)))
        if 0 <= ade < hist["best_ade"]:
            hist["best_ade"] = ade
            if on_best is not None:
                on_best(epoch, ade, ctx)
        if getattr(args, "eval_every", 0) and epoch % args.eval_every == 0:
            a, f = evaluate(args, model, test_loader, args.num_samples, seed=epoch_seed(seed, epoch, 2))
            av, fv = evaluate(args, model, valid_loader, args.num_samples, seed=epoch_seed(seed, epoch, 3))
            hist["ade_list_test"].append((a, epoch)); hist["fde_list_test"].append((f, epoch))
            hist["ade_list_val"].append((av, epoch)); hist["fde_list_val"].append((fv, epoch))
        if on_epoch is not None:
            on_epoch(epoch, ctx)
    return hist
