"""The differentiable (train-mode) forward of the three models and one training step (row T):
models/dagnet/model.py:138-194, models/avrnn/model.py:93-143, models/baseline/model.py:70-115 and
models/dagnet/train_bsk.py:259-291.

Layer by layer over all N rows of the batch: every layer is an operator of dagnet_amd._ops (HIP forward and backward);
torch only allocates, keeps the autograd tape and adds up the per-step loss scalars.  BatchNorm uses batch statistics
over all rows of the batch (gat_model.py:21 in train mode) and updates its running statistics like nn.BatchNorm1d.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np
import torch

from . import _capi, _ops
from ._ops import ACT_LRELU, ACT_NONE, ACT_RELU, LossSumFn, NllFn, SoftmaxFn, VaeFn, gat_train, gcn_train, gru_step, linear

_PLANS: "OrderedDict[tuple, _capi.HipPlan]" = OrderedDict()


def table_plan(scene_ptr, device) -> _capi.HipPlan:
    """Scene tables only (no packed weights: the training operators read the live parameters)."""
    sp = np.ascontiguousarray(np.asarray(scene_ptr, dtype=np.int32))
    key = (torch.device(device).index or 0, sp.tobytes())
    plan = _PLANS.get(key)
    if plan is None:
        plan = _capi.HipPlan(None, sp, 1, device=device)
        _PLANS[key] = plan
        while len(_PLANS) > 8:
            _PLANS.popitem(last=False)
    else:
        _PLANS.move_to_end(key)
    return plan


def attach_grad_arena(model, optimizer=None, flat_params=None) -> torch.Tensor:
    """Give every parameter's .grad a view of ONE flat fp32 buffer (zeroed) and mark the parameters as gradient sinks: the
    weight / bias gradient kernels of _ops then accumulate in place over the time steps instead of returning fresh tensors
    for autograd to add up, and the data-parallel all-reduce is one call on the flat buffer.  Plain autograd semantics
    otherwise (gradients accumulate over backward() calls until zeroed); torch.autograd.grad(loss, params) is not
    supported for sink parameters (they report None).

    flat_params (default: when ``optimizer`` is an _ops.ClipAdam): the parameters' storage and the optimizer's moments become
    views of flat buffers in the SAME layout (every tensor starts at a multiple of 64 floats, so the kernels' 16-byte paths stay
    available; the gaps hold zeros), and clip + Adam over the whole model is ONE launch instead of one per parameter
    (_ops.ClipAdam.step).  The parameters stay nn.Parameters of their modules; whatever re-allocates one later (module.to(), a
    GRU's flatten_parameters) only sends the optimizer back to its per-tensor launches."""
    params = [p for p in model.parameters() if p.requires_grad]
    if flat_params is None:
        flat_params = isinstance(optimizer, _ops.ClipAdam) and [id(p) for p in optimizer.params] == [id(p) for p in params]
    align = 64 if flat_params else 1
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + align - 1) // align * align
    total = off if flat_params else sum(p.numel() for p in params)
    dev = params[0].device
    arena = torch.zeros(total, dtype=torch.float32, device=dev)
    defer = _ops.DeferredGrads()              # partial sums over the time steps, one reduction per backward pass
    for p, o in zip(params, offs):
        p.grad = arena[o:o + p.numel()].view_as(p)
        p._hip_sink = True
        p._hip_defer = defer
    model._hip_grad_arena = arena
    if optimizer is not None:
        optimizer.arena = arena
    if flat_params:
        fp, fm, fv = (torch.zeros(total, dtype=torch.float32, device=dev) for _ in range(3))
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(params, offs)):
                n = p.numel()
                fp[o:o + n].copy_(p.detach().reshape(-1))
                fm[o:o + n].copy_(optimizer.m[i].reshape(-1))
                fv[o:o + n].copy_(optimizer.v[i].reshape(-1))
                p.data = fp[o:o + n].view_as(p)
                optimizer.m[i] = fm[o:o + n].view_as(p)
                optimizer.v[i] = fv[o:o + n].view_as(p)
        optimizer.flat = dict(p=fp, m=fm, v=fv, offs=offs)
    return arena


def _mlp2(seq, xs, cols, gidx=None, gcol=0, gdim=0):
    if len(xs) <= 3:
        return _ops.mlp2(seq, xs, cols, act=ACT_LRELU, gidx=gidx, gcol=gcol, gdim=gdim)     # both layers in one kernel
    h = linear(seq[0], xs, cols, act=ACT_LRELU, gidx=gidx, gcol=gcol, gdim=gdim)
    return linear(seq[2], [h], [0], act=ACT_LRELU)


def _mlp2_heads(seq, mean, logvar, xs, cols, gidx=None, gcol=0, gdim=0):
    """-> (block output as a fan-out alias, mean, logvar): the block and its two heads in one kernel."""
    if len(xs) <= 3:
        return _ops.mlp2(seq, xs, cols, act=ACT_LRELU, gidx=gidx, gcol=gcol, gdim=gdim, heads=(mean, logvar))
    out = _ops.fanout(_mlp2(seq, xs, cols, gidx, gcol, gdim))
    return out, linear(mean, [out], [0]), linear(logvar, [out], [0])


def _graph(mod, x, adj_t, plan, stacks=None):
    from .gat import GAT
    return gat_train(mod, x, adj_t, plan, stacks) if isinstance(mod, GAT) else gcn_train(mod, x, adj_t, plan)


def _eps(eps, t, N, Z, device):
    if eps is not None:
        return eps[t - 1].contiguous()
    return torch.rand((N, Z), dtype=torch.float32, device=device)      # torch.rand_like in the reference (model.py:118-121)


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def forward_train(model, scene_ptr, traj_rel, traj=None, goals=None, adj=None, eps=None):
    """-> (KLD[1], NLL[1], CE[1], h[L,N,R]); differentiable w.r.t. the parameters of ``model``.

    Storage comes from _ops.POOL: what an operator produces at time step t is slice t of one stacked buffer per call site,
    so that every layer's weight gradient is one product over all steps at the end of the backward pass; tensors with
    several consumers go through _ops.fanout (one shared input-gradient buffer instead of a chain of torch adds); the
    per-row losses of all steps are summed by one kernel each (_ops.LossSumFn); the stacked GAT weights are built once."""
    from .gat import GAT
    kind = {"VRNN": "vrnn", "AVRNN": "avrnn", "DAGNet": "dagnet"}[type(model).__name__]
    dev = traj_rel.device
    if not traj_rel.is_cuda:
        raise _capi.DagnetHipError("the training step needs a ROCm device; there is no CPU path")
    T, N, _ = traj_rel.shape
    R, H, Z, L = model.rnn_dim, model.h_dim, model.z_dim, model.n_layers
    traj_rel = _f32(traj_rel)
    plan = table_plan(scene_ptr, dev) if kind != "vrnn" else None
    if kind == "dagnet":
        G, D = model.g_dim, model.d_dim
        traj = _f32(traj)
        goals = goals.to(torch.int32).contiguous()
    if eps is not None:
        eps = _f32(eps)
    fan = _ops.fanout
    pool = _ops.POOL
    pool.begin(T, dev)
    try:
        graph = blend = None
        st_goal = st_hid = None
        if kind != "vrnn":
            graph = model.graph_hiddens if kind == "dagnet" else model.graph
            blend = model.lg_hiddens if kind == "dagnet" else model.lg
            st_hid = _ops.gat_stacks(graph) if isinstance(graph, GAT) else None
            if kind == "dagnet" and isinstance(model.graph_goals, GAT):
                st_goal = _ops.gat_stacks(model.graph_goals)
        h = [torch.zeros((N, R), dtype=torch.float32, device=dev) for _ in range(L)]
        packs = _ops.gru_packs(model.rnn, (H, H))           # the GRU weights in the fused layer kernel's order, once per step
        # phi_x does not depend on the recurrence: one launch over the whole sequence instead of one per time step
        phi_x_all = None
        if T > 1 and os.environ.get("DAGNET_NO_TIME_BATCH") is None:
            x_all = traj_rel[1:].reshape((T - 1) * N, traj_rel.shape[2])
            phi_x_all = _ops.UnstackFn.apply(_mlp2(model.phi_x, [x_all], [0]), T - 1)
        k_rows, n_rows, c_rows = [], [], []
        br = _ops.BR
        for t in range(1, T):
            pool.next_t(t)
            x_t = traj_rel[t]
            hl = h[-1]
            # Three branches of the step feed only losses or the latent draw, not the recurrence: the goal head and the decoder go to
            # side streams (_ops.Branches) and run beside the phi_x -> encoder -> phi_z -> GRU -> graph chain; the prior is marked as
            # a branch too but stays on the main stream by default (Branches.inline).
            if kind == "dagnet":
                d_t = _ops.disposition(plan, traj[t], D)
                g_prev, g_t = goals[t - 1], goals[t]
                with br.fork("goal", dev):          # goal head: ends in the CE rows
                    hid = linear(model.dec_goal[0], [d_t, hl], [0, D], act=ACT_RELU, gidx=g_prev, gcol=D + R, gdim=G)
                    dec_g = fan(SoftmaxFn.apply(linear(model.dec_goal[2], [hid], [0])))
                    g_graph = _graph(model.graph_goals, dec_g, adj[t], plan, st_goal)
                    c_rows.append(_ops.SoftmaxCeFn.apply(linear(model.lg_goals[0], [dec_g, g_graph], [0, G]), g_t))
            with br.fork("prior", dev):             # prior: joins at the latent draw
                if kind == "dagnet":
                    _, mu_p, lv_p = _mlp2_heads(model.prior, model.prior_mean, model.prior_logvar, [hl], [G], gidx=g_t, gcol=0, gdim=G)
                else:
                    _, mu_p, lv_p = _mlp2_heads(model.prior, model.prior_mean, model.prior_logvar, [hl], [0])
            phi_x = fan(phi_x_all[t - 1] if phi_x_all is not None else _mlp2(model.phi_x, [x_t], [0]))
            if kind == "dagnet":
                _, mu_e, lv_e = _mlp2_heads(model.enc, model.enc_mean, model.enc_logvar, [phi_x, hl], [0, H + G], gidx=g_t, gcol=H, gdim=G)
            else:
                _, mu_e, lv_e = _mlp2_heads(model.enc, model.enc_mean, model.enc_logvar, [phi_x, hl], [0, H])
            br.join("prior", device=dev)
            z, k = VaeFn.apply(mu_e, lv_e, mu_p, lv_p, _eps(eps, t, N, Z, dev))
            k_rows.append(k)
            phi_z = fan(_mlp2(model.phi_z, [z], [0]))
            with br.fork("dec", dev):               # decoder: ends in the NLL rows
                if kind == "dagnet":
                    _, mu_d, lv_d = _mlp2_heads(model.dec, model.dec_mean, model.dec_logvar, [d_t, phi_z, hl], [0, D + G, D + G + H],
                                                gidx=g_t, gcol=D, gdim=G)
                else:
                    _, mu_d, lv_d = _mlp2_heads(model.dec, model.dec_mean, model.dec_logvar, [phi_z, hl], [0, H])
                n_rows.append(NllFn.apply(mu_d, lv_d, x_t))
            h = gru_step(model.rnn, [phi_x, phi_z], [0, H], h, packs)
            if kind != "vrnn":
                h_g = _graph(graph, h[-1], adj[t], plan, st_hid)
                h[-1] = fan(_ops.linear_pieces(blend, [h_g, h[-1]], [0, R]))
        br.join(device=dev)
        zero = torch.zeros(1, dtype=torch.float32, device=dev)
        kld = LossSumFn.apply(1.0 / N, *k_rows) if k_rows else zero
        nll = LossSumFn.apply(1.0 / N, *n_rows) if n_rows else zero
        ce = LossSumFn.apply(1.0, *c_rows) if c_rows else zero
        return kld, nll, ce, torch.stack(h, 0)
    finally:
        pool.end()


def train_step(model, optimizer: "_ops.ClipAdam", batch, kld_weight=1.0, ce_weight=1.0, eps=None, row_fraction=1.0):
    """The reference's training step (train_bsk.py:259-291): zero_grad, train-mode forward, loss = kld_weight KLD + NLL
    (+ ce_weight CE), backward, clip_grad_norm_ + Adam (both inside ``optimizer.step()``).

    batch: VRNN (x_rel,) | A-VRNN (x_rel, adj[, seq_start_end]) | DAG-Net (traj, traj_rel, goals_ohe, seq_start_end, adj).
    Data parallel (scenes sharded over ranks): KLD and NLL are means over the rows, CE is a sum, so a rank holding
    ``row_fraction`` = N_local / N_global of the rows scales its KLD/NLL terms by that fraction; the all-reduced (summed)
    gradients then equal the single-process gradients (up to per-shard BatchNorm statistics).
    -> dict(loss, kld, nll, ce) of [1] device tensors (no host synchronisation)."""
    model.train()
    optimizer.zero_grad()
    name = type(model).__name__
    if name == "VRNN":
        kld, nll, _ = model(batch[0], eps=eps)
        ce = torch.zeros_like(kld)
        loss = row_fraction * (kld_weight * kld + nll)
    elif name == "AVRNN":
        kld, nll, _ = model(batch[0], batch[1], seq_start_end=batch[2] if len(batch) > 2 else None, eps=eps)
        ce = torch.zeros_like(kld)
        loss = row_fraction * (kld_weight * kld + nll)
    else:
        kld, nll, ce, _ = model(*batch, eps=eps)
        loss = row_fraction * (kld_weight * kld + nll) + ce_weight * ce
    loss.backward()
    _ops.BR.join()                   # the branches' streams (forward_train) rejoin before the gradients are read
    allreduce_gradients(optimizer if getattr(optimizer, "arena", None) is not None else optimizer.params)
    optimizer.step()
    return dict(loss=loss.detach(), kld=kld.detach(), nll=nll.detach(), ce=ce.detach())


def empty_step(model, optimizer: "_ops.ClipAdam"):
    """The step of a rank whose share of the batch is empty (a short last batch split over more ranks than it has scenes):
    zero gradients, the same gradient all-reduce and optimizer step as everybody else, so that the ranks' collectives stay
    paired and their weights and Adam moments identical."""
    model.train()
    optimizer.zero_grad()
    for p in optimizer.params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    allreduce_gradients(optimizer if getattr(optimizer, "arena", None) is not None else optimizer.params)
    optimizer.step()


class GraphedForwardBackward:
    """Forward + backward of one training step captured once in a hipGraph (torch.cuda.CUDAGraph) and replayed: the
    layer-by-layer step is ~20 000 small launches, which the host cannot issue as fast as the GPU retires them.
    The batch shape (scene layout, sequence length) is fixed at capture; new data is copied into the static inputs.
    The gradient all-reduce and the optimizer step stay outside the graph.

        gfb = GraphedForwardBackward(model, loss_fn, example_inputs)     # loss_fn(model, *inputs) -> (loss, aux tuple)
        loss, aux = gfb(*inputs); optimizer.step()
    """

    def __init__(self, model, loss_fn, inputs, warmup=2):
        self.model, self.loss_fn = model, loss_fn
        self.static = [t.clone() if torch.is_tensor(t) else t for t in inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        arena = getattr(model, "_hip_grad_arena", None)

        def zero():
            if arena is not None:
                arena.zero_()                    # flat gradient buffer (attach_grad_arena): one memset, captured with the step
            else:
                for p in model.parameters():
                    p.grad = None
        with torch.cuda.stream(side):
            for _ in range(warmup):              # plans, scratch buffers and the allocator reach their steady state
                zero()
                loss, _ = loss_fn(model, *self.static)
                loss.backward()
                _ops.BR.join()
        torch.cuda.current_stream().wait_stream(side)
        if arena is None:
            zero()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            if arena is not None:
                zero()
            self.loss, self.aux = loss_fn(model, *self.static)
            self.loss.backward()
            _ops.BR.join()                       # every side stream of the step rejoins the capturing stream

    def __call__(self, *inputs):
        for dst, src in zip(self.static, inputs):
            if torch.is_tensor(dst) and src is not dst:
                dst.copy_(src)
        self.graph.replay()
        return self.loss, self.aux


def allreduce_gradients(params, world_size=None):
    """``params``: a parameter list, or an optimizer whose gradients live in one flat arena (attach_grad_arena).
    Data-parallel training: one flat all-reduce(SUM) of every gradient over the ranks (RCCL over xGMI),
    between backward() and the optimizer step.  Scenes are sharded, weights replicated; the losses are sums over rows,
    so gradients add up (no averaging), exactly as one rank seeing the whole batch -- except BatchNorm, whose batch
    statistics stay per shard (documented deviation)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or (world_size or dist.get_world_size()) == 1:
        return
    arena = getattr(params, "arena", None)
    if arena is not None:                      # attach_grad_arena: the gradients already are one flat buffer
        dist.all_reduce(arena)
        return
    # every parameter takes part, a missing gradient as zeros: the ranks' flat buffers must have the same length whatever
    # each rank's graph touched (an empty shard's empty_step materialises zeros for all of them)
    params = [p for p in params if p.requires_grad]
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat)
    off = 0
    for p in params:
        n = p.numel()
        if p.grad is None:
            p.grad = flat[off:off + n].view_as(p).clone()
        else:
            p.grad.copy_(flat[off:off + n].view_as(p))
        off += n
