"""GPU: one training step through the HIP operators (dagnet_amd.training / dagnet_amd._ops, C ABI dagnet_op_*) against
the reference's own step captured in tests/golden/train_*.npz: losses, every parameter's gradient, the clipped-gradient
total norm, the parameters after one Adam step and the BatchNorm running statistics."""
import numpy as np
import pytest
import torch

from tests import cases
from tests.test_oracle_train import check_step

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def hip_train_step(meta, model, arena=False):
    from dagnet_amd import _ops
    from dagnet_amd.adjacency_matrix import compute_adjs, compute_adjs_distsim, compute_adjs_knnsim
    from dagnet_amd.utils import to_goals_one_hot
    batch, eps = cases.build_train_inputs(meta)
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = [t.to(DEV) for t in batch]
    eps = eps.to(DEV)
    args = cases.syn.make_args(obs_len=meta["obs_len"], pred_len=meta["pred_len"], **meta["args"])
    kind = meta["kind"]
    model.train()
    opt = _ops.ClipAdam(list(model.parameters()), lr=meta["lr"], clip=meta["clip"])
    if arena:
        from dagnet_amd.training import attach_grad_arena
        attach_grad_arena(model, opt)
    opt.zero_grad()
    if kind != "vrnn":
        if args.adjacency_type == 0:
            adj = compute_adjs(args, sse, device=DEV)
        elif args.adjacency_type == 1:
            adj = compute_adjs_distsim(args, sse, obs, gt)
        else:
            adj = compute_adjs_knnsim(args, sse, obs, gt)
    if kind == "vrnn":
        kld, nll, _ = model(obs_rel, eps=eps)
        ce = torch.zeros_like(kld)
        loss = meta["kld_weight"] * kld + nll
    elif kind == "avrnn":
        kld, nll, _ = model(obs_rel, adj, seq_start_end=sse, eps=eps)
        ce = torch.zeros_like(kld)
        loss = meta["kld_weight"] * kld + nll
    else:
        traj, rel = torch.cat((obs, gt), 0), torch.cat((obs_rel, gt_rel), 0)
        g_ohe = to_goals_one_hot(torch.cat((obs_goals, gt_goals), 0), args.g_dim)
        kld, nll, ce, _ = model(traj, rel, g_ohe, sse, adj, eps=eps)
        loss = meta["kld_weight"] * kld + nll + meta["ce_weight"] * ce
    loss.backward()
    grads = {k: (p.grad.detach().cpu().clone() if p.grad is not None else torch.zeros_like(p).cpu()) for k, p in model.named_parameters()}
    opt.step()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    return dict(kld=kld.detach().cpu(), nll=nll.detach().cpu(), ce=ce.detach().cpu(), loss=loss.detach().cpu(), grads=grads,
                total_norm=opt.total_norm().cpu(), params={k: sd[k] for k, _ in model.named_parameters()},
                buffers={k: v for k, v in sd.items()})


@pytest.mark.parametrize("name", cases.list_train_cases())
def test_train_step_matches_reference(name):
    meta, gold = cases.load_golden(name)
    _, model, _ = cases.build_model(meta)
    model = model.to(DEV)
    out = hip_train_step(meta, model)
    check_step(out, gold, meta["lr"], meta["clip"])


@pytest.mark.parametrize("name", ["train_dagnet_tiny_gat", "train_dagnet_tiny_gcn", "train_vrnn_tiny", "train_dagnet_sdd_g960"])
def test_train_step_with_gradient_arena(name):
    """Flat gradient buffer + in-place accumulation by the backward kernels: same step as the plain autograd path."""
    meta, gold = cases.load_golden(name)
    _, model, _ = cases.build_model(meta)
    out = hip_train_step(meta, model.to(DEV), arena=True)
    check_step(out, gold, meta["lr"], meta["clip"])


def test_train_mode_forward_without_grad_uses_batch_statistics():
    """model.train() under no_grad: same losses as the differentiable call (batch-statistics BatchNorm), no tape."""
    meta, gold = cases.load_golden("train_avrnn_tiny_gat")
    _, model, _ = cases.build_model(meta)
    model = model.to(DEV).train()
    from dagnet_amd.adjacency_matrix import compute_adjs_distsim
    batch, eps = cases.build_train_inputs(meta)
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = [t.to(DEV) for t in batch]
    args = cases.syn.make_args(obs_len=meta["obs_len"], pred_len=meta["pred_len"], **meta["args"])
    with torch.no_grad():
        kld, nll, h = model(obs_rel, compute_adjs_distsim(args, sse, obs, gt), seq_start_end=sse, eps=eps.to(DEV))
    assert not kld.requires_grad
    np.testing.assert_allclose(kld.cpu().numpy(), gold["kld"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(nll.cpu().numpy(), gold["nll"], rtol=1e-4, atol=1e-5)


# ---- three consecutive optimizer steps ((f)3) ------------------------------------------------------------------------
def _multistep_batch(meta, i):
    from dagnet_amd.adjacency_matrix import compute_adjs, compute_adjs_distsim, compute_adjs_knnsim
    from tests.test_oracle_train import multistep_inputs
    batch, eps = multistep_inputs(meta, i)
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = [t.to(DEV) for t in batch]
    args = cases.syn.make_args(obs_len=meta["obs_len"], pred_len=meta["pred_len"], **meta["args"])
    if meta["kind"] == "vrnn":
        return (obs_rel,), eps.to(DEV)
    adj = (compute_adjs(args, sse, device=DEV) if args.adjacency_type == 0 else
           compute_adjs_distsim(args, sse, obs, gt) if args.adjacency_type == 1 else compute_adjs_knnsim(args, sse, obs, gt))
    if meta["kind"] == "avrnn":
        return (obs_rel, adj, sse), eps.to(DEV)
    goals = torch.cat((obs_goals, gt_goals), 0).to(torch.int32)
    return (torch.cat((obs, gt), 0), torch.cat((obs_rel, gt_rel), 0), goals, sse, adj), eps.to(DEV)


def _run_step(meta, model, opt, i):
    """training.train_step on the batch of step i -> the per-step record check_multistep wants.  The optimizer clips the
    gradients in place, so the raw gradients are p.grad / coef."""
    from dagnet_amd import training
    batch, eps = _multistep_batch(meta, i)
    r = training.train_step(model, opt, batch, kld_weight=meta["kld_weights"][i], ce_weight=meta["ce_weight"], eps=eps)
    tn = float(opt.total_norm().cpu())
    coef = min(1.0, meta["clip"] / (tn + 1e-6))
    grads = {k: (p.grad.detach().cpu() / coef if p.grad is not None else torch.zeros_like(p).cpu()) for k, p in model.named_parameters()}
    return dict(kld=r["kld"].cpu(), nll=r["nll"].cpu(), ce=r["ce"].cpu(), loss=r["loss"].cpu(), grads=grads, total_norm=tn)


def _final(model, opt):
    names = [k for k, _ in model.named_parameters()]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    return dict(params={k: sd[k] for k in names}, buffers=sd, m={k: m.cpu() for k, m in zip(names, opt.m)},
                v={k: v.cpu() for k, v in zip(names, opt.v)})


@pytest.mark.parametrize("arena", [False, True])
@pytest.mark.parametrize("name", ["multistep_dagnet_tiny_gat", "multistep_avrnn_tiny_gcn"])
def test_three_training_steps_match_reference(name, arena):
    """Adam bias correction, moment carry-over and BatchNorm running statistics over three reference steps
    (train_bsk.py:286-291 x 3, different batches and KLD weights)."""
    from dagnet_amd import _ops
    from dagnet_amd.training import attach_grad_arena
    from tests.test_oracle_train import check_multistep
    meta, gold = cases.load_golden(name)
    _, model, _ = cases.build_model(meta)
    model = model.to(DEV)
    opt = _ops.ClipAdam(list(model.parameters()), lr=meta["lr"], clip=meta["clip"])
    if arena:
        attach_grad_arena(model, opt)
    steps = [_run_step(meta, model, opt, i) for i in range(meta["steps"])]
    assert opt.step_count == meta["steps"]
    check_multistep(steps, _final(model, opt), gold, meta)


@pytest.mark.parametrize("name", ["multistep_dagnet_tiny_gat", "multistep_avrnn_tiny_gcn"])
def test_checkpoint_resume_reproduces_the_third_step(name, tmp_path):
    """two steps -> loop.save_checkpoint -> a fresh model / optimizer / scheduler via loop.load_checkpoint -> the third step:
    parameters, Adam moments and BatchNorm buffers must equal the reference's after its third step (train_bsk.py:80-93,
    :340-366)."""
    import dagnet_amd
    from dagnet_amd import _ops, loop
    from tests.test_oracle_train import check_multistep
    meta, gold = cases.load_golden(name)
    args, model, _ = cases.build_model(meta)
    model = model.to(DEV)
    opt = _ops.ClipAdam(list(model.parameters()), lr=meta["lr"], clip=meta["clip"])
    sched = loop.PlateauScheduler(opt)
    steps = [_run_step(meta, model, opt, i) for i in range(2)]
    fn = tmp_path / "checkpoint_epoch_2.pth"
    loop.save_checkpoint(fn, args, 2, model, opt, sched, 1.0, [], [], [], [])
    fresh = (dagnet_amd.DAGNet(args, meta["n_max_agents"]) if meta["kind"] == "dagnet" else dagnet_amd.AVRNN(args)).to(DEV)
    opt2 = _ops.ClipAdam(list(fresh.parameters()), lr=123.0, clip=meta["clip"])
    ck = loop.load_checkpoint(fn, fresh, opt2, loop.PlateauScheduler(opt2))
    assert ck["epoch"] == 2 and opt2.step_count == 2 and opt2.lr == pytest.approx(meta["lr"])
    # the reference's own optimizer accepts the file's optimizer state
    ref_opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros_like(p, device="cpu")) for p in fresh.parameters()], lr=1.0)
    ref_opt.load_state_dict(ck["optimizer_state_dict"])
    steps.append(_run_step(meta, fresh, opt2, 2))
    check_multistep(steps, _final(fresh, opt2), gold, meta)


@pytest.mark.parametrize("kind,kw", [("dagnet", dict(n_layers=2)), ("avrnn", dict(n_layers=1)), ("vrnn", dict(n_layers=2))])
def test_side_streams_do_not_change_the_step(kind, kw):
    """training.forward_train forks the prior, the goal head and the decoder of every time step onto side streams
    (_ops.Branches).  On 512 scenes x 10 agents x 12 steps - kernels long enough to overlap - the losses and every gradient
    must equal the single-stream step (same weights, same noise) up to summation order, eagerly and as a replayed hipGraph."""
    import dagnet_amd
    from dagnet_amd import _ops, training, synthetic as syn
    from dagnet_amd.adjacency_matrix import compute_adjs_distsim
    a = syn.make_args(obs_len=4, pred_len=8, **kw)
    sizes = [10] * 512
    obs, gt, obs_rel, gt_rel, og, pg, sse = [t.to(DEV) for t in syn.make_batch(sizes, 4, 8, 17)]
    traj, rel = torch.cat((obs, gt), 0), torch.cat((obs_rel, gt_rel), 0)
    goals = torch.cat((og, pg), 0).to(torch.int32)
    N = rel.shape[1]
    eps = syn.uniform_noise(17, "eps", (rel.shape[0] - 1, N, a.z_dim)).to(DEV)

    def run(streams, graphed):
        m = {"vrnn": lambda: dagnet_amd.VRNN(a), "avrnn": lambda: dagnet_amd.AVRNN(a), "dagnet": lambda: dagnet_amd.DAGNet(a, 10)}[kind]()
        m.load_state_dict(syn.random_state_dict(m, 23))
        m = m.to(DEV).train()
        adj = compute_adjs_distsim(a, sse, obs, gt, compact=True) if kind != "vrnn" else None
        opt = _ops.ClipAdam(list(m.parameters()), lr=1e-3, clip=10.0)
        training.attach_grad_arena(m, opt)

        def loss_fn(mm, e):
            if kind == "vrnn":
                k, n, _ = mm(rel, eps=e)
                c = torch.zeros_like(k)
            elif kind == "avrnn":
                k, n, _ = mm(rel, adj, seq_start_end=sse, eps=e)
                c = torch.zeros_like(k)
            else:
                k, n, c, _ = mm(traj, rel, goals, sse, adj, eps=e)
            return k + n + 1e-2 * c, (k, n, c)
        was = _ops.BR.enabled
        _ops.BR.enabled = streams
        try:
            if graphed:
                gfb = training.GraphedForwardBackward(m, loss_fn, [eps])
                loss, aux = gfb(eps)
            else:
                opt.zero_grad()
                loss, aux = loss_fn(m, eps)
                loss.backward()
                _ops.BR.join()
            torch.cuda.synchronize()
        finally:
            _ops.BR.enabled = was
        return [float(v.detach()) for v in aux], {k_: p.grad.detach().clone() for k_, p in m.named_parameters()}
    base_l, base_g = run(False, False)
    for streams, graphed in ((True, False), (True, True)):
        l, g = run(streams, graphed)
        for x, y in zip(l, base_l):
            assert abs(x - y) <= 1e-5 * max(1.0, abs(y)), (streams, graphed, l, base_l)
        for k_ in base_g:
            ref = base_g[k_]
            tol = 1e-4 * float(ref.abs().max()) + 1e-6
            assert float((g[k_] - ref).abs().max()) <= tol, (streams, graphed, k_)


@pytest.mark.parametrize("kind,kw", [("dagnet", dict(n_layers=2)), ("avrnn", dict(n_layers=1, z_dim=16)), ("vrnn", dict(n_layers=2))])
def test_fused_kernels_do_not_change_the_step(kind, kw):
    """The one-kernel forms of the GRU layer, of the dense blocks and of the blocks' backward (csrc/gru_fused.hip, mlp_fused.hip: taken
    from 2048 rows up, so the small golden cases above never reach them) against the layer-by-layer forms on the same 2560-row
    batch: the losses, every gradient and the final hidden state agree to fp32 summation-order noise."""
    import dagnet_amd
    from dagnet_amd import _ops, training, synthetic as syn
    from dagnet_amd.adjacency_matrix import compute_adjs_distsim
    a = syn.make_args(obs_len=4, pred_len=6, **kw)
    obs, gt, obs_rel, gt_rel, og, pg, sse = [t.to(DEV) for t in syn.make_batch([10] * 256, 4, 6, 29)]
    traj, rel = torch.cat((obs, gt), 0), torch.cat((obs_rel, gt_rel), 0)
    goals = torch.cat((og, pg), 0).to(torch.int32)
    eps = syn.uniform_noise(29, "eps", (rel.shape[0] - 1, rel.shape[1], a.z_dim)).to(DEV)

    def run(mask):
        m = {"vrnn": lambda: dagnet_amd.VRNN(a), "avrnn": lambda: dagnet_amd.AVRNN(a), "dagnet": lambda: dagnet_amd.DAGNet(a, 10)}[kind]()
        m.load_state_dict(syn.random_state_dict(m, 31))
        m = m.to(DEV).train()
        adj = compute_adjs_distsim(a, sse, obs, gt, compact=True) if kind != "vrnn" else None
        opt = _ops.ClipAdam(list(m.parameters()), lr=1e-3, clip=10.0)
        training.attach_grad_arena(m, opt)
        prev = _ops.set_fused(mask)
        try:
            opt.zero_grad()
            if kind == "vrnn":
                k, n, h = m(rel, eps=eps)
                c = torch.zeros_like(k)
            elif kind == "avrnn":
                k, n, h = m(rel, adj, seq_start_end=sse, eps=eps)
                c = torch.zeros_like(k)
            else:
                k, n, c, h = m(traj, rel, goals, sse, adj, eps=eps)
            (k + n + 1e-2 * c).backward()
            _ops.BR.join()
            torch.cuda.synchronize()
        finally:
            assert _ops.set_fused(prev) == mask
        return [float(v.detach()) for v in (k, n, c)], {k_: p.grad.detach().clone() for k_, p in m.named_parameters()}, h.detach().clone()
    pl, pg_, ph = run(0)
    gmax = max(float(v.abs().max()) for v in pg_.values())
    # mask 6: the dense blocks' kernels only - the same arithmetic in another summation order; mask 7 adds the GRU layer kernel, whose
    # gates use the hardware exp2 / rcp (1 ulp) instead of expf / tanhf: with random weights the 9-step recurrence through BatchNorm
    # amplifies that to ~1e-4 in h and ~1e-3 in a few gradients (measured; gradients that are analytically zero - out_att.a, whose
    # softmax is shift invariant - are noise of magnitude 1e-10 and fall under the absolute term)
    for mask, rtol in ((6, 1e-4), (7, 3e-3)):
        fl, fg, fh = run(mask)
        for x, y in zip(fl, pl):
            assert abs(x - y) <= 1e-5 * max(1.0, abs(y)), (mask, fl, pl)
        assert len(fg) > 10
        for name, ref in pg_.items():
            assert float((fg[name] - ref).abs().max()) <= rtol * float(ref.abs().max()) + 1e-7 * gmax, (mask, name)
        torch.testing.assert_close(fh, ph, rtol=10 * rtol, atol=10 * rtol)


def test_second_forward_before_backward_is_refused():
    """the step pool (dagnet_amd/_ops.StepPool) keeps ONE training step's activations; a second train-mode forward
    overwrites them, so the first graph's backward must raise instead of returning gradients of the wrong activations."""
    meta, _ = cases.load_golden("train_vrnn_tiny")
    _, model, _ = cases.build_model(meta)
    model = model.to(DEV).train()
    batch, eps = cases.build_train_inputs(meta)
    obs_rel, eps = batch[2].to(DEV), eps.to(DEV)
    kld1, nll1, _ = model(obs_rel, eps=eps)
    kld2, nll2, _ = model(obs_rel * 0.5, eps=eps)
    with pytest.raises(RuntimeError, match="one training step in flight"):
        (kld1 + nll1).backward()
    for p in model.parameters():
        p.grad = None
    (kld2 + nll2).backward()                        # the graph of the LAST forward is intact
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters())


BN_CASES = [n for n in cases.list_train_cases() if "vrnn" not in n or "avrnn" in n]      # the models with a BatchNorm (graph models)


@pytest.mark.parametrize("name", BN_CASES)
def test_sync_batchnorm_virtual_shards_match_the_whole_batch_golden(name):
    """Synchronised BatchNorm for data-parallel training: the batch is cut into two row shards that go through
    the staged kernels - moments per shard -> merge -> apply; backward sums per shard -> add -> apply (what the ranks do around
    their all-gather / all-reduce, _ops.SyncBnFn) - and the step must equal the reference's WHOLE-batch step: losses, every
    gradient, total norm, Adam step, running statistics, at the tolerances of the unsharded test."""
    from dagnet_amd import _ops
    meta, gold = cases.load_golden(name)
    _, model, _ = cases.build_model(meta)
    model = model.to(DEV)
    sizes = meta["sizes"]
    cut = int(sum(sizes[:max(1, len(sizes) // 2)]))
    prev = _ops.set_sync_batchnorm("virtual", [0, cut, int(sum(sizes))])
    try:
        out = hip_train_step(meta, model)
    finally:
        _ops.set_sync_batchnorm(*prev)
    check_step(out, gold, meta["lr"], meta["clip"])


@pytest.mark.parametrize("name", BN_CASES[:2])
def test_sync_batchnorm_dist_mode_in_one_process_matches_the_golden(name):
    """The "dist" form of synchronised BatchNorm (what `bench.py --sync-bn` and `loop.train(sync_batchnorm=True)` select) in a
    single process: the gathered moments are this process's own, so the step must equal the whole-batch golden.  The gathered
    buffer of every time step is a view of one stacked allocation that torch writes in place (collective / copy_): it must not be
    among the tensors autograd saves (a step with several time steps failed with "modified by an inplace operation")."""
    from dagnet_amd import _ops
    meta, gold = cases.load_golden(name)
    _, model, _ = cases.build_model(meta)
    model = model.to(DEV)
    prev = _ops.set_sync_batchnorm("dist")
    try:
        out = hip_train_step(meta, model)
    finally:
        _ops.set_sync_batchnorm(*prev)
    check_step(out, gold, meta["lr"], meta["clip"])


def test_per_shard_batchnorm_deviates_from_the_whole_batch_step():
    """What the default (per-rank BatchNorm statistics) costs under data parallelism: the same batch as two scene shards, each a
    forward / backward of its own with row_fraction weighting, gradients summed as the all-reduce would - against the reference's
    whole-batch gradients.  The deviation is real (far above the parity tolerance) and is recorded in DESIGN.md section 6."""
    from dagnet_amd import _ops
    from dagnet_amd.adjacency_matrix import compute_adjs_distsim
    from dagnet_amd.sharding import shard_batch
    from dagnet_amd.utils import to_goals_one_hot
    name = "train_dagnet_bsk_gat_n5"
    meta, gold = cases.load_golden(name)
    args, model, _ = cases.build_model(meta)
    model = model.to(DEV).train()
    batch, eps = cases.build_train_inputs(meta)
    sizes = meta["sizes"]
    N = int(sum(sizes))
    total = {k: torch.zeros_like(p) for k, p in model.named_parameters()}
    bn_before = {k: v.clone() for k, v in model.state_dict().items() if "running" in k}
    for r in range(2):
        (obs, gt, obs_rel, gt_rel, og, pg, sse), local = shard_batch(batch, sizes, 2, r)
        lo = int(sum(sizes[:0 if r == 0 else len(sizes) - len(local)]))
        n_loc = int(local.sum())
        e = eps[:, lo:lo + n_loc].to(DEV)
        obs, gt, obs_rel, gt_rel, og, pg = [t.to(DEV) for t in (obs, gt, obs_rel, gt_rel, og, pg)]
        traj, rel = torch.cat((obs, gt), 0), torch.cat((obs_rel, gt_rel), 0)
        adj = compute_adjs_distsim(args, sse, obs, gt)
        g_ohe = to_goals_one_hot(torch.cat((og, pg), 0), args.g_dim)
        for p in model.parameters():
            p.grad = None
        kld, nll, ce, _ = model(traj, rel, g_ohe, sse, adj, eps=e)
        loss = (n_loc / N) * (meta["kld_weight"] * kld + nll) + meta["ce_weight"] * ce
        loss.backward()
        for k, p in model.named_parameters():
            if p.grad is not None:
                total[k] += p.grad
        for k, v in bn_before.items():                      # each "rank" starts from the same running statistics
            model.state_dict()[k].copy_(v)
    worst, worst_key = 0.0, None
    for key, want in gold.items():
        if key.startswith("grad/"):
            got = total[key[5:]].cpu().numpy()
            dev = float(np.abs(got - want).max() / (np.abs(want).max() + 1e-12))
            if np.abs(want).max() > 1e-3 * float(gold["total_norm"]) and dev > worst:
                worst, worst_key = dev, key[5:]
    print(f"\nper-shard BatchNorm, 2 shards of {name}: worst gradient deviation {worst:.3f} of the tensor's max |g| ({worst_key})")
    assert worst > 10 * cases.GRAD_RTOL, "per-shard statistics should differ visibly from whole-batch statistics on 15 rows"
