"""GPU: the two documented hazards of the drop-in are guarded (csrc/api.hip "Hazard guards").

1. NaN in the state a call starts from.  The register-resident kernels are compiled without NaN bookkeeping and the others clamp
   with fmaxf, so a NaN could be laundered into an ordinary number; the reference (restated by the oracle, which runs stock torch
   arithmetic) gives NaN for the row and, with a graph, for the whole batch from the next step on.  The outputs' NaN masks must
   equal the oracle's, and rows the oracle keeps finite must equal a clean run bit for bit.
2. Two calls on one plan whose workspace they both write, in flight on different streams: the second is refused."""
import contextlib
import os

import pytest
import torch

import dagnet_amd
from dagnet_amd import _capi, synthetic as syn
from dagnet_amd.adjacency_matrix import compute_adjs_distsim
from oracle import dagnet_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SIZES = [10, 7, 3, 10, 1, 9]


def _model(kind, **kw):
    a = syn.make_args(**kw)
    m = {"vrnn": lambda: dagnet_amd.VRNN(a), "avrnn": lambda: dagnet_amd.AVRNN(a), "dagnet": lambda: dagnet_amd.DAGNet(a, 10)}[kind]()
    sd = syn.random_state_dict(m, 7)
    m.load_state_dict(sd)
    return a, m.to(DEV).eval(), sd


@contextlib.contextmanager
def _generic(on):
    if on:
        os.environ["DAGNET_FORCE_GENERIC"] = "1"
    try:
        yield
    finally:
        os.environ.pop("DAGNET_FORCE_GENERIC", None)


def _sample(kind, m, K, T, h, obs, og, sse, eps, q):
    if kind == "vrnn":
        return m.sample_k(K, T, h, eps=eps)
    if kind == "avrnn":
        return m.sample_k(K, T, h, obs[-1], sse, eps=eps)
    return m.sample_k(K, T, h, obs[-1], og[-1].to(torch.int32), sse, eps=eps, q=q)


@pytest.mark.parametrize("kind", ["vrnn", "avrnn", "dagnet"])
@pytest.mark.parametrize("generic", [False, True])
def test_nan_in_start_state_propagates_like_the_reference(kind, generic):
    kw = dict(vrnn=dict(n_layers=2), avrnn=dict(n_layers=1), dagnet=dict(n_layers=2))[kind]
    a, m, sd = _model(kind, **kw)
    K, T = 2, 4
    obs, gt, obs_rel, gt_rel, og, pg, sse = syn.make_batch(SIZES, 8, T, 3)
    N = obs.shape[1]
    h = syn.uniform_noise(3, "h0", (a.n_layers, N, 64)) - 0.5
    eps = syn.uniform_noise(3, "eps", (K, T, N, a.z_dim))
    q = syn.exp1_noise(3, "q", (K, T, N, a.g_dim))
    h_bad, obs_bad = h.clone(), obs
    h_bad[a.n_layers - 1, 12, 5] = float("nan")
    cfg = orc.make_cfg(n_max_agents=10, n_layers=a.n_layers)
    want = []
    with torch.no_grad():
        for k in range(K):
            if kind == "vrnn":
                want.append(orc.vrnn_sample(sd, cfg, T, h_bad, eps[k]))
            elif kind == "avrnn":
                want.append(orc.avrnn_sample(sd, cfg, T, h_bad, obs_bad[-1], sse, eps[k]))
            else:
                want.append(orc.dagnet_sample(sd, cfg, T, h_bad, obs_bad[-1], orc.one_hot(og[-1], a.g_dim), sse, eps[k], q[k]))
    want_nan = torch.isnan(torch.stack(want))
    d = lambda t: t.to(DEV)  # noqa: E731
    with _generic(generic):
        clean = _sample(kind, m, K, T, d(h), d(obs), d(og), d(sse), d(eps), d(q)).cpu()
        got = _sample(kind, m, K, T, d(h_bad), d(obs_bad), d(og), d(sse), d(eps), d(q)).cpu()
    assert torch.isfinite(clean).all()
    assert want_nan.any()
    assert torch.equal(torch.isnan(got), want_nan)
    assert torch.equal(got[~want_nan], clean[~want_nan])


@pytest.mark.parametrize("kind", ["vrnn", "avrnn", "dagnet"])
@pytest.mark.parametrize("generic", [False, True])
def test_nan_in_observed_trajectory_poisons_the_losses(kind, generic):
    kw = dict(vrnn=dict(n_layers=2), avrnn=dict(n_layers=1), dagnet=dict(n_layers=2))[kind]
    a, m, sd = _model(kind, **kw)
    obs, gt, obs_rel, gt_rel, og, pg, sse = [t.to(DEV) for t in syn.make_batch(SIZES, 8, 4, 3)]
    adj = compute_adjs_distsim(a, sse, obs, gt) if kind != "vrnn" else None

    def run(rel):
        if kind == "vrnn":
            return m(rel, seed=4)
        if kind == "avrnn":
            return m(rel, adj, seq_start_end=sse, seed=4)
        return m(obs, rel, og.to(torch.int32), sse, adj, seed=4)
    with _generic(generic):
        clean = run(obs_rel)
        bad_rel = obs_rel.clone()
        bad_rel[2, 12, 0] = float("nan")
        bad = run(bad_rel)
    assert all(bool(torch.isfinite(torch.as_tensor(v)).all()) for v in clean)
    assert torch.isnan(torch.as_tensor(bad[0])) and torch.isnan(torch.as_tensor(bad[1]))
    hb, hc = bad[-1], clean[-1]
    if kind == "vrnn":          # independent rows: only row 12's state is lost
        rows = torch.isnan(hb).any(dim=2).any(dim=0)
        assert rows.nonzero().flatten().tolist() == [12]
        keep = torch.ones(hb.shape[1], dtype=torch.bool, device=hb.device)
        keep[12] = False
        assert torch.equal(hb[:, keep], hc[:, keep])
    else:                       # dense block-diagonal graph products spread it over the batch
        assert torch.isnan(hb).all()


def test_calls_on_one_plan_from_two_streams_are_serialised():
    """dagnet_forward sums its losses in buffers of the plan.  A call that arrives on another stream while the previous one is
    still running is ordered behind it on the device (hipStreamWaitEvent on its completion event): both return the right
    losses, whatever the timing, and so does a program that had ordered the streams itself."""
    a, m, _ = _model("dagnet", n_layers=2)
    sizes = [10] * 2048
    obs, gt, obs_rel, gt_rel, og, pg, sse = [t.to(DEV) for t in syn.make_batch(sizes, 40, 2, 5)]
    adj = compute_adjs_distsim(a, sse, obs, gt, compact=True)
    run = lambda: m(obs, obs_rel, og.to(torch.int32), sse, adj, seed=4)  # noqa: E731
    first = run()
    second = run()                                   # same stream: ordered
    assert float(first[0]) == float(second[0])
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    with torch.cuda.stream(s1):
        for _ in range(4):
            outs.append(run())                       # several milliseconds of work queued on s1
    with torch.cuda.stream(s2):
        outs.append(run())                           # arrives while s1 is busy with the same plan's buffers
        s2.wait_stream(s1)                           # a program that orders the streams itself is not refused either
        outs.append(run())
    torch.cuda.synchronize()
    for o in outs:
        assert [float(v) for v in o[:3]] == [float(v) for v in first[:3]]


@pytest.mark.parametrize("kind", ["vrnn", "dagnet"])
def test_sample_replays_from_a_captured_hip_graph(kind):
    """sample_k inside torch.cuda.graph: the rollout kernel, the NaN guard and (on workspace-owning plans) the in-flight
    bookkeeping are all capturable - no allocation, no event query on a capturing stream; the replay reproduces the eager call
    bit for bit and follows new inputs copied into the static tensors."""
    kw = dict(vrnn=dict(n_layers=2), dagnet=dict(n_layers=2))[kind]
    a, m, _ = _model(kind, **kw)
    K, T = 3, 8
    obs, gt, obs_rel, gt_rel, og, pg, sse = [t.to(DEV) for t in syn.make_batch(SIZES, 8, T, 3)]
    N = obs.shape[1]
    h = (syn.uniform_noise(3, "h0", (a.n_layers, N, 64)) - 0.5).to(DEV)
    eps = syn.uniform_noise(3, "eps", (K, T, N, a.z_dim)).to(DEV)
    q = syn.exp1_noise(3, "q", (K, T, N, a.g_dim)).to(DEV)
    if kind == "dagnet":
        g = torch.cuda.CUDAGraph()
        with pytest.raises(ValueError, match="host tensor"), torch.cuda.graph(g):
            _sample(kind, m, K, T, h, obs, og, sse, eps, q)              # the scene list must not need a blocking copy
        del g
    sse = sse.cpu()
    eager = _sample(kind, m, K, T, h, obs, og, sse, eps, q)
    torch.cuda.synchronize()
    h_static = h.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _sample(kind, m, K, T, h_static, obs, og, sse, eps, q)          # plan creation happens outside the capture
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = _sample(kind, m, K, T, h_static, obs, og, sse, eps, q)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    h2 = h * 0.5
    h_static.copy_(h2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, _sample(kind, m, K, T, h2, obs, og, sse, eps, q))


def test_forward_replays_from_a_captured_hip_graph():
    """The teacher-forced forward (eval mode) captures too: losses come back as device scalars, the seed is a launch argument
    (so a replay repeats its noise - by design of a graph)."""
    a, m, _ = _model("dagnet", n_layers=2)
    obs, gt, obs_rel, gt_rel, og, pg, sse = [t.to(DEV) for t in syn.make_batch(SIZES, 8, 4, 3)]
    sse = sse.cpu()
    adj = compute_adjs_distsim(a, sse, obs, gt, compact=True)
    rel = obs_rel.clone()
    run = lambda: m(obs, rel, og.to(torch.int32), sse, adj, seed=4)  # noqa: E731
    eager = run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run()
    g.replay()
    torch.cuda.synchronize()
    for x, y in zip(out, eager):
        assert torch.equal(torch.as_tensor(x), torch.as_tensor(y))
    rel.mul_(0.5)
    g.replay()
    torch.cuda.synchronize()
    for x, y in zip(out, run()):
        assert torch.equal(torch.as_tensor(x), torch.as_tensor(y))
