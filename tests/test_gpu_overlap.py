"""GPU: K separate sample() calls - the reference's evaluation loop as written (models/dagnet/evaluate_bsk.py:113-114) - go to
side streams and overlap on the device (_capi._overlap_stream).  The results must be what one call after the other on the
caller's stream gives, whatever the caller does with its tensors in between."""
import os

import numpy as np
import pytest
import torch

import dagnet_amd
from dagnet_amd import _capi, synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(kind, n_scenes=400, n_layers=2):
    a = syn.make_args(n_layers=n_layers)
    m = {"vrnn": lambda: dagnet_amd.VRNN(a), "avrnn": lambda: dagnet_amd.AVRNN(a), "dagnet": lambda: dagnet_amd.DAGNet(a, 10)}[kind]()
    m.load_state_dict(syn.random_state_dict(m, 7))
    m = m.to(DEV).eval()
    sizes = [10] * n_scenes
    obs, gt, obs_rel, gt_rel, og, pg, sse = [t.to(DEV) for t in syn.make_batch(sizes, 10, 12, 3)]
    h = (syn.uniform_noise(3, "h0", (a.n_layers, 10 * n_scenes, 64)) - 0.5).to(DEV)
    return a, m, h, obs[-1].clone(), og[-1].to(torch.int32), sse


def _call(kind, m, T, h, x0, g0, sse, seed):
    if kind == "vrnn":
        return m.sample(T, h, seed=seed)
    if kind == "avrnn":
        return m.sample(T, h, x0, sse, seed=seed)
    return m.sample(T, h, x0, g0, sse, seed=seed)


class no_overlap:
    def __enter__(self):
        os.environ["DAGNET_NO_OVERLAP"] = "1"

    def __exit__(self, *a):
        del os.environ["DAGNET_NO_OVERLAP"]


@pytest.mark.parametrize("kind", ["vrnn", "avrnn", "dagnet"])
def test_loop_of_sample_calls_equals_serial_launches(kind):
    """20 calls from a loop, each output consumed on the caller's stream and dropped (so that the allocator recycles the blocks
    while later launches are in flight): bitwise what the same calls give on the caller's stream alone."""
    a, m, h, x0, g0, sse = _setup(kind)
    T, K = 12, 20
    with no_overlap():
        ref = [_call(kind, m, T, h, x0, g0, sse, 100 + k).clone() for k in range(K)]
    torch.cuda.synchronize()
    assert "_ovl" not in m.hip_model(DEV).__dict__
    sums, keep = [], []
    for k in range(K):
        out = _call(kind, m, T, h, x0, g0, sse, 100 + k)
        sums.append(out.double().sum())             # queued on the caller's stream right behind the call
        keep.append(out.clone() if k % 3 == 0 else None)
        del out
    assert m.hip_model(DEV).__dict__["_ovl"]["n"] == K, "the launches did not go through the side streams"
    for k in range(K):
        assert sums[k].item() == ref[k].double().sum().item()
        if keep[k] is not None:
            assert torch.equal(keep[k], ref[k])


def test_inputs_written_between_calls_are_seen():
    """In-place writes of the caller (torch bumps the version counter), new tensors and a host-side seq_start_end list: every call
    starts from the inputs as they are when it is made."""
    kind = "dagnet"
    a, m, h, x0, g0, sse = _setup(kind, n_scenes=300)
    T = 8
    h2 = h * 0.5
    x2 = x0 + 1.0
    with no_overlap():
        r_a = _call(kind, m, T, h, x0, g0, sse, 5).clone()
        r_b = _call(kind, m, T, h2, x0, g0, sse, 5).clone()
        r_c = _call(kind, m, T, h2, x2, g0, sse, 5).clone()
    hh, xx = h.clone(), x0.clone()
    o_a = _call(kind, m, T, hh, xx, g0, sse, 5)
    hh.mul_(0.5)                                     # in place, behind the first launch on the caller's stream
    o_b = _call(kind, m, T, hh, xx, g0, sse, 5)
    xx.add_(1.0)
    o_c = _call(kind, m, T, hh, xx, g0, sse.cpu().tolist(), 5)
    o_a2 = _call(kind, m, T, h.clone(), x0.clone(), g0, sse, 5)
    assert torch.equal(o_a, r_a) and torch.equal(o_b, r_b) and torch.equal(o_c, r_c) and torch.equal(o_a2, r_a)
    assert not torch.equal(r_a, r_b) and not torch.equal(r_b, r_c)


def test_one_hot_goals_and_other_streams():
    """The reference passes one-hot goals (a new index tensor per call unless remembered) and callers may run on a stream of their
    own; large launches and graph-capturing streams stay on the caller's stream."""
    kind = "dagnet"
    a, m, h, x0, g0, sse = _setup(kind, n_scenes=200)
    T = 6
    ohe = torch.nn.functional.one_hot(g0.long(), a.g_dim).float()
    with no_overlap():
        ref = [m.sample(T, h, x0, g0, sse, seed=k).clone() for k in range(4)]
    s = torch.cuda.Stream(DEV)
    s.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(s):
        outs = [m.sample(T, h, x0, ohe, sse, seed=k) for k in range(4)]
        tot = [o.sum() for o in outs]
    torch.cuda.current_stream(DEV).wait_stream(s)
    for k in range(4):
        assert torch.equal(outs[k], ref[k]) and tot[k].item() == ref[k].sum().item()
    # a launch of many rounds is not diverted
    n0 = m.hip_model(DEV).__dict__["_ovl"]["n"]
    m.sample_k(400, T, h, x0, g0, sse, seed=1)
    assert m.hip_model(DEV).__dict__["_ovl"]["n"] == n0
