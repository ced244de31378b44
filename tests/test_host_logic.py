"""CPU: host-side logic, the drop-in contract (state-dict keys, signatures, error behaviour) and the
C-ABI library's exported symbols.  No compute entry point is called here (there is no GPU)."""
import ctypes
import inspect
import json
import os
import re

import numpy as np
import pytest
import torch

import dagnet_amd
from dagnet_amd import _capi, synthetic as syn
from dagnet_amd._runtime import infer_scene_ptr_from_adj, scene_ptr_from_sse
from dagnet_amd.sharding import shard_batch, shard_scenes
from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "dagnet_hip.h")).read()
    declared = set(re.findall(r"\b(dagnet_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"dagnet_status", "dagnet_config", "dagnet_tensor", "dagnet_model", "dagnet_plan"}
    from dagnet_amd._ops import OP_EXPORTS
    assert declared == set(_capi.EXPORTS) | set(OP_EXPORTS), declared ^ (set(_capi.EXPORTS) | set(OP_EXPORTS))
    lib = ctypes.CDLL(_capi.lib_path())
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} is declared in include/dagnet_hip.h but not exported"
    lib.dagnet_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.dagnet_version()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "_LIB_PATH", "/nonexistent/libdagnet_hip.so")
    with pytest.raises(_capi.DagnetHipError, match="no CPU fallback"):
        _capi.load()


def test_cpu_tensors_are_rejected_not_computed():
    a = syn.make_args()
    m = dagnet_amd.VRNN(a).eval()
    with pytest.raises(_capi.DagnetHipError, match="no CPU fallback"):
        m.sample(3, torch.zeros(a.n_layers, 4, a.rnn_dim))
    g = dagnet_amd.GAT(8, 4, 8, 0.2, 2).eval()
    with pytest.raises(_capi.DagnetHipError):
        g(torch.zeros(3, 8), torch.ones(3, 3))


def test_train_mode_batchnorm_is_refused():
    m = dagnet_amd.AVRNN(syn.make_args(n_layers=1))
    assert m.training
    with pytest.raises(NotImplementedError, match="BatchNorm"):
        m.sample(2, torch.zeros(1, 4, 64), torch.zeros(4, 2), torch.tensor([[0, 4]]))


@pytest.mark.parametrize("tag", ["real_gat", "real_gcn"])
def test_state_dict_keys_match_the_reference(tag):
    want = json.load(open(os.path.join(cases.GOLDEN, "state_dict_layouts.json")))
    kw = dict(h_dim=64, z_dim=32, rnn_dim=64, g_dim=90)
    kw.update(dict(n_layers=2) if tag == "real_gat" else dict(n_layers=1, graph_model="gcn"))
    a = syn.make_args(**kw)
    for name, mod in (("vrnn", dagnet_amd.VRNN(a)), ("avrnn", dagnet_amd.AVRNN(a)), ("dagnet", dagnet_amd.DAGNet(a, 10))):
        got = {k: list(v.shape) for k, v in mod.state_dict().items()}
        assert got == want[f"{name}_{tag}"], f"{name}: state-dict layout differs from the reference"


def test_signatures_match_the_reference():
    sig = lambda f: [p for p in inspect.signature(f).parameters][1:]
    assert sig(dagnet_amd.VRNN.forward)[:1] == ["x"]
    assert sig(dagnet_amd.VRNN.sample)[:2] == ["samples_seq_len", "h"]
    assert sig(dagnet_amd.AVRNN.forward)[:2] == ["x", "adj"]
    assert sig(dagnet_amd.AVRNN.sample)[:4] == ["samples_seq_len", "h", "x_abs_start", "seq_start_end"]
    assert sig(dagnet_amd.DAGNet.forward)[:5] == ["traj", "traj_rel", "goals_ohe", "seq_start_end", "adj_out"]
    assert sig(dagnet_amd.DAGNet.sample)[:5] == ["samples_seq_len", "h", "x_abs_start", "g_start", "seq_start_end"]
    assert sig(dagnet_amd.GAT.__init__) == ["nin", "nhid", "nout", "alpha", "nheads"]
    assert sig(dagnet_amd.GCN.__init__) == ["nin", "nhid", "nout"]
    assert sig(dagnet_amd.GAT.forward)[:2] == ["x", "adj"]


def test_dagnet_constructor_guards():
    with pytest.raises(Exception, match="top_k_neigh"):
        dagnet_amd.DAGNet(syn.make_args(adjacency_type=2, top_k_neigh=None), 5)
    with pytest.raises(AssertionError):
        dagnet_amd.DAGNet(syn.make_args(n_heads=None), 5)


def test_scene_ptr_from_sse():
    assert scene_ptr_from_sse(torch.tensor([[0, 2], [2, 7], [7, 10]])).tolist() == [0, 2, 7, 10]
    for bad in ([[1, 3]], [[0, 2], [3, 5]], [[0, 2], [2, 2]], []):
        with pytest.raises(ValueError):
            scene_ptr_from_sse(np.asarray(bad).reshape(-1, 2))
    with pytest.raises(ValueError):
        scene_ptr_from_sse([[0, 4]], n_rows=5)


def test_infer_scene_ptr_from_adjacency_support():
    from oracle import dagnet_oracle as orc
    sse = syn.seq_start_end([2, 5, 1, 3])
    pos = torch.rand(11, 2) * 10
    for adj in (orc.adj_fully_connected(sse), orc.adj_distance_sim(1.2, sse, pos), orc.adj_knn_sim(2, sse, pos)):
        got = infer_scene_ptr_from_adj(adj)
        if adj is not None and (adj > 0).sum() == sum(n * n for n in (2, 5, 1, 3)):
            assert got.tolist() == [0, 2, 7, 8, 11]
        else:       # kNN may split a scene into finer blocks, never merge two scenes
            assert set(got.tolist()) >= {0, 2, 7, 8, 11}
    t = torch.stack([orc.adj_fully_connected(sse)] * 3)
    assert infer_scene_ptr_from_adj(t).tolist() == [0, 2, 7, 8, 11]


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shard_scenes_partitions_and_balances(world):
    sizes = syn.ragged_sizes(101, seed=3)
    seen = []
    loads = []
    for r in range(world):
        lo, hi = shard_scenes(sizes, world, r)
        seen += list(range(lo, hi))
        loads.append(int(sizes[lo:hi].sum()))
    assert seen == list(range(len(sizes)))
    assert max(loads) - min(loads) <= 2 * int(sizes.max())


def test_shard_batch_layout():
    sizes = [2, 5, 3, 4]
    batch = syn.make_batch(sizes, 3, 4, 0)
    (obs, gt, obs_rel, gt_rel, og, pg, sse), local = shard_batch(batch, sizes, 2, 1)
    assert local.tolist() == [3, 4] and obs.shape == (3, 7, 2) and gt.shape == (4, 7, 2) and og.shape == (3, 7)
    assert sse.tolist() == [[0, 3], [3, 7]]
    assert torch.equal(obs, batch[0][:, 7:14])


def test_synthetic_batch_layout_matches_seq_collate():
    obs, gt, obs_rel, gt_rel, og, pg, sse = syn.make_batch([10, 10], 10, 40, 1)
    assert obs.shape == (10, 20, 2) and gt.shape == (40, 20, 2) and og.shape == (10, 20) and pg.shape == (40, 20)
    assert obs.dtype == torch.float32 and sse.dtype == torch.int64 and sse.tolist() == [[0, 10], [10, 20]]
    full = torch.cat([obs, gt])
    rel = torch.cat([obs_rel, gt_rel])
    assert torch.all(rel[0] == 0)                                 # rel[0] = 0, rel[t] = abs[t] - abs[t-1]
    assert torch.allclose(rel[1:], full[1:] - full[:-1], atol=1e-6)
    assert og.min() >= 0 and og.max() < 90 and torch.all(og == og.round())


def test_random_state_dict_is_order_independent():
    a = syn.make_args()
    m = dagnet_amd.DAGNet(a, 5)
    sd = syn.random_state_dict(m, 9)
    shapes = {k: tuple(v.shape) for k, v in reversed(list(m.state_dict().items()))}
    sd2 = syn.random_state_dict(shapes, 9)
    assert all(torch.equal(sd[k], sd2[k]) for k in sd)


def test_train_mode_forward_rejects_cpu_tensors():
    """model.train() routes forward() to the HIP training operators; there is no CPU path behind them."""
    a = syn.make_args()
    m = dagnet_amd.VRNN(a).train()
    with pytest.raises(_capi.DagnetHipError, match="no CPU path"):
        m(torch.zeros(4, 3, 2))


def test_gradient_arena_is_one_flat_buffer():
    from dagnet_amd import _ops, training
    a = syn.make_args(h_dim=16, z_dim=8, rnn_dim=16, n_layers=1)
    m = dagnet_amd.AVRNN(a)
    arena = training.attach_grad_arena(m)
    params = list(m.parameters())
    assert arena.numel() == sum(p.numel() for p in params) and float(arena.abs().sum()) == 0.0
    off = 0
    for p in params:
        assert p.grad.shape == p.shape and p.grad.is_contiguous() and p.grad.data_ptr() == arena.data_ptr() + 4 * off
        assert getattr(p, "_hip_sink") and isinstance(p._hip_defer, _ops.DeferredGrads)
        off += p.numel()
    arena.fill_(1.0)
    assert all(float(p.grad.min()) == 1.0 for p in params)
