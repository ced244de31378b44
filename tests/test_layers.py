"""Rows D1 / D3 / E3: the stand-alone layers GraphAttentionLayer.forward (gat_layers.py:25-40) and GraphConvolution.forward
(gcn_layers.py:20-26) and block_diag_irregular (models/utils/utils.py:290-302) against tests/golden/layers.npz, which holds
the outputs AND gradients of the reference's own classes (oracle/make_golden.py layers).
CPU: the oracle's layer functions and the (pure layout) block-diagonal assembly; GPU: the HIP layers through the C ABI."""
import json
import os

import numpy as np
import pytest
import torch

from dagnet_amd import synthetic as syn
from dagnet_amd.utils import block_diag_irregular
from oracle import dagnet_oracle as orc
from tests import cases

SPECS = (("gat_concat", "gat", 64, 8, True), ("gat_out", "gat", 32, 64, False), ("gat_wide", "gat", 32, 90, False),
         ("gcn", "gcn", 64, 8, True), ("gcn_nobias", "gcn", 8, 64, False))


def _load():
    z = np.load(os.path.join(cases.GOLDEN, "layers.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return meta, {k: z[k] for k in z.files if k != "meta"}


def _inputs(meta, tag, nin, nout):
    n, seed = int(sum(meta["sizes"])), meta["seed"]
    x = syn.uniform_noise(seed, "layer_x_" + tag, (n, nin)) * 2 - 1
    probe = syn.uniform_noise(seed, "layer_probe_" + tag, (n, nout)) * 2 - 1
    assert abs(cases._sum(x) + cases._sum(probe) - meta["checks"][tag]) < 1e-6, "input generator drifted"
    pos = syn.uniform_noise(seed, "layer_pos", (n, 2)) * 10
    assert abs(cases._sum(pos) - meta["checks"]["pos"]) < 1e-6
    return x, probe, pos, syn.seq_start_end(meta["sizes"])


def _make(kind, nin, nout, flag, seed):
    from dagnet_amd.gat import GraphAttentionLayer
    from dagnet_amd.gcn import GraphConvolution
    mod = GraphAttentionLayer(nin, nout, 0.2, concat=flag) if kind == "gat" else GraphConvolution(nin, nout, bias=flag)
    mod.load_state_dict(syn.random_state_dict(mod, seed))
    return mod


def test_block_diag_irregular_is_bit_exact():
    _, gold = _load()
    for tag, count in (("bd2", 4), ("bd3", 3)):
        blocks = [torch.from_numpy(gold[f"{tag}_in_{i}"]) for i in range(count)]         # fp64 blocks, as scipy hands them over
        got = block_diag_irregular(blocks)
        assert got.dtype == torch.float32 and tuple(got.shape) == gold[f"{tag}_out"].shape
        assert np.array_equal(got.numpy(), gold[f"{tag}_out"])
        got_np = block_diag_irregular([b.numpy() for b in blocks])                        # numpy blocks work too
        assert np.array_equal(got_np.numpy(), gold[f"{tag}_out"])
    # the oracle's assembly (2-D) agrees bit for bit as well
    blocks = [torch.from_numpy(gold[f"bd2_in_{i}"]) for i in range(4)]
    assert np.array_equal(orc.block_diag_dense(blocks).numpy(), gold["bd2_out"])


@pytest.mark.parametrize("tag,kind,nin,nout,flag", SPECS)
def test_oracle_layers_match_reference(tag, kind, nin, nout, flag):
    meta, gold = _load()
    x, probe, pos, sse = _inputs(meta, tag, nin, nout)
    mod = _make(kind, nin, nout, flag, meta["seed"])
    for at, adj in ((1, orc.adj_distance_sim(1.2, sse, pos)), (2, orc.adj_knn_sim(3, sse, pos))):
        with torch.no_grad():
            if kind == "gat":
                y = orc.gat_layer_forward(mod.W, mod.a, x, adj, 0.2, flag)
            else:
                y = orc.gcn_layer_forward(mod.weight, mod.bias, x, adj)
        np.testing.assert_allclose(y.numpy(), gold[f"{tag}_adj{at}_out"], atol=2e-6, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("tag,kind,nin,nout,flag", SPECS)
def test_hip_layers_match_reference(tag, kind, nin, nout, flag, compact):
    """forward and the gradients w.r.t. the input and the layer's parameters, dense [N,N] adjacency (the reference's call)
    and the compact per-scene blocks."""
    from dagnet_amd.adjacency_matrix import adjs_distance_sim_pred, adjs_knn_sim_pred
    dev = "cuda:0"
    meta, gold = _load()
    x, probe, pos, sse = _inputs(meta, tag, nin, nout)
    mod = _make(kind, nin, nout, flag, meta["seed"]).to(dev)
    for at in (1, 2):
        if at == 1:
            adj = adjs_distance_sim_pred(1.2, sse, pos.to(dev), compact=compact)
        else:
            adj = adjs_knn_sim_pred(3, sse, pos.to(dev), compact=compact)
        xi = x.to(dev).requires_grad_(True)
        mod.zero_grad()
        y = mod(xi, adj)
        (y * probe.to(dev)).sum().backward()
        want = gold[f"{tag}_adj{at}_out"]
        np.testing.assert_allclose(y.detach().cpu().numpy(), want, atol=3e-6 + 1e-5 * np.abs(want).max(), rtol=1e-5)
        grads = {"x": xi.grad}
        grads.update({k: p.grad for k, p in mod.named_parameters()})
        for k, g in grads.items():
            w = gold[f"{tag}_adj{at}_d{k}"]
            tol = 2e-4 * np.abs(w).max() + 2e-6
            assert np.abs(g.cpu().numpy().reshape(w.shape) - w).max() <= tol, f"{tag} adj{at} d{k}"


@pytest.mark.gpu
def test_hip_layers_refuse_cpu_tensors():
    from dagnet_amd import _capi
    mod = _make("gat", 8, 4, True, 1)
    with pytest.raises(_capi.DagnetHipError):
        mod(torch.zeros(3, 8), torch.ones(3, 3))
