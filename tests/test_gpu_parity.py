"""GPU: the HIP path (through the C ABI, via the drop-in modules) against the golden outputs of
the reference and against the CPU oracle on larger seeded inputs."""
import numpy as np
import pytest
import torch

from tests import cases
from tests.cases import DISP_ATOL, DISP_RTOL, H_ATOL, LOSS_RTOL

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, want, atol, rtol, what):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = np.abs(got - want) - (atol + rtol * np.abs(want))
    assert err.max() <= 0, f"{what}: max abs diff {np.abs(got - want).max():.3e} (atol {atol}, rtol {rtol})"


def _run_forward(kind, model, args, batch, eps_f):
    from dagnet_amd.adjacency_matrix import compute_adjs, compute_adjs_distsim, compute_adjs_knnsim
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = [t.to(DEV) for t in batch]
    eps_f = eps_f.to(DEV)
    if kind == "vrnn":
        kld, nll, h = model(obs_rel, eps=eps_f)
        return kld, nll, torch.zeros(1), h
    if args.adjacency_type == 0:
        adj = compute_adjs(args, sse, device=DEV)
    elif args.adjacency_type == 1:
        adj = compute_adjs_distsim(args, sse, obs, gt)
    else:
        adj = compute_adjs_knnsim(args, sse, obs, gt)
    if kind == "avrnn":
        kld, nll, h = model(obs_rel, adj, eps=eps_f)       # scenes inferred from the dense adjacency
        return kld, nll, torch.zeros(1), h
    from dagnet_amd.utils import to_goals_one_hot
    g_ohe = to_goals_one_hot(obs_goals, args.g_dim)
    return model(obs, obs_rel, g_ohe, sse, adj, eps=eps_f)


@pytest.mark.parametrize("name", cases.list_cases())
def test_forward_matches_reference(name):
    meta, gold = cases.load_golden(name)
    args, model, _ = cases.build_model(meta)
    model = model.to(DEV)
    batch, eps_f, _, _ = cases.build_inputs(meta)
    kld, nll, ce, h = _run_forward(meta["kind"], model, args, batch, eps_f)
    assert kld.shape == (1,) and nll.shape == (1,) and h.shape == gold["fwd_h"].shape
    _close(h.cpu(), gold["fwd_h"], H_ATOL, 0, "h")
    _close(kld.cpu(), gold["fwd_kld"], 1e-5, LOSS_RTOL, "KLD")
    _close(nll.cpu(), gold["fwd_nll"], 1e-5, LOSS_RTOL, "NLL")
    _close(ce.cpu(), gold["fwd_ce"], 1e-5, LOSS_RTOL, "CE")


@pytest.mark.parametrize("name", cases.list_cases())
def test_sample_matches_reference(name):
    meta, gold = cases.load_golden(name)
    args, model, _ = cases.build_model(meta)
    model = model.to(DEV)
    (obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse), _, eps_s, q_s = cases.build_inputs(meta)
    h = torch.from_numpy(gold["fwd_h"]).to(DEV)
    K, T, kind = meta["K"], meta["pred_len"], meta["kind"]
    eps_s = eps_s.to(DEV)
    k1 = K - 1
    if kind == "vrnn":
        s = model.sample_k(K, T, h, eps=eps_s)
        s1 = model.sample(T, h, eps=eps_s[k1])
    elif kind == "avrnn":
        s = model.sample_k(K, T, h, obs[-1].to(DEV), sse, eps=eps_s)
        s1 = model.sample(T, h, obs[-1].to(DEV), sse, eps=eps_s[k1])
    else:
        g0 = obs_goals[-1].to(DEV).to(torch.int32)
        s, goals = model.sample_k(K, T, h, obs[-1].to(DEV), g0, sse, eps=eps_s, q=q_s.to(DEV), return_goals=True)
        assert gold["margins"].min() > cases.MARGIN_MIN
        assert np.array_equal(goals.cpu().numpy(), gold["goals"]), "sampled goals differ from the reference"
        from dagnet_amd.utils import to_goals_one_hot
        s1 = model.sample(T, h, obs[-1].to(DEV), to_goals_one_hot(obs_goals, args.g_dim)[-1].to(DEV), sse,
                          eps=eps_s[k1], q=q_s[k1].to(DEV))
    assert s.shape == gold["samples"].shape
    _close(s.cpu(), gold["samples"], DISP_ATOL, DISP_RTOL, "samples")
    # the reference-signature single-sample call is the K=1 slice of the batched launch (a scene may sit
    # at another position of its tile, which reorders the attention sums: equal to rounding, not bitwise)
    _close(s1.cpu(), s[k1].cpu(), 2e-6, 1e-5, "sample() vs sample_k()[K-1]")
    # metrics (relative_to_abs + ADE/FDE raw + min-over-K per scene)
    from dagnet_amd import _capi
    from dagnet_amd._runtime import scene_ptr_from_sse
    plan = _capi.HipPlan(None, scene_ptr_from_sse(sse), 1, device=DEV)
    r = _capi.best_of_k(plan, torch.from_numpy(gold["samples"]).to(DEV), obs[-1].to(DEV), gt.to(DEV))
    _close(r["ade_raw"].cpu(), gold["ade_raw"], 1e-4, 1e-5, "ade_raw")
    _close(r["fde_raw"].cpu(), gold["fde_raw"], 1e-4, 1e-5, "fde_raw")
    _close(r["sums"].cpu()[0], gold["ade_sum"], 0, LOSS_RTOL, "ade_sum")
    _close(r["sums"].cpu()[1], gold["fde_sum"], 0, LOSS_RTOL, "fde_sum")


@pytest.mark.parametrize("name", cases.list_graph_cases())
def test_graph_modules_match_reference(name):
    from dagnet_amd import GAT, GCN
    from dagnet_amd.adjacency_matrix import adjs_distance_sim_pred, adjs_fully_connected_pred, adjs_knn_sim_pred
    from dagnet_amd.synthetic import random_state_dict
    meta, gold = cases.load_golden(name)
    x, pos, sse = cases.graph_inputs(meta)
    mod = GAT(meta["nin"], meta["nhid"], meta["nout"], meta["alpha"], meta["nheads"]) if meta["kind"] == "gat" \
        else GCN(meta["nin"], meta["nhid"], meta["nout"])
    mod.load_state_dict(random_state_dict(mod, meta["seed"]))
    mod = mod.to(DEV).eval()
    x, pos = x.to(DEV), pos.to(DEV)
    adjs = [adjs_fully_connected_pred(sse, device=DEV), adjs_distance_sim_pred(1.2, sse, pos), adjs_knn_sim_pred(3, sse, pos)]
    for at, adj in enumerate(adjs):
        assert adj.dtype == torch.float32 and tuple(adj.shape) == gold[f"adj{at}"].shape
        _close(adj.cpu(), gold[f"adj{at}"], 0, 1e-6, f"adjacency type {at}")
        out = mod(x, adj, seq_start_end=sse)
        _close(out.cpu(), gold[f"out{at}"], 1e-5, 1e-5, f"graph out (adj {at})")
        if at == 0:      # the reference signature: no seq_start_end, blocks inferred from adj
            assert torch.equal(mod(x, adj), out)
