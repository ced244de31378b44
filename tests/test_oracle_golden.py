"""CPU: the oracle restatement (oracle/dagnet_oracle.py) against the golden outputs captured from
the reference itself (tests/golden, made by oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import dagnet_oracle as orc
from tests import cases


@pytest.mark.parametrize("name", cases.list_cases())
def test_oracle_matches_reference(name):
    meta, gold = cases.load_golden(name)
    args, model, sd = cases.build_model(meta)
    (obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse), eps_f, eps_s, q_s = cases.build_inputs(meta)
    cfg = orc.make_cfg(n_max_agents=meta["n_max_agents"], **meta["args"])
    kind, T = meta["kind"], meta["pred_len"]
    with torch.no_grad():
        if kind == "vrnn":
            kld, nll, h = orc.vrnn_forward(sd, cfg, obs_rel, eps_f)
            ce = torch.zeros(1)
        else:
            adj = orc.adjacency_seq(cfg, sse, torch.cat([obs, gt]))
            if kind == "avrnn":
                kld, nll, h = orc.avrnn_forward(sd, cfg, obs_rel, adj, eps_f)
                ce = torch.zeros(1)
            else:
                g_ohe = orc.one_hot(obs_goals, cfg.g_dim)
                kld, nll, ce, h = orc.dagnet_forward(sd, cfg, obs, obs_rel, g_ohe, sse, adj, eps_f)
        np.testing.assert_allclose(h.numpy(), gold["fwd_h"], atol=2e-5, rtol=0)
        np.testing.assert_allclose(kld.numpy(), gold["fwd_kld"], rtol=2e-5)
        np.testing.assert_allclose(nll.numpy(), gold["fwd_nll"], rtol=2e-5)
        np.testing.assert_allclose(ce.numpy(), gold["fwd_ce"], rtol=2e-5, atol=1e-6)
        h = torch.from_numpy(gold["fwd_h"])
        outs = []
        for k in range(meta["K"]):
            if kind == "vrnn":
                s = orc.vrnn_sample(sd, cfg, T, h, eps_s[k])
            elif kind == "avrnn":
                s = orc.avrnn_sample(sd, cfg, T, h, obs[-1], sse, eps_s[k])
            else:
                tr = []
                s = orc.dagnet_sample(sd, cfg, T, h, obs[-1], g_ohe[-1], sse, eps_s[k], q_s[k], trace=tr)
                goals = torch.stack([t["goal"] for t in tr]).numpy()
                assert np.array_equal(goals, gold["goals"][k]), "goal draws differ from the reference"
            outs.append(s)
        samples = torch.stack(outs)
        np.testing.assert_allclose(samples.numpy(), gold["samples"], atol=2e-5, rtol=2e-5)
        a_sum, f_sum = orc.evaluate_best_of_k(torch.from_numpy(gold["samples"]), obs[-1], gt, sse)
        np.testing.assert_allclose(float(a_sum), float(gold["ade_sum"]), rtol=1e-6)
        np.testing.assert_allclose(float(f_sum), float(gold["fde_sum"]), rtol=1e-6)
        a0 = orc.relative_to_abs(torch.from_numpy(gold["samples"][0]), obs[-1])
        np.testing.assert_allclose(orc.ade_raw(a0, gt).numpy(), gold["ade_raw"][0], rtol=1e-6)
        np.testing.assert_allclose(orc.fde_raw(a0[-1], gt[-1]).numpy(), gold["fde_raw"][0], rtol=1e-6)


@pytest.mark.parametrize("name", cases.list_graph_cases())
def test_oracle_graph_modules(name):
    meta, gold = cases.load_golden(name)
    x, pos, sse = cases.graph_inputs(meta)
    from dagnet_amd import GAT, GCN
    from dagnet_amd.synthetic import random_state_dict
    mod = GAT(meta["nin"], meta["nhid"], meta["nout"], meta["alpha"], meta["nheads"]) if meta["kind"] == "gat" \
        else GCN(meta["nin"], meta["nhid"], meta["nout"])
    sd = {"g." + k: v for k, v in random_state_dict(mod, meta["seed"]).items()}
    adjs = [orc.adj_fully_connected(sse), orc.adj_distance_sim(1.2, sse, pos), orc.adj_knn_sim(3, sse, pos)]
    for at, adj in enumerate(adjs):
        np.testing.assert_allclose(adj.numpy(), gold[f"adj{at}"], rtol=1e-6, atol=0)
        with torch.no_grad():
            out = orc.gat_forward(sd, "g", x, adj, meta["alpha"], meta["nheads"]) if meta["kind"] == "gat" \
                else orc.gcn_forward(sd, "g", x, adj)
        np.testing.assert_allclose(out.numpy(), gold[f"out{at}"], atol=2e-6, rtol=1e-5)
