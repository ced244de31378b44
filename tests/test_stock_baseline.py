"""CPU: the stock-torch.nn stand-in that bench.py times as the reference's CPU path (oracle/stock_torch.py, dense [N,N,2F]
attention, per-scene scipy adjacency) against the golden ``samples`` captured from the reference itself.  Noise is injected
the way oracle/make_golden.py injects it into the reference: torch.rand_like / torch.multinomial are wrapped."""
import numpy as np
import pytest
import torch

from oracle.stock_torch import StockRollout
from tests import cases


class _Inject:
    def __init__(self, eps, q):
        self.eps, self.q = list(eps), list(q) if q is not None else []
        self._r, self._m = torch.rand_like, torch.multinomial

    def __enter__(self):
        torch.rand_like = lambda x, *a, **k: self.eps.pop(0).clone()
        torch.multinomial = lambda p, n, *a, **k: torch.argmax(p / self.q.pop(0), dim=-1, keepdim=True)
        return self

    def __exit__(self, *a):
        torch.rand_like, torch.multinomial = self._r, self._m


@pytest.mark.parametrize("name", ["vrnn_bsk", "avrnn_bsk_gat", "dagnet_bsk_gat_n10", "dagnet_tiny_gcn_adj2", "avrnn_tiny_gat_adj2",
                                  "dagnet_tiny_gat_adj0"])
def test_stock_torch_rollout_matches_reference(name):
    meta, gold = cases.load_golden(name)
    args, _, sd = cases.build_model(meta)
    (obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse), eps_f, eps_s, q_s = cases.build_inputs(meta)
    kind, T = meta["kind"], meta["pred_len"]
    m = StockRollout(kind, args, meta["n_max_agents"])
    m.load_state_dict(sd)
    m.eval()
    h = torch.from_numpy(gold["fwd_h"])
    g0 = torch.nn.functional.one_hot(obs_goals[-1].long(), args.g_dim).float() if kind == "dagnet" else None
    for k in range(meta["K"]):
        with _Inject([eps_s[k, t] for t in range(T)], [q_s[k, t] for t in range(T)] if q_s is not None else None) as inj:
            s = m.sample(T, h.clone(), obs[-1] if kind != "vrnn" else None, g0, sse)
            assert not inj.eps and not inj.q
        np.testing.assert_allclose(s.numpy(), gold["samples"][k], atol=2e-5, rtol=2e-5)
