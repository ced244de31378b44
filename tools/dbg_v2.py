import os, sys, subprocess, numpy as np, torch
sys.path.insert(0, '.')
from tests import cases
names = sys.argv[1:] or ["vrnn_bsk", "avrnn_bsk_gat", "dagnet_bsk_gat_n10"]
for name in names:
    meta, gold = cases.load_golden(name)
    args, model, _ = cases.build_model(meta); model = model.to("cuda:0")
    (obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse), _, eps_s, q_s = cases.build_inputs(meta)
    h = torch.from_numpy(gold["fwd_h"]).cuda(); K, T, kind = meta["K"], meta["pred_len"], meta["kind"]
    eps_s = eps_s.cuda()
    if kind == "vrnn": s = model.sample_k(K, T, h, eps=eps_s)
    elif kind == "avrnn": s = model.sample_k(K, T, h, obs[-1].cuda(), sse, eps=eps_s)
    else:
        s, goals = model.sample_k(K, T, h, obs[-1].cuda(), obs_goals[-1].cuda().to(torch.int32), sse, eps=eps_s, q=q_s.cuda(), return_goals=True)
        g = goals.cpu().numpy(); print(name, "goal mismatches per t:", (g != gold["goals"]).reshape(K, T, -1).sum(-1)[0][:10])
    d = np.abs(s.cpu().numpy() - gold["samples"])
    print(name, "max diff per step (k=0):", np.array2string(d[0].reshape(T, -1).max(1)[:8], precision=2), " overall", d.max())
