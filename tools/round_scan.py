"""Time of the DAG-Net rollout launch against the number of tiles (GPU box): how much of the launch is round quantisation and
per-tile prologue?   python tools/round_scan.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dagnet_amd import synthetic as syn

wl = bench.WORKLOADS["dagnet_bsk_4096x20"]
dev = torch.device("cuda", 0)
a, model = bench.build(wl, dev)
K = 20
for S, T in ((1536, 40), (3072, 40), (4096, 40), (4608, 40), (3072, 20), (3072, 80), (153, 40), (307, 40), (307, 400), (153, 400), (614, 400)):
    obs, gt, obs_rel, gt_rel, og, gg, sse = [t.to(dev) for t in syn.make_batch([10] * S, 10, T, 5)]
    h = torch.rand(a.n_layers, S * 10, a.rnn_dim, device=dev) - 0.5
    g0 = og[-1].to(torch.int32)
    for _ in range(2):
        model.sample_k(K, T, h, obs[-1], g0, sse, seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        model.sample_k(K, T, h, obs[-1], g0, sse, seed=1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    tiles = (S * K + 2) // 3
    print(f"S={S:5d} T={T:3d} tiles={tiles:6d} rounds={tiles / 2048:6.2f}  {ms:8.3f} ms  {ms / (tiles / 2048) / T * 1e3:8.2f} us per round and step", flush=True)
