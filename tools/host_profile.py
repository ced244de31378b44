"""Host-side cost of one sample() call (the reference's evaluation loop is K of them): cProfile over 200 calls.
python tools/host_profile.py [workload]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dagnet_amd import synthetic as syn

name = sys.argv[1] if len(sys.argv) > 1 else "avrnn_bsk_512x20"
wl = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
a, model = bench.build(wl, dev)
T = wl["pred"]
sizes = bench.scene_sizes(wl)
obs, gt, obs_rel, gt_rel, og, pg, sse = [t.to(dev) for t in syn.make_batch(sizes, wl["obs"], T, 1000, kind=wl.get("data", "bsk"),
                                                                           g_cells=wl.get("g_cells", (9, 10)))]
n = sse[-1, 1].item()
h = (torch.rand(a.n_layers, n, a.rnn_dim, device=dev) - 0.5)
x0, g0 = obs[-1].contiguous(), og[-1].to(torch.int32).contiguous()


def call(k):
    if wl["kind"] == "dagnet":
        return model.sample(T, h, x0, g0, sse, seed=1000 + k)
    if wl["kind"] == "avrnn":
        return model.sample(T, h, x0, sse, seed=1000 + k)
    return model.sample(T, h, seed=1000 + k)


for k in range(3):
    call(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(200):
    call(k)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{name}: host {1e6 * (t1 - t0) / 200:.0f} us per call, wall {1e6 * (t2 - t0) / 200:.0f} us per call")
pr = cProfile.Profile()
pr.enable()
for k in range(200):
    call(k)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
