"""The reference's evaluation loop as written - K separate sample() calls (models/dagnet/evaluate_bsk.py:113-114) - against one
sample_k launch: wall time, host time of issuing the calls, with and without the side-stream overlap.
python tools/ref_loop.py [workload] [reps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dagnet_amd import synthetic as syn

name = sys.argv[1] if len(sys.argv) > 1 else "dagnet_bsk_4096x20"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
a, model = bench.build(wl, dev)
K, T = wl["K"], wl["pred"]
sizes = bench.scene_sizes(wl)
obs, gt, obs_rel, gt_rel, og, pg, sse = [t.to(dev) for t in syn.make_batch(sizes, wl["obs"], T, 1000, kind=wl.get("data", "bsk"),
                                                                           g_cells=wl.get("g_cells", (9, 10)))]
n = sse[-1, 1].item()
h = (torch.rand(a.n_layers, n, a.rnn_dim, device=dev) - 0.5)
x0, g0 = obs[-1].contiguous(), og[-1].to(torch.int32).contiguous()


def call(k, fn="sample"):
    f = getattr(model, fn)
    pre = (K, T) if fn == "sample_k" else (T,)
    if wl["kind"] == "dagnet":
        return f(*pre, h, x0, g0, sse, seed=1000 + k)
    if wl["kind"] == "avrnn":
        return f(*pre, h, x0, sse, seed=1000 + k)
    return f(*pre, h, seed=1000 + k)


def loop():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        call(k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, (t1 - t0) * 1e3


call(0, "sample_k")
torch.cuda.synchronize()
t0 = time.perf_counter()
call(1, "sample_k")
torch.cuda.synchronize()
print(f"{name}: sample_k {(time.perf_counter() - t0) * 1e3:.2f} ms")
for mode in ("overlap", "serial"):
    if mode == "serial":
        os.environ["DAGNET_NO_OVERLAP"] = "1"
    loop()
    for _ in range(reps):
        w, hst = loop()
        print(f"  {mode}: loop of {K} sample() calls {w:.2f} ms (host issue {hst:.2f} ms)")
