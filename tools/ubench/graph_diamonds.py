"""How long does hipGraph capture take for D fork/join diamonds in series?  (Each diamond: the capturing stream and one side
stream run one small GEMM each, then join.)  The time is printed per D; the run stops once a capture takes over 5 s.
    python tools/ubench/graph_diamonds.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dagnet_amd import _ops  # noqa: E402

dev = "cuda:0"
M = 4096
W = torch.randn(64, 64, device=dev) / 8
a, b, c, d = (torch.randn(M, 64, device=dev) for _ in range(4))
_ops.gemm(a, W, b)
torch.cuda.synchronize()
side = torch.cuda.Stream()
keep = []
for D in (2, 4, 8, 12, 16, 20, 24, 28, 32, 48, 64, 128):
    g = torch.cuda.CUDAGraph()
    t0 = time.perf_counter()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for _ in range(D):
            e = torch.cuda.Event(); e.record(main); side.wait_event(e); keep.append(e)
            with torch.cuda.stream(side):
                _ops.gemm(c, W, d)
            _ops.gemm(a, W, b)
            e = torch.cuda.Event(); e.record(side); main.wait_event(e); keep.append(e)
    dt = time.perf_counter() - t0
    print(f"D={D:4d}: capture {dt * 1e3:9.1f} ms", flush=True)
    if dt > 5:
        break
