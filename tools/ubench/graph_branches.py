"""Do independent branches of a captured hipGraph run concurrently?  Two chains of 64 -> 64 skinny GEMMs over 81 920 rows (the
training step's layer shape), captured (a) back to back on one stream, (b) forked onto two streams that join at the end.
    python tools/ubench/graph_branches.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dagnet_amd import _ops  # noqa: E402

dev = "cuda:0"
M, L = 81920, 12
W = torch.randn(64, 64, device=dev) / 8
b = torch.zeros(64, device=dev)
bufs = [[torch.randn(M, 64, device=dev) for _ in range(L + 1)] for _ in range(3)]


def chain(k):
    for i in range(L):
        _ops.gemm(bufs[k][i], W, bufs[k][i + 1], bias=b, act=1)


def capture(nbranch, forked):
    g = torch.cuda.CUDAGraph()
    side = [torch.cuda.Stream() for _ in range(nbranch - 1)]
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        if not forked:
            for k in range(nbranch):
                chain(k)
        else:
            ev = torch.cuda.Event()
            ev.record(main)
            done = []
            for k, s in enumerate(side):
                s.wait_event(ev)
                with torch.cuda.stream(s):
                    chain(k + 1)
                    e = torch.cuda.Event()
                    e.record(s)
                    done.append(e)
            chain(0)
            for e in done:
                main.wait_event(e)
    return g


def timeit(g, n=20):
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


chain(0); chain(1); chain(2)
torch.cuda.synchronize()
for nb in (1, 2, 3):
    t_seq = timeit(capture(nb, False))
    t_par = timeit(capture(nb, True)) if nb > 1 else float("nan")
    print(f"{nb} chain(s) of {L} GEMMs: one stream {t_seq:8.1f} us ({t_seq / nb / L:5.1f} us per GEMM)   forked {t_par:8.1f} us", flush=True)
