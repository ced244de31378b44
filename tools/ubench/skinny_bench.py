"""Time dagnet_op_gemm on the training step's tall-and-skinny shapes with operands rotated through > 1 GB of buffers (cold in
L2 / Infinity Cache, like consecutive layers of a step) and, for comparison, a device copy of the same bytes.
    python tools/ubench/skinny_bench.py [M]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dagnet_amd import _ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 81920
dev = "cuda:0"
NB = 24


def bench(fn, reps=5):
    fn(0)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(NB):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / NB * 1e3)
    return best


def pad4(n):
    return (n + 3) & ~3


SHAPES = ((64, 64, False, False, 1), (64, 64, True, False, 0), (64, 64, True, True, 0), (192, 64, False, False, 0),
          (64, 192, True, False, 0), (32, 64, False, False, 0), (64, 32, False, False, 1), (92, 64, False, False, 0),
          (90, 90, True, False, 0), (90, 90, False, False, 0), (64, 192, True, True, 0), (128, 192, True, False, 0),
          (64, 96, True, False, 0), (64, 80, True, False, 0), (96, 64, True, False, 0), (96, 96, True, False, 0), (90, 90, True, True, 0))
if len(sys.argv) > 2:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
for (N, K, transb, acc, act) in SHAPES:
    transb, acc = bool(transb), bool(acc)
    Kk = K
    # rows padded to 16 bytes, as StepPool hands them out in the training step
    A = [torch.randn((M, pad4(Kk)), device=dev)[:, :Kk] for _ in range(NB)]
    C = [torch.empty((M, pad4(N)), device=dev)[:, :N] for _ in range(NB)]
    W = torch.randn((Kk, N) if transb else (N, Kk), device=dev)
    b = torch.randn(N, device=dev)
    t = bench(lambda i: _ops.gemm(A[i], W, C[i], transb=transb, bias=None if acc else b, act=act, accumulate=acc))
    byt = 4 * M * (Kk + N * (2 if acc else 1))
    tc = bench(lambda i: C[i][:, :min(N, Kk)].copy_(A[i][:, :min(N, Kk)]))
    print(f"M={M} N={N} K={Kk} transb={int(transb)} acc={int(acc)} act={act}: {t:7.1f} us  {byt / t / 1e6:6.2f} TB/s   "
          f"(copy of [M,{min(N, Kk)}]: {tc:6.1f} us)", flush=True)
