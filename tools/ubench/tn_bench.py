"""Time dagnet_op_gemm_tn (weight gradients, C[N,K] = A[M,N]^T B[M,K]) at the training step's stacked shapes (M = steps x rows),
operands rotated through several buffers so they come from HBM.
    python tools/ubench/tn_bench.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dagnet_amd import _ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 49 * 40960
dev = "cuda:0"
NB = 3


def bench(fn, reps=4):
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


for (N, K, bias) in ((64, 64, True), (64, 64, False), (192, 64, True), (64, 128, True), (64, 32, True), (32, 64, True), (96, 64, True),
                     (64, 92, True), (8, 64, False), (64, 2, True)):
    A = [torch.randn((M, N), device=dev) for _ in range(NB)]
    B = [torch.randn((M, K), device=dev) for _ in range(NB)]
    Cw = torch.empty((N, K), device=dev)
    db = torch.empty(N, device=dev) if bias else None
    t = bench(lambda i: _ops.gemm_tn(A[i], B[i], Cw, dbias=db))
    byt = 4 * M * (N + K)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} bias={int(bias)}: {t:8.1f} us  {byt / t / 1e6:6.2f} TB/s  {fl / t / 1e6:6.1f} TFLOP/s", flush=True)
    del A, B
