"""Time dagnet_op_gemm on the wide layers of the SDD-shaped goal head (g_dim = 960) at 1/8 of the bench's rows.
    python tools/ubench/gemm_bench.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dagnet_amd import _ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 61440
dev = "cuda:0"
# DAGNET_FUSED_MASK: 7 = fp32 MFMA, 15 = three bf16 terms in the wide layers (default), 31 = two fp16 terms
if "DAGNET_FUSED_MASK" in os.environ:
    _ops.set_fused(int(os.environ["DAGNET_FUSED_MASK"]))
    print("fused mask", os.environ["DAGNET_FUSED_MASK"])


def bench(fn, reps=3, n=4):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


SHAPES = ((960, 1920, 0), (960, 960, 0), (960, 64, 0), (960, 32, 2), (40, 960, 0))
if "GEMM_ONLY" in os.environ:          # one shape, by its K (counter passes)
    SHAPES = tuple(sh for sh in SHAPES if sh[1] == int(os.environ["GEMM_ONLY"]) and sh[0] == 960)
for (N, K, act) in SHAPES:
    A = torch.randn((M, K), device=dev)
    W = torch.randn((N, K), device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    C = torch.empty((M, N), device=dev)
    t = bench(lambda: _ops.gemm(A, W, C, bias=b, act=act))
    ref = torch.nn.functional.linear(A[:512], W, b)
    if act == 2:
        ref = torch.tanh(ref)
    err = float((C[:512] - ref).abs().max())
    print(f"M={M} N={N} K={K}: {t * 1e3:8.1f} us  {2.0 * M * N * K / t / 1e9:6.1f} TFLOP/s  {4.0 * M * (N + K) / t / 1e9:5.2f} TB/s  max err vs torch {err:.2e}",
          flush=True)
