"""Times the weight-gradient GEMM (dagnet_op_gemm_tn_partial) and the narrow forward GEMM alone, at the training bench's shapes.
    python tools/ubench/tn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dagnet_amd import _ops
from dagnet_amd._ops import _lib, _m, _p, _stream

dev = "cuda:0"
M = 81920
lib = _lib()
for N, K in ((64, 64), (64, 128), (192, 128), (96, 64)):
    A = torch.randn(M, N, device=dev)
    B = torch.randn(M, K, device=dev)
    part = torch.empty(int(lib.dagnet_op_gemm_tn_workspace(M, N, K)), device=dev)
    (a, lda), (b, ldb) = _m(A, "A"), _m(B, "B")
    def run(first):
        lib.dagnet_op_gemm_tn_partial(a, lda, b, ldb, None, 1, M, N, K, _p(part), first, _stream(A.device))
    for first in (1, 0):
        for _ in range(3): run(first)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(first)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"gemm_tn partial M={M} N={N} K={K} first={first}: {us:.1f} us, {(M*(N+K)*4)/us/1e6:.2f} TB/s of inputs, {2*M*N*K/us/1e6:.1f} TFLOP/s")
    W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    for _ in range(3): _ops.gemm(B, W, C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): _ops.gemm(B, W, C)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"gemm_nt M={M} N={N} K={K}: {us:.1f} us, {(M*(N+K)*4)/us/1e6:.2f} TB/s, {2*M*N*K/us/1e6:.1f} TFLOP/s")
