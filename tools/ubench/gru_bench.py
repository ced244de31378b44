"""Time dagnet_op_gru_layer_fwd (one fused kernel per GRU layer step, csrc/gru_fused.hip) alone on the chip, operands rotated
through several buffers, against the unfused form (DAGNET_NO_GRU_FUSED=1: two or three gate products + cell).
    python tools/ubench/gru_bench.py [M]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dagnet_amd import _ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 81920
dev, R, NB = "cuda:0", 64, 6
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
lib = _ops._lib()
for pieces in ((64, 64), (64,)):
    K = sum(pieces)
    xs = [[torch.randn((M, k), device=dev) for k in pieces] for _ in range(NB)]
    h = [torch.randn((M, R), device=dev) for _ in range(NB)]
    Wih, Whh = torch.randn((3 * R, K), device=dev) / K ** 0.5, torch.randn((3 * R, R), device=dev) / 8
    bih, bhh = torch.randn(3 * R, device=dev), torch.randn(3 * R, device=dev)
    gi, gh = [torch.empty((M, 3 * R), device=dev) for _ in range(NB)], [torch.empty((M, 3 * R), device=dev) for _ in range(NB)]
    hn, gates = [torch.empty((M, R), device=dev) for _ in range(NB)], [torch.empty((M, 3 * R), device=dev) for _ in range(NB)]
    k1 = pieces[1] if len(pieces) > 1 else 0
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    n_pack = lib.dagnet_op_gru_pack_floats(R, pieces[0], k1)
    pack = torch.empty(n_pack, device=dev)
    _ops._check(lib.dagnet_op_gru_pack(R, pieces[0], k1, p(Wih), K, p(bih), p(Whh), R, p(bhh), p(pack), st))
    for use_pack in (True, False):
        def run(i):
            x = xs[i]
            _ops._check(lib.dagnet_op_gru_layer_fwd(M, R, p(x[0]), pieces[0], pieces[0], p(x[1]) if k1 else None, k1, k1, p(h[i]), R,
                                                    p(Wih), K, p(bih), p(Whh), R, p(bhh), p(gi[i]), p(gh[i]), p(hn[i]), p(gates[i]),
                                                    p(pack) if use_pack else None, st))
        run(0)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(NB):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / NB * 1e3)
        fl = 2.0 * M * 3 * R * (K + R)
        print(f"M={M} pieces={pieces} pack={int(use_pack)}: {best:7.1f} us  {fl / best / 1e6:6.1f} TFLOP/s", flush=True)
