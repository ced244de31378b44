"""Per-phase cycle breakdown of one tile of the rollout kernel (diagnostic build with -DDN_PROFILE).

    make -C dagnet_amd/csrc ../_lib/libdagnet_hip_prof.so
    python tools/phase_profile.py [workload [scenes]]      (scenes * K / 3 <= 1024 tiles: one wave per SIMD)
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from dagnet_amd import _capi

_capi._LIB_PATH = os.path.join(ROOT, "dagnet_amd", "_lib", "libdagnet_hip_prof.so")
import bench

PHASES = ["0 d_t+adjacency", "1 dec_goal+softmax", "2 goal graph", "3 lg_goals+softmax+draw", "4 prior", "5 z",
          "6 phi_z+dec+dec_mean", "7 phi_x", "8 GRU", "9 hidden graph+blend+xabs",
          "10 goal GAT: heads+fold MFMAs", "11 goal GAT: s1/s2 exchange", "12 goal GAT: 4 heads softmax+aggregation",
          "13 goal GAT: ELU+fold_out+out softmax", "14 goal GAT: out tiles", "15 goal GAT: shift+tanh"]


def main():
    wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "dagnet_bsk_4096x20"]
    dev = torch.device("cuda", 0)
    a, model = bench.build(wl, dev)
    from dagnet_amd import synthetic as syn
    S, n, K, T = (int(sys.argv[2]) if len(sys.argv) > 2 else 512), wl["agents"], wl["K"], wl["pred"]
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = [t.to(dev) for t in syn.make_batch([n] * S, wl["obs"], T, 5)]
    h = (torch.rand(a.n_layers, S * n, a.rnn_dim, device=dev) - 0.5)
    prof = torch.zeros(16, dtype=torch.int64, device=dev)
    lib = _capi.load()
    lib.dagnet_debug_set_prof.argtypes = [ctypes.c_void_p]
    for it in range(2):
        prof.zero_()
        lib.dagnet_debug_set_prof(ctypes.c_void_p(prof.data_ptr()))
        if wl["kind"] == "dagnet":
            model.sample_k(K, T, h, obs[-1], obs_goals[-1].to(torch.int32), sse, seed=1)
        elif wl["kind"] == "avrnn":
            model.sample_k(K, T, h, obs[-1], sse, seed=1)
        else:
            model.sample_k(K, T, h, seed=1)
        torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(float)
    tot = p.sum()
    for i, name in enumerate(PHASES):
        print(f"{name:32s} {p[i] / T:12.0f} ticks/step  {100 * p[i] / tot:5.1f}%")
    print(f"{'total':32s} {tot / T:12.0f} ticks/step (clock64 = s_memtime: shader clock cycles)")


if __name__ == "__main__":
    main()
