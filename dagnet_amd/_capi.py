"""ctypes binding of libdagnet_hip.so (include/dagnet_hip.h).

There is no fallback: if the library is missing, loading raises with the build command, and every
compute entry point needs a ROCm device.  torch is used only for device buffers and streams.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Dict, Optional, Sequence

import numpy as np
import torch

# DAGNET_HIP_LIB: another build of the same library (kernel experiments: tools/variants.sh); the package never falls back to it
_LIB_PATH = os.environ.get("DAGNET_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libdagnet_hip.so")
_lib = None
_lock = threading.Lock()

MODEL_VRNN, MODEL_AVRNN, MODEL_DAGNET, MODEL_GRAPH = 0, 1, 2, 3
GRAPH_NONE, GRAPH_GAT, GRAPH_GCN = 0, 1, 2

# every symbol include/dagnet_hip.h declares (tests check the .so exports all of them)
EXPORTS = (
    "dagnet_last_error", "dagnet_version", "dagnet_model_create", "dagnet_model_destroy",
    "dagnet_plan_create", "dagnet_plan_destroy", "dagnet_plan_info", "dagnet_plan_kernel", "dagnet_sample", "dagnet_forward",
    "dagnet_graph_forward", "dagnet_adjacency", "dagnet_plan_compact_adj_size", "dagnet_best_of_k",
    "dagnet_relative_to_abs", "dagnet_categorical", "dagnet_displacement_errors", "dagnet_scene_min_sum",
)


class DagnetHipError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "model", "x_dim", "h_dim", "z_dim", "rnn_dim", "n_layers", "g_dim", "n_max_agents",
        "graph_model", "graph_hid", "n_heads", "adjacency_type", "top_k_neigh")] + [
        ("alpha", C.c_float), ("sigma", C.c_double)]


class _Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("h_data", C.c_void_p), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load the HIP library (once).  Raises DagnetHipError when it has not been built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise DagnetHipError(
                f"{_LIB_PATH} not found: build it with `make -C dagnet_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        lib = C.CDLL(_LIB_PATH)
        lib.dagnet_last_error.restype = C.c_char_p
        lib.dagnet_version.restype = C.c_char_p
        lib.dagnet_plan_compact_adj_size.restype = C.c_int64
        lib.dagnet_plan_compact_adj_size.argtypes = [C.c_void_p]
        lib.dagnet_model_create.argtypes = [C.POINTER(_Config), C.POINTER(_Tensor), C.c_int32, C.POINTER(C.c_void_p)]
        lib.dagnet_model_destroy.argtypes = [C.c_void_p]
        lib.dagnet_model_destroy.restype = None
        lib.dagnet_plan_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        lib.dagnet_plan_destroy.argtypes = [C.c_void_p]
        lib.dagnet_plan_destroy.restype = None
        lib.dagnet_plan_kernel.argtypes = [C.c_void_p]
        lib.dagnet_plan_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        vp = C.c_void_p
        lib.dagnet_sample.argtypes = [vp, vp, C.c_int32, vp, vp, vp, vp, vp, vp, C.c_uint64, vp, vp, vp]
        lib.dagnet_forward.argtypes = [vp, vp, C.c_int32, vp, vp, vp, vp, C.c_int64, vp, C.c_uint64, vp, vp, vp]
        lib.dagnet_graph_forward.argtypes = [vp, vp, C.c_int32, vp, vp, C.c_int64, vp, vp]
        lib.dagnet_adjacency.argtypes = [vp, C.c_int32, C.c_double, C.c_int32, C.c_int32, vp, vp, C.c_int64, vp]
        lib.dagnet_best_of_k.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.dagnet_relative_to_abs.argtypes = [C.c_int32, C.c_int32, vp, vp, vp, vp]
        lib.dagnet_displacement_errors.argtypes = [C.c_int32, C.c_int32, vp, vp, vp, vp, vp]
        lib.dagnet_scene_min_sum.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp]
        lib.dagnet_categorical.argtypes = [C.c_int32, C.c_int32, vp, vp, C.c_uint64, vp, vp]
        _lib = lib
        return lib


def _check(rc: int):
    if rc != 0:
        raise DagnetHipError(f"libdagnet_hip: {load().dagnet_last_error().decode()} (status {rc})")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _need_cuda(t: torch.Tensor, name: str, dtype=torch.float32):
    if not t.is_cuda:
        raise DagnetHipError(f"{name} must live on a ROCm device (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise DagnetHipError(f"{name} must be {dtype} (got {t.dtype})")
    return t.contiguous()


class HipModel:
    """Packed weights of one model on one device (dagnet_model*)."""

    def __init__(self, cfg: Dict, state_dict: Dict[str, torch.Tensor], device):
        lib = load()
        self.device = torch.device(device)
        c = _Config()
        for k, v in cfg.items():
            setattr(c, k, v)
        self.cfg = dict(cfg)
        keep = []
        arr = (_Tensor * len(state_dict))()
        n = 0
        for name, t in state_dict.items():
            if name.endswith("num_batches_tracked"):
                continue
            a = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
            keep.append(a)
            arr[n].name = name.encode()
            arr[n].h_data = a.ctypes.data
            arr[n].ndim = a.ndim
            for i, s in enumerate(a.shape):
                arr[n].shape[i] = s
            n += 1
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib.dagnet_model_create(C.byref(c), arr, n, C.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.dagnet_model_destroy(h)


class HipPlan:
    """Scene -> tile tables for one batch layout (dagnet_plan*). model=None: tables only."""

    def __init__(self, model: Optional[HipModel], scene_ptr: Sequence[int], n_samples: int = 1, device=None):
        lib = load()
        self.model = model
        self.device = model.device if model is not None else torch.device(device)
        sp = np.ascontiguousarray(np.asarray(scene_ptr, dtype=np.int32))
        self.scene_ptr = sp
        self.S = int(sp.shape[0] - 1)
        self.N = int(sp[-1])
        self.K = int(n_samples)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib.dagnet_plan_create(model._h if model is not None else None, sp.ctypes.data, self.S, self.K, C.byref(h)))
        self._h = h
        self.compact_size = int(lib.dagnet_plan_compact_adj_size(h))

    def info(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int64()
        _check(load().dagnet_plan_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        kern = {3: "gemm_nt_kernel (wide goal head: batched MFMA GEMMs, fp16 x2 split products in the wide layers, + register-resident step kernel)",
                2: "rollout_v2_kernel (register-resident)", 1: "dagnet_rollout_big_kernel (generic, goal buffers in global scratch)",
                0: "dagnet_rollout_kernel (generic, LDS-tiled)"}.get(load().dagnet_plan_kernel(self._h), "none")
        return dict(tile_rows=a.value, n_tiles=b.value, lds_bytes=c.value, sample_kernel=kern)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.dagnet_plan_destroy(h)


def sample(model: HipModel, plan: HipPlan, T: int, h0, x0=None, g0=None, eps=None, q=None, force_goals=None,
           seed: int = 0, out_goals: bool = False):
    """-> samples [K,T,N,x_dim] (and goal indices int32 [K,T,N] if out_goals)."""
    dev = model.device
    h0 = _need_cuda(h0, "h")
    x0 = None if x0 is None else _need_cuda(x0, "x_abs_start")
    g0 = None if g0 is None else _need_cuda(g0, "g_start", torch.int32)
    eps = None if eps is None else _need_cuda(eps, "eps")
    q = None if q is None else _need_cuda(q, "q")
    force_goals = None if force_goals is None else _need_cuda(force_goals, "force_goals", torch.int32)
    K, N, X = plan.K, plan.N, model.cfg["x_dim"]
    L, R, Z, G = model.cfg["n_layers"], model.cfg["rnn_dim"], model.cfg["z_dim"], model.cfg["g_dim"]
    if tuple(h0.shape) != (L, N, R):
        raise DagnetHipError(f"h must be [{L},{N},{R}], got {tuple(h0.shape)}")
    if x0 is not None and tuple(x0.shape) != (N, 2):
        raise DagnetHipError(f"x_abs_start must be [{N},2], got {tuple(x0.shape)}")
    if g0 is not None and tuple(g0.shape) != (N,):
        raise DagnetHipError(f"g_start must be [{N}] goal indices, got {tuple(g0.shape)}")
    if eps is not None and tuple(eps.shape) != (K, T, N, Z):
        raise DagnetHipError(f"eps must be [{K},{T},{N},{Z}], got {tuple(eps.shape)}")
    if q is not None and tuple(q.shape) != (K, T, N, G):
        raise DagnetHipError(f"q must be [{K},{T},{N},{G}], got {tuple(q.shape)}")
    if force_goals is not None and tuple(force_goals.shape) != (K, T, N):
        raise DagnetHipError(f"force_goals must be [{K},{T},{N}], got {tuple(force_goals.shape)}")
    ins = [t for t in (h0, x0, g0, eps, q, force_goals) if t is not None]
    with torch.cuda.device(dev):
        side = _overlap_stream(model, plan, ins) if T > 0 else None
        if side is None:
            out = torch.empty((K, T, N, X), dtype=torch.float32, device=dev)
            goals = torch.empty((K, T, N), dtype=torch.int32, device=dev) if out_goals else None
            _check(load().dagnet_sample(model._h, plan._h, T, _ptr(h0), _ptr(x0), _ptr(g0), _ptr(eps), _ptr(q),
                                        _ptr(force_goals), C.c_uint64(seed & (2**64 - 1)), _ptr(out), _ptr(goals), _stream(dev)))
        else:
            cur = torch.cuda.current_stream(dev)
            with torch.cuda.stream(side):        # outputs from the side stream's pool: no block of the caller's stream is reused early
                out = torch.empty((K, T, N, X), dtype=torch.float32, device=dev)
                goals = torch.empty((K, T, N), dtype=torch.int32, device=dev) if out_goals else None
            try:
                _check(load().dagnet_sample(model._h, plan._h, T, _ptr(h0), _ptr(x0), _ptr(g0), _ptr(eps), _ptr(q),
                                            _ptr(force_goals), C.c_uint64(seed & (2**64 - 1)), _ptr(out), _ptr(goals),
                                            C.c_void_p(side.cuda_stream)))
            finally:
                done = torch.cuda.Event()
                done.record(side)
                cur.wait_event(done)             # everything the caller queues from here on sees the finished samples
            out.record_stream(cur)
            if goals is not None:
                goals.record_stream(cur)
    return (out, goals) if out_goals else out


# --- small launches from a loop: K separate sample() calls (the reference's evaluation loop, models/dagnet/evaluate_bsk.py:113-114) ---
# A launch of the register-resident kernel that does not fill the chip's wave slots a few times over leaves SIMDs idle while its
# last waves finish, and the next call's launch waits behind it on the caller's stream although it depends on nothing the first one
# writes.  Such launches go to one of three side streams instead: the side stream waits for an event recorded on the caller's stream
# when these INPUTS were first seen (same storage, same version counters: nothing torch knows of has written them since; the
# tensors are kept referenced so that their addresses cannot be handed to other tensors), the caller's stream waits for the
# launch's completion event, and consecutive calls overlap on the device like the tiles of one large launch.  DAGNET_NO_OVERLAP=1
# switches it off.  The bookkeeping is per model and not locked: callers that drive ONE model from several host threads should set
# that variable.  Not used under graph capture, for launches of >= _OVERLAP_ROUNDS rounds, or for kernels that work in the
# plan's buffers (wide head, generic kernel with global scratch).
_OVERLAP_ROUNDS = 4
_side_streams: Dict[int, list] = {}
_wave_slots: Dict[int, int] = {}


def _overlap_stream(model: HipModel, plan: HipPlan, ins):
    if os.environ.get("DAGNET_NO_OVERLAP") or torch.cuda.is_current_stream_capturing():
        return None
    lib = load()
    if lib.dagnet_plan_kernel(plan._h) != 2:
        return None
    idx = model.device.index if model.device.index is not None else torch.cuda.current_device()
    slots = _wave_slots.get(idx)
    if slots is None:
        slots = _wave_slots[idx] = torch.cuda.get_device_properties(idx).multi_processor_count * 8   # 4 SIMDs x 2 waves of 256 registers
    n_tiles = C.c_int32()
    _check(lib.dagnet_plan_info(plan._h, None, C.byref(n_tiles), None))
    if n_tiles.value >= _OVERLAP_ROUNDS * slots:
        return None
    cur = torch.cuda.current_stream(model.device)
    key = (cur.cuda_stream,) + tuple((t.data_ptr(), t._version) for t in ins)
    st = model.__dict__.setdefault("_ovl", {"n": 0})
    if st.get("key") != key:
        ready = torch.cuda.Event()
        ready.record(cur)
        st.update(key=key, refs=list(ins), ready=ready)
    streams = _side_streams.get(idx)
    if streams is None:
        # three side streams + the caller's = the four hardware queues a process has here (GPU_MAX_HW_QUEUES); launches of a few
        # hundred tiles (512 scenes: 171) need more than two in flight to fill the chip
        n_side = max(1, int(os.environ.get("DAGNET_OVERLAP_STREAMS", "3")))
        streams = _side_streams[idx] = [torch.cuda.Stream(device=idx) for _ in range(n_side)]
    # two launches of half the chip's wave slots or more already fill it (and every further stream is one more allocator pool to
    # warm: a hipMalloc inside a loop of calls synchronises the device)
    n_use = min(len(streams), 2) if 2 * n_tiles.value >= slots else len(streams)
    side = streams[st["n"] % n_use]
    st["n"] += 1
    side.wait_event(st["ready"])
    return side


def forward(model: HipModel, plan: HipPlan, traj_rel, traj=None, goals=None, adj=None, adj_dense: bool = True,
            eps=None, seed: int = 0):
    """-> (losses float64[3] on device: KLD, NLL, CE ; h [L,N,R])."""
    dev = model.device
    traj_rel = _need_cuda(traj_rel, "traj_rel")
    traj = None if traj is None else _need_cuda(traj, "traj")
    goals = None if goals is None else _need_cuda(goals, "goals", torch.int32)
    adj = None if adj is None else _need_cuda(adj, "adj")
    eps = None if eps is None else _need_cuda(eps, "eps")
    T, N = int(traj_rel.shape[0]), plan.N
    L, R, Z = model.cfg["n_layers"], model.cfg["rnn_dim"], model.cfg["z_dim"]
    if tuple(traj_rel.shape) != (T, N, model.cfg["x_dim"]):
        raise DagnetHipError(f"traj_rel must be [T,{N},{model.cfg['x_dim']}], got {tuple(traj_rel.shape)}")
    if traj is not None and tuple(traj.shape) != (T, N, 2):
        raise DagnetHipError(f"traj must be [{T},{N},2], got {tuple(traj.shape)}")
    if goals is not None and tuple(goals.shape) != (T, N):
        raise DagnetHipError(f"goals must be [{T},{N}] indices, got {tuple(goals.shape)}")
    if eps is not None and tuple(eps.shape) != (T - 1, N, Z):
        raise DagnetHipError(f"eps must be [{T - 1},{N},{Z}], got {tuple(eps.shape)}")
    if adj is not None:
        want = (N, N) if adj_dense else (plan.compact_size,)
        if adj.shape[0] < T or tuple(adj.shape[1:]) != want:
            raise DagnetHipError(f"adj must be [>={T},{','.join(map(str, want))}], got {tuple(adj.shape)}")
    losses = torch.empty(3, dtype=torch.float64, device=dev)
    h = torch.empty((L, N, R), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _check(load().dagnet_forward(model._h, plan._h, T, _ptr(traj), _ptr(traj_rel), _ptr(goals), _ptr(adj),
                                     N if adj_dense else 0, _ptr(eps), C.c_uint64(seed & (2**64 - 1)), _ptr(losses),
                                     _ptr(h), _stream(dev)))
    return losses, h


def graph_forward(model: HipModel, plan: HipPlan, which: int, x, adj, adj_dense: bool = True):
    dev = model.device
    x = _need_cuda(x, "x")
    adj = _need_cuda(adj, "adj")
    nin = model.cfg["rnn_dim"] if which == 0 else model.cfg["g_dim"]
    if tuple(x.shape) != (plan.N, nin):
        raise DagnetHipError(f"x must be [{plan.N},{nin}], got {tuple(x.shape)}")
    want = (plan.N, plan.N) if adj_dense else (plan.compact_size,)
    if tuple(adj.shape) != want:
        raise DagnetHipError(f"adj must be {want}, got {tuple(adj.shape)}")
    nout = model.cfg["h_dim"] if model.cfg["model"] == MODEL_GRAPH else nin
    out = torch.empty((plan.N, nout), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _check(load().dagnet_graph_forward(model._h, plan._h, which, _ptr(x), _ptr(adj), plan.N if adj_dense else 0,
                                           _ptr(out), _stream(dev)))
    return out


def adjacency(plan: HipPlan, adjacency_type: int, sigma: float, top_k: int, pos, T: int, dense: bool = True):
    """pos [T,N,2] (None for fully connected) -> [T,N,N] dense or [T,sum n^2] compact, on plan.device."""
    dev = plan.device
    if pos is not None:
        pos = _need_cuda(pos, "positions")
        if tuple(pos.shape) != (T, plan.N, 2):
            raise DagnetHipError(f"positions must be [{T},{plan.N},2], got {tuple(pos.shape)}")
    shape = (T, plan.N, plan.N) if dense else (T, plan.compact_size)
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _check(load().dagnet_adjacency(plan._h, adjacency_type, float(sigma), int(top_k or 0), T, _ptr(pos), _ptr(out),
                                       plan.N if dense else 0, _stream(dev)))
    return out


def best_of_k(plan: HipPlan, samples_rel, start, gt):
    """samples_rel [K,T,N,2] -> dict(ade_raw [K,N], fde_raw [K,N], scene_best [S,2], sums float64[2])."""
    dev = plan.device
    samples_rel = _need_cuda(samples_rel, "samples_rel")
    start = _need_cuda(start, "start")
    gt = _need_cuda(gt, "gt")
    K, T, N = int(samples_rel.shape[0]), int(samples_rel.shape[1]), plan.N
    if tuple(samples_rel.shape) != (K, T, N, 2) or tuple(start.shape) != (N, 2) or tuple(gt.shape) != (T, N, 2):
        raise DagnetHipError("best_of_k: want samples_rel [K,T,N,2], start [N,2], gt [T,N,2]; got "
                             f"{tuple(samples_rel.shape)}, {tuple(start.shape)}, {tuple(gt.shape)}")
    ade = torch.empty((K, N), dtype=torch.float32, device=dev)
    fde = torch.empty((K, N), dtype=torch.float32, device=dev)
    best = torch.empty((plan.S, 2), dtype=torch.float32, device=dev)
    sums = torch.empty(2, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _check(load().dagnet_best_of_k(plan._h, K, T, _ptr(samples_rel), _ptr(start), _ptr(gt), _ptr(ade), _ptr(fde),
                                       _ptr(best), _ptr(sums), _stream(dev)))
    return dict(ade_raw=ade, fde_raw=fde, scene_best=best, sums=sums)


def scene_min_sum(plan: HipPlan, err_kn):
    """err [K,N] -> fp32 scalar: sum over scenes of min over K of the scene-summed error (evaluate_helper)."""
    err_kn = _need_cuda(err_kn, "error")
    K, N = int(err_kn.shape[0]), int(err_kn.shape[1])
    if N != plan.N:
        raise DagnetHipError(f"error must be [K,{plan.N}], got {tuple(err_kn.shape)}")
    best = torch.empty((plan.S, 2), dtype=torch.float32, device=err_kn.device)
    sums = torch.empty(2, dtype=torch.float64, device=err_kn.device)
    with torch.cuda.device(err_kn.device):
        _check(load().dagnet_scene_min_sum(plan._h, K, _ptr(err_kn), _ptr(err_kn), _ptr(best), _ptr(sums), _stream(err_kn.device)))
    return sums[0].to(torch.float32)


def displacement_errors(plan: HipPlan, pred_abs, gt):
    """pred_abs, gt [T,N,2] absolute coords -> dict(ade [N], fde [N]) (raw per-agent errors)."""
    pred_abs = _need_cuda(pred_abs, "pred_traj")
    gt = _need_cuda(gt, "pred_traj_gt")
    T, N = int(pred_abs.shape[0]), int(pred_abs.shape[1])
    if tuple(pred_abs.shape) != (T, N, 2) or tuple(gt.shape) != (T, N, 2):
        raise DagnetHipError(f"want two [T,N,2] tensors, got {tuple(pred_abs.shape)}, {tuple(gt.shape)}")
    ade = torch.empty(N, dtype=torch.float32, device=pred_abs.device)
    fde = torch.empty(N, dtype=torch.float32, device=pred_abs.device)
    with torch.cuda.device(pred_abs.device):
        _check(load().dagnet_displacement_errors(T, N, _ptr(pred_abs), _ptr(gt), _ptr(ade), _ptr(fde), _stream(pred_abs.device)))
    return dict(ade=ade, fde=fde)


def relative_to_abs(rel, start):
    rel = _need_cuda(rel, "traj_rel")
    start = _need_cuda(start, "start_pos")
    T, N = int(rel.shape[0]), int(rel.shape[1])
    if tuple(rel.shape) != (T, N, 2) or tuple(start.shape) != (N, 2):
        raise DagnetHipError(f"relative_to_abs: want [T,N,2] and [N,2], got {tuple(rel.shape)}, {tuple(start.shape)}")
    out = torch.empty_like(rel)
    with torch.cuda.device(rel.device):
        _check(load().dagnet_relative_to_abs(T, N, _ptr(rel), _ptr(start), _ptr(out), _stream(rel.device)))
    return out


def categorical(weights, q=None, seed: int = 0):
    """argmax_g weights[n,g] / q[n,g] (q ~ Exp(1), injected or Philox) -> int32 [N]."""
    weights = _need_cuda(weights, "probs")
    q = None if q is None else _need_cuda(q, "q")
    N, G = int(weights.shape[0]), int(weights.shape[1])
    if q is not None and tuple(q.shape) != (N, G):
        raise DagnetHipError(f"q must be [{N},{G}], got {tuple(q.shape)}")
    out = torch.empty(N, dtype=torch.int32, device=weights.device)
    with torch.cuda.device(weights.device):
        _check(load().dagnet_categorical(N, G, _ptr(weights), _ptr(q), C.c_uint64(seed & (2**64 - 1)), _ptr(out),
                                         _stream(weights.device)))
    return out
