"""Differentiable operators of the training step (row T), each a torch.autograd.Function whose forward and backward
are HIP kernels behind the C ABI (dagnet_op_* in include/dagnet_hip.h: fp32 MFMA GEMMs of csrc/gemm_f32.hip, row-wise
and per-scene kernels of csrc/train_ops.hip).  torch allocates the buffers and runs the autograd tape; no torch
arithmetic touches activations or gradients in here (only memsets and the reporting-only sqrt of the gradient norm).  There is no CPU path: every tensor must live on a ROCm device.

All matrices are 2-D fp32 with unit column stride; a row-strided view (a column slice of a wider matrix) is passed as
(pointer, row stride) without a copy.
"""
from __future__ import annotations

import contextlib
import os
import ctypes as C
from typing import Dict, Optional, Sequence

import torch

from . import _capi
from ._capi import DagnetHipError, _check, _stream

ACT_NONE, ACT_LRELU, ACT_TANH, ACT_RELU = 0, 1, 2, 3

OP_EXPORTS = (
    "dagnet_op_gemm", "dagnet_op_gemm_dact", "dagnet_op_gemm_tn", "dagnet_op_gemm_tn_workspace", "dagnet_op_gemm_tn_partial", "dagnet_op_gemm_tn_finish", "dagnet_op_act_bwd", "dagnet_op_copy2d",
    "dagnet_op_onehot_grad", "dagnet_op_softmax_fwd", "dagnet_op_softmax_bwd", "dagnet_op_gru_fwd", "dagnet_op_gru_bwd", "dagnet_op_gru_layer_fwd", "dagnet_op_mlp2_fwd", "dagnet_op_mlp_chain_fwd", "dagnet_op_mlp2_bwd", "dagnet_op_linear_pieces_bwd", "dagnet_op_set_fused", "dagnet_op_gru_pack", "dagnet_op_gru_pack_floats", "dagnet_op_gru_pack_bwd", "dagnet_op_gru_pack_bwd_floats", "dagnet_op_gru_layer_bwd",
    "dagnet_op_vae_fwd", "dagnet_op_vae_bwd", "dagnet_op_nll_fwd", "dagnet_op_nll_bwd", "dagnet_op_ce_fwd", "dagnet_op_ce_bwd", "dagnet_op_softmax_ce_bwd",
    "dagnet_op_sum", "dagnet_op_col_workspace", "dagnet_op_colsum", "dagnet_op_colsum_static", "dagnet_op_bn_fwd", "dagnet_op_bn_bwd",
    "dagnet_op_bn_moments", "dagnet_op_bn_merge", "dagnet_op_bn_apply", "dagnet_op_bn_bwd_sums", "dagnet_op_bn_bwd_apply",
    "dagnet_op_gat_attn_fwd", "dagnet_op_gat_attn_bwd", "dagnet_op_adj_matmul", "dagnet_op_disposition", "dagnet_op_sumsq",
    "dagnet_op_adam",
)

_bound = False


def _lib():
    global _bound
    lib = _capi.load()
    if not _bound:
        vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
        lib.dagnet_op_gemm.argtypes = [vp, i64, vp, i64, i32, vp, vp, i64, i32, i32, i32, i32, f32, i32, vp, vp, i64, i64, vp]
        lib.dagnet_op_gemm_dact.argtypes = [vp, i64, vp, i64, vp, i64, i32, f32, vp, i64, i32, vp, i64, i32, i32, i32, i32, vp]
        lib.dagnet_op_gemm_tn_workspace.argtypes = [i32, i32, i32]
        lib.dagnet_op_gemm_tn_workspace.restype = i64
        lib.dagnet_op_gemm_tn.argtypes = [vp, i64, vp, i64, vp, i64, vp, i32, i32, i32, vp, i32, vp]
        lib.dagnet_op_gemm_tn_partial.argtypes = [vp, i64, vp, i64, vp, i32, i32, i32, i32, vp, i32, vp]
        lib.dagnet_op_gemm_tn_finish.argtypes = [vp, vp, i64, vp, i32, i32, i32, i32, vp]
        lib.dagnet_op_act_bwd.argtypes = [i64, i32, vp, i64, vp, i64, vp, i64, i32, f32, vp]
        lib.dagnet_op_copy2d.argtypes = [i64, i32, vp, i64, vp, i64, i32, vp]
        lib.dagnet_op_onehot_grad.argtypes = [i64, i32, i32, vp, i64, vp, vp, i64, vp, i32, vp]
        lib.dagnet_op_softmax_fwd.argtypes = [i64, i32, vp, i64, vp, i64, vp]
        lib.dagnet_op_softmax_bwd.argtypes = [i64, i32, vp, i64, vp, i64, vp, i64, vp]
        lib.dagnet_op_gru_fwd.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp]
        lib.dagnet_op_gru_bwd.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.dagnet_op_gru_layer_fwd.argtypes = [i64, i32, vp, i64, i32, vp, i64, i32, vp, i64, vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp]
        lib.dagnet_op_mlp2_fwd.argtypes = [i64, i32, vp, vp, vp, vp, vp, i64, vp, i32, vp, i32, i32, C.c_float, vp, i64, vp, i32, i32, C.c_float,
                                           vp, i64, vp, i64, vp]
        lib.dagnet_op_mlp_chain_fwd.argtypes = [i64, i32, vp, vp, vp, vp, vp, i64, vp, i32, vp, i32, i32, C.c_float, vp, i64, vp, i32, i32, C.c_float,
                                                vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, i32, vp, i64, vp, i64, vp]
        f32 = C.c_float
        lib.dagnet_op_mlp2_bwd.argtypes = [i64, vp, i64, vp, i64, vp, i64, i32, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i32, f32, vp, i64, i32, f32,
                                           vp, i64, vp, i64, vp, i64, i32, i32, i32, vp, i64, i32, i32, i32, vp]
        lib.dagnet_op_linear_pieces_bwd.argtypes = [i64, vp, i64, vp, i64, vp, i64, i32, i32, i32, vp, i64, i32, i32, i32, vp]
        lib.dagnet_op_gru_pack.argtypes = [i32, i32, i32, vp, i64, vp, vp, i64, vp, vp, vp]
        lib.dagnet_op_gru_pack_floats.argtypes = [i32, i32, i32]
        lib.dagnet_op_gru_pack_floats.restype = i64
        lib.dagnet_op_gru_pack_bwd.argtypes = [i32, i32, i32, vp, i64, vp, i64, vp, vp]
        lib.dagnet_op_gru_pack_bwd_floats.argtypes = [i32, i32, i32]
        lib.dagnet_op_gru_pack_bwd_floats.restype = i64
        lib.dagnet_op_gru_layer_bwd.argtypes = [i64, i32, i32, i32, vp, i64, vp, vp, vp, i64, vp, vp, vp, vp, i64, i32, vp, i64, i32, vp, i64, i32, vp]
        lib.dagnet_op_vae_fwd.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.dagnet_op_vae_bwd.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp]
        lib.dagnet_op_nll_fwd.argtypes = [i64, i32, vp, vp, vp, vp, vp]
        lib.dagnet_op_nll_bwd.argtypes = [i64, i32, vp, vp, vp, vp, f32, vp, vp, vp]
        lib.dagnet_op_ce_fwd.argtypes = [i64, vp, i64, vp, vp, vp]
        lib.dagnet_op_ce_bwd.argtypes = [i64, i32, vp, vp, vp, i64, vp]
        lib.dagnet_op_softmax_ce_bwd.argtypes = [i64, i32, vp, i64, vp, vp, vp, i64, vp]
        lib.dagnet_op_sum.argtypes = [i64, vp, f32, vp, i32, vp]
        lib.dagnet_op_col_workspace.argtypes = [i64, i32]
        lib.dagnet_op_col_workspace.restype = i64
        lib.dagnet_op_colsum.argtypes = [i64, i32, vp, i64, vp, i32, vp, vp]
        lib.dagnet_op_colsum_static.argtypes = [i64, i32, vp, i64, vp, i32, vp]
        lib.dagnet_op_bn_fwd.argtypes = [i64, i32, vp, i64, vp, vp, f32, f32, i32, vp, i64, vp, vp, vp, vp, vp, vp]
        lib.dagnet_op_bn_bwd.argtypes = [i64, i32, vp, i64, vp, vp, vp, i64, vp, i64, i32, vp, i64, vp, vp, i32, vp, vp, vp]
        lib.dagnet_op_bn_moments.argtypes = [i64, i32, vp, i64, vp, vp, vp, vp]
        lib.dagnet_op_bn_merge.argtypes = [i32, i32, vp, f32, f32, vp, vp, vp, vp, vp]
        lib.dagnet_op_bn_apply.argtypes = [i64, i32, vp, i64, vp, vp, vp, i32, vp, i64, vp]
        lib.dagnet_op_bn_bwd_sums.argtypes = [i64, i32, vp, i64, vp, vp, i64, vp, i64, i32, vp, vp, vp]
        lib.dagnet_op_bn_bwd_apply.argtypes = [i64, i64, i32, vp, i64, vp, vp, vp, i64, vp, i64, i32, vp, vp, i64, vp]
        lib.dagnet_op_gat_attn_fwd.argtypes = [vp, vp, i64, vp, i64, vp, i32, i32, f32, i32, vp, i64, vp]
        lib.dagnet_op_gat_attn_bwd.argtypes = [vp, vp, i64, vp, i64, vp, i32, i32, f32, i32, vp, i64, vp, i64, vp, i64, vp, vp]
        lib.dagnet_op_adj_matmul.argtypes = [vp, i32, vp, i64, vp, i64, vp, i32, vp, i64, vp]
        lib.dagnet_op_disposition.argtypes = [vp, i32, vp, vp, i64, vp]
        lib.dagnet_op_sumsq.argtypes = [i64, vp, vp, vp]
        lib.dagnet_op_adam.argtypes = [i64, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, i32, vp]
        _bound = True
    return lib


# ---------------------------------------------------------------------------------------------
# plumbing
# ---------------------------------------------------------------------------------------------
_scratch: Dict[tuple, torch.Tensor] = {}
_scratch_retired = []        # outgrown buffers stay alive: a captured hipGraph may have their address baked into its nodes


def _ws(device, nfloats: int) -> torch.Tensor:
    """Scratch per device AND stream (split-reduction partials).  Kernels of one stream run in order, so one buffer per stream
    is enough; the branches of a training step (Branches) run on streams of their own and must not share one.
    A buffer that is outgrown is retired, never freed (training.GraphedForwardBackward replays launches that hold its
    pointer; handing the memory back to the caching allocator would let a replay scribble over another tensor)."""
    key = (torch.device(device), torch.cuda.current_stream(device).cuda_stream)
    t = _scratch.get(key)
    if t is None or t.numel() < nfloats:
        if t is not None:
            _scratch_retired.append(t)
        t = torch.empty(max(int(nfloats), 1 << 20), dtype=torch.float32, device=device)
        _scratch[key] = t
    return t


def _m(t: torch.Tensor, what="operand"):
    """-> (pointer, row stride) of a 2-D fp32 device matrix with unit column stride."""
    if not t.is_cuda or t.dtype != torch.float32:
        raise DagnetHipError(f"{what} must be an fp32 tensor on a ROCm device (got {t.dtype} on {t.device}); there is no CPU path")
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise DagnetHipError(f"{what} must be 2-D with unit column stride (got shape {tuple(t.shape)}, strides {t.stride()})")
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)
    return C.c_void_p(t.data_ptr()), int(ld)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _adj_ld(adj: torch.Tensor) -> int:
    """Adjacency of one timestep: compact block-diagonal values (1-D) -> 0, dense [N, N] -> N."""
    if adj.dim() == 1:
        return 0
    if adj.dim() == 2 and adj.shape[0] == adj.shape[1] and adj.is_contiguous():
        return int(adj.shape[0])
    raise DagnetHipError(f"adjacency must be compact (1-D) or a contiguous dense [N, N] matrix, got shape {tuple(adj.shape)}")


def _rows(t: torch.Tensor) -> torch.Tensor:
    """A gradient as autograd hands it over -> usable matrix (expanded / non-unit-stride grads are materialised by our copy)."""
    if t.dim() == 2 and (t.shape[1] <= 1 or t.stride(1) == 1) and (t.shape[0] <= 1 or t.stride(0) >= t.shape[1]):
        return t
    return t.contiguous()


def set_fused(mask: int) -> int:
    """Select the fused kernels (bit 0 GRU layer forward, 1 dense blocks forward, 2 dense blocks backward, 3 split products on the
    16-bit matrix pipe in the wide GEMM layers and in the register-resident rollout kernels, 4 two fp16 terms instead of three
    bf16 ones in dagnet_op_gemm's wide layers - tests only; < 0: default = bits 0-3).  -> the previous mask."""
    return int(_lib().dagnet_op_set_fused(int(mask)))


def gemm(A, B, C_out, *, transb=False, bias=None, act=ACT_NONE, slope=0.01, accumulate=False, gather=None, gidx=None, gs_g=0, gs_n=0,
         K=None):
    """C = act([C +] A op(B) + bias + gather[gidx]) ;  op(B) = B^T for B [N,K] (nn.Linear), B for B [K,N] when transb."""
    (a, lda), (b, ldb), (c, ldc) = _m(A, "A"), _m(B, "B"), _m(C_out, "C")
    M, N = C_out.shape
    K = A.shape[1] if K is None else K
    _check(_lib().dagnet_op_gemm(a, lda, b, ldb, int(transb), _p(bias), c, ldc, M, N, K, act, slope, int(accumulate),
                                 gather, _p(gidx), gs_g, gs_n, _stream(A.device)))


def gemm_dact(dY, Y, dP, act, slope, B, C_out, *, transb=True, accumulate=False):
    """dP = dY * act'(Y);  C = [C +] dP op(B)  - the activation backward fused into the first input-gradient product."""
    (g, ldg), (y, ldy), (p, ldp), (b, ldb), (c, ldc) = _m(dY, "dY"), _m(Y, "Y"), _m(dP, "dP"), _m(B, "B"), _m(C_out, "C")
    M, N = C_out.shape
    K = dY.shape[1]
    _check(_lib().dagnet_op_gemm_dact(g, ldg, y, ldy, p, ldp, act, slope, b, ldb, int(transb), c, ldc, M, N, K, int(accumulate),
                                      _stream(dY.device)))


def gemm_tn(A, B, C_out, accumulate=False, dbias=None):
    """C[N,K] (+)= A[M,N]^T B[M,K];  dbias [N] (+)= column sums of A (same pass)."""
    (a, lda), (b, ldb), (c, ldc) = _m(A, "A"), _m(B, "B"), _m(C_out, "C")
    M, N, K = A.shape[0], A.shape[1], B.shape[1]
    ws = _ws(A.device, _lib().dagnet_op_gemm_tn_workspace(M, N, K))
    _check(_lib().dagnet_op_gemm_tn(a, lda, b, ldb, c, ldc, _p(dbias), M, N, K, _p(ws), int(accumulate), _stream(A.device)))


def colsum(X, out, accumulate=False):
    (x, ldx) = _m(X, "X")
    M, N = X.shape
    ws = _ws(X.device, _lib().dagnet_op_col_workspace(M, N))
    _check(_lib().dagnet_op_colsum(M, N, x, ldx, _p(out), int(accumulate), _p(ws), _stream(X.device)))


def colsum_static(X, out, accumulate=False):
    """colsum without a workspace tensor: the partial sums live in the library's static device array (at most 1024 columns;
    nothing is allocated, so the call is safe under hipGraph capture).  Calls on one GPU must not overlap."""
    (x, ldx) = _m(X, "X")
    M, N = X.shape
    _check(_lib().dagnet_op_colsum_static(M, N, x, ldx, _p(out), int(accumulate), _stream(X.device)))


def copy2d(src, dst, accumulate=False):
    (s, lds), (d, ldd) = _m(src, "src"), _m(dst, "dst")
    _check(_lib().dagnet_op_copy2d(src.shape[0], src.shape[1], s, lds, d, ldd, int(accumulate), _stream(src.device)))


class StepPool:
    """Stacked storage for one training step.  Every operator call of a time step is a *site* (numbered in call order, the
    same in every step); what a site produces for step t - its output in the forward pass, its gradients in the backward
    pass - is slice t of one persistent [T, ...] buffer per (site, tag).  Consequences:
      * the saved input and the pre-activation gradient of a layer are contiguous over time, so its weight gradient is ONE
        [T*M, N]^T [T*M, K] product at the end of the backward pass (DeferredGrads) instead of one product per step;
      * no allocator traffic inside the step, and fixed addresses for hipGraph capture.
    Inactive (no training.forward_train in flight): plain torch.empty.

    ONE step in flight: the buffers are reused by the next train-mode forward, so the backward of a graph must run before
    another forward_train begins (gradient accumulation over two forwards, a logging forward in train mode ... would read
    the second forward's activations: the kernels write through raw pointers, autograd's version counters see nothing).
    Every slot carries the generation of the begin() that issued it; a backward that arrives with a stale one raises."""

    def __init__(self):
        self.bufs = {}
        self.T, self.t, self.n_site, self.active, self.device = 0, 0, 0, False, None
        self.gen = 0

    def begin(self, T, device):
        """Sites reserved before the first next_t() are per-step (one slot): stacked weights, loop invariants."""
        self.T, self.t, self.n_site, self.active, self.device = int(T), -1, 0, True, torch.device(device)
        self.gen += 1
        self.bn_calls = {}
        BR.keep.clear()

    def next_t(self, t):
        self.t, self.n_site = int(t), 0

    def end(self):
        self.active = False
        for bn, n in getattr(self, "bn_calls", {}).items():     # nn.BatchNorm1d counts its train-mode calls
            bn.num_batches_tracked += n
        self.bn_calls = {}

    def site(self):
        """Reserve the next site of the current time step -> slot (site, t), or None when the pool is not active."""
        if not self.active:
            return None
        self.n_site += 1
        if self.t < 0:
            return (("pre", self.n_site - 1), 0, 1, self.gen)
        return (self.n_site - 1, self.t, self.T, self.gen)

    def at(self, slot, tag, shape, like=None, dtype=torch.float32):
        shape = tuple(int(v) for v in shape)
        if slot is None:
            return torch.empty(shape, dtype=dtype, device=like.device if like is not None else self.device)
        if slot[3] != self.gen:
            raise RuntimeError("a train-mode forward has begun since this graph was built: its saved activations live in the "
                               "step pool (dagnet_amd/_ops.StepPool) and have been overwritten - run backward() before the next "
                               "train-mode forward (one training step in flight)")
        # rows of wide matrices are padded to 16 bytes (g_dim = 90 -> row stride 92): float4 rows for the skinny GEMM
        pad = len(shape) == 2 and shape[1] > 16 and (shape[1] & 3) != 0
        full = (shape[0], (shape[1] + 3) & ~3) if pad else shape
        n = 1
        for v in full:
            n *= v
        key = (slot[0], tag, dtype, self.device)
        flat = self.bufs.get(key)
        if flat is None or flat.numel() < slot[2] * n:
            if flat is not None:
                _scratch_retired.append(flat)           # a captured graph may still hold its address
            flat = self.bufs[key] = torch.empty(slot[2] * max(n, 1), dtype=dtype, device=self.device)
        out = flat[slot[1] * n:(slot[1] + 1) * n].view(full)
        return out[:, :shape[1]] if pad else out

    def release(self):
        self.bufs.clear()


POOL = StepPool()


class Branches:
    """Side streams for the independent branches of a training step (training.forward_train): the goal head and the decoder / NLL
    of a time step do not feed the recurrence, so they run beside the encoder -> phi_z -> GRU -> graph chain instead of in line
    with it (the prior could too, but see `inline`).  Every kernel of the step moves ~40 MB in ~17 us and leaves the chip half idle at its
    head and tail; two or three such chains side by side fill those gaps (tools/ubench/graph_branches.py: 3 chains of 12
    GEMMs 598 -> 395 us, also as branches of a captured hipGraph).  The backward pass follows by itself: autograd runs a
    node's backward on the stream of its forward and orders the streams along the edges it knows.  What it does NOT know are
    this module's side channels - fan-out gradient buffers (_GradAcc: used only by consumers on the alias's own stream),
    stacked-weight and deferred weight-gradient records (their collectors join every side stream first), per-stream scratch (_ws).
    DAGNET_TRAIN_STREAMS=0 keeps everything on one stream."""

    def __init__(self):
        self.enabled = os.environ.get("DAGNET_TRAIN_STREAMS", "1") != "0"
        # branches kept on the main stream after all: the prior sits on the loop-carried path (the latent draw needs it going forward,
        # d h(t-1) needs its backward right after the VAE's), where two stream crossings per time step cost more than the overlap
        # gives (measured: 117.8 ms forked, 115.8 inline; the decoder either way; the goal head inline: 126.1)
        self.inline = set(filter(None, os.environ.get("DAGNET_INLINE_BRANCHES", "prior").split(",")))
        self.side = {}
        self.keep = []           # events of the step in flight: one that is destroyed while the hipGraph capture that recorded it is
                                 # still open takes the runtime down at capture end, so they live until the next step begins

    def order(self, waiter, producer):
        """``waiter`` continues after what ``producer`` has been given so far (stream.wait_stream with an event that stays alive)."""
        ev = torch.cuda.Event()
        ev.record(producer)
        waiter.wait_event(ev)
        self.keep.append(ev)

    def _stream(self, name, device):
        key = (name, torch.device(device))
        st = self.side.get(key)
        if st is None:
            st = self.side[key] = torch.cuda.Stream(device=device)
        return st

    @contextlib.contextmanager
    def fork(self, name, device):
        """Run the block on side stream ``name``, ordered after everything issued on the current stream so far."""
        if not (self.enabled and POOL.active) or name in self.inline:
            yield
            return
        st = self._stream(name, device)
        self.order(st, torch.cuda.current_stream(device))
        with torch.cuda.stream(st):
            yield

    def join(self, *names, device=None):
        """The current stream waits for what the named side streams (default: all) have been given so far."""
        if not self.enabled:
            return
        for (name, dev), st in self.side.items():
            if (not names or name in names) and (device is None or dev == torch.device(device)):
                cur = torch.cuda.current_stream(dev)
                if cur != st:
                    self.order(cur, st)


BR = Branches()


class DeferredGrads:
    """Weight / bias gradients of sink parameters (training.attach_grad_arena).  The backward pass only RECORDS the operand
    pairs (dY_t, X_t) of every layer; at the end of the pass (autograd engine callback) the records of a layer are grouped
    into runs that are contiguous over time (StepPool slices) and every run is one weight-gradient GEMM over all its steps
    (a record that is not part of a run falls back to a product of its own)."""

    def __init__(self):
        self.recs = {}
        self.pending = False

    def add(self, key, A, B, dst, gidx=None, bias_dst=None):
        """dst[N,K] (a view of the gradient arena) += A[M,N]^T B[M,K]  (B None: one-hot from gidx); bias_dst[N] += colsum(A)."""
        M, N = A.shape
        rec = self.recs.get(key)
        if rec is None:
            rec = self.recs[key] = dict(dst=dst, bias=bias_dst, items=[])
        rec["items"].append((A, B, gidx))
        if not self.pending:
            self.pending = True
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)

    def reset(self):
        """Forget an accumulation that never finished (a backward pass that raised drops the engine callback)."""
        self.pending = False
        self.recs.clear()

    def flush(self, key):
        rec = self.recs.pop(key, None)
        if rec is None or not rec["items"]:
            return
        flush_products(rec["items"], rec["dst"], rec["bias"], True)

    def finish(self):
        BR.join()                # the records' operands were written on the branches' streams
        for key in list(self.recs):
            self.flush(key)
        self.pending = False


def flush_products(items, dst, bias, accumulate):
    """dst[N,K] (+)= sum over the records (A [M,N], B [M,K] | None, gidx | None) of A^T B, bias[N] (+)= column sums of A.
    Records whose operands are consecutive slices of stacked matrices (StepPool) are multiplied as ONE product."""
    lib = _lib()
    (c, ldc) = _m(dst, "gradient")
    K = dst.shape[1]
    i = 0
    while i < len(items):
        A, B, gidx = items[i]
        M, N = A.shape
        lda = A.stride(0) if M > 1 else N
        ldb = (B.stride(0) if M > 1 else B.shape[1]) if B is not None else 0
        j, sign = i + 1, 0
        while j < len(items):                       # extend the run while every operand advances by exactly one slice
            A2, B2, g2 = items[j]
            if A2.shape != A.shape or (B2 is None) != (B is None) or (B is not None and (B2.shape != B.shape or B2.stride(0) != B.stride(0))) \
                    or A2.stride(0) != A.stride(0):
                break
            A1, B1, g1 = items[j - 1]
            dA = A2.data_ptr() - A1.data_ptr()
            sg = 1 if dA > 0 else -1
            if abs(dA) != M * lda * 4 or (sign and sg != sign):
                break
            if B is not None and B2.data_ptr() - B1.data_ptr() != sg * M * ldb * 4:
                break
            if gidx is not None and g2.data_ptr() - g1.data_ptr() != sg * M * 4:
                break
            sign = sg
            j += 1
        n = j - i
        base = items[i] if sign >= 0 else items[j - 1]          # the lowest address of the run
        Mt = n * M
        a_ptr = C.c_void_p(base[0].data_ptr())
        acc = int(bool(accumulate) or i > 0)
        if B is not None:
            ws = _ws(A.device, lib.dagnet_op_gemm_tn_workspace(Mt, N, K))
            _check(lib.dagnet_op_gemm_tn(a_ptr, lda, C.c_void_p(base[1].data_ptr()), ldb, c, ldc, _p(bias), Mt, N, K, _p(ws), acc,
                                         _stream(A.device)))
        else:
            ws = _ws(A.device, lib.dagnet_op_gemm_tn_workspace(Mt, N, K))
            _check(lib.dagnet_op_onehot_grad(Mt, N, K, a_ptr, lda, C.c_void_p(base[2].data_ptr()), c, ldc, _p(ws), acc, _stream(A.device)))
        i = j


def flush_colsums(items, out, accumulate):
    """out[N] (+)= column sums of every [M,N] matrix in ``items``; consecutive StepPool slices are reduced as one matrix."""
    lib = _lib()
    i = 0
    while i < len(items):
        X = items[i]
        M, N = X.shape
        j = i + 1
        sign = 0
        while j < len(items) and items[j].shape == X.shape and items[j].is_contiguous() and X.is_contiguous():
            d = items[j].data_ptr() - items[j - 1].data_ptr()
            sg = 1 if d > 0 else -1
            if abs(d) != M * N * 4 or (sign and sg != sign):
                break
            sign = sg
            j += 1
        base = items[i] if sign >= 0 else items[j - 1]
        Mt = (j - i) * M if X.is_contiguous() else M
        ws = _ws(X.device, lib.dagnet_op_col_workspace(Mt, N))
        _check(lib.dagnet_op_colsum(Mt, N, C.c_void_p(base.data_ptr()), X.stride(0) if M > 1 else N, _p(out), int(bool(accumulate) or i > 0),
                                    _p(ws), _stream(X.device)))
        i = j


class _WAcc:
    """Gradient bookkeeping of a stacked, non-leaf weight (StackFn output) that is read once per time step: its consumers
    record their operand pairs, StackFn.backward multiplies them in one go."""

    def __init__(self):
        self.products, self.colsums = [], []

    def add(self, x, dy):               # d(stacked W [K,N]) += x[M,K]^T dy[M,N]
        self.products.append((x, dy, None))

    def add_rows(self, part):           # d(stacked a [H,2F]) += column sums of part [S, H*2F]
        self.colsums.append(part)


def _defer(p: torch.Tensor):
    return getattr(p, "_hip_defer", None)


def _sink(p: torch.Tensor):
    """Gradient arena view of a parameter (training.attach_grad_arena): the backward kernels then accumulate straight into
    p.grad and hand autograd None, which saves one zero-fill and one torch add per parameter use and time step."""
    if p is None or not p.is_leaf or not getattr(p, "_hip_sink", False):
        return None
    g = p.grad
    if g is not None and getattr(p, "_hip_sink", False) and g.is_contiguous() and g.dtype == torch.float32 and g.shape == p.shape:
        return g
    return None


def _empty(M, N, like):
    return torch.empty((M, N), dtype=torch.float32, device=like.device)


# ---------------------------------------------------------------------------------------------
# dense layers
# ---------------------------------------------------------------------------------------------
_LAZY_LINEAR = False         # set by gru_step around the two gate products whose values the fused layer kernel produces itself


class LinearFn(torch.autograd.Function):
    """y = act(sum_s x_s W[:, c_s : c_s + K_s]^T + b + W[:, gcol + gidx])  -- an nn.Linear whose input is the
    concatenation of dense pieces and (optionally) one one-hot piece given by its index (never materialised)."""

    @staticmethod
    def forward(ctx, W, b, act, slope, cols, gidx, gcol, gdim, *xs):
        N, Ktot = W.shape
        M = xs[0].shape[0]
        ctx.slot = POOL.site()
        y = POOL.at(ctx.slot, "y", (M, N), xs[0])
        ctx.gaccs = [_gacc_of(x) for x in xs]
        last = len(xs) - 1
        for s, (x, c0) in enumerate(() if _LAZY_LINEAR else zip(xs, cols)):       # lazy: the consumer computes y (GruLayerFn)
            Ks = x.shape[1]
            Wp = W[:, c0:c0 + Ks]
            fin = s == last
            if fin and gidx is not None:
                gemm(x, Wp, y, bias=b, act=act, slope=slope, accumulate=s > 0,
                     gather=C.c_void_p(W.data_ptr() + 4 * gcol), gidx=gidx, gs_g=1, gs_n=Ktot)
            else:
                gemm(x, Wp, y, bias=b if fin else None, act=act if fin else ACT_NONE, slope=slope, accumulate=s > 0)
        ctx.save_for_backward(W, y, gidx, *xs)
        ctx.W_param, ctx.b_param = W, b          # the leaf parameters themselves (for their gradient sinks)
        ctx.meta = (act, slope, tuple(cols), gcol, gdim, b is not None)
        ctx.set_materialize_grads(False)       # a lazy node may get no gradient at all (its block's backward ran as one kernel)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:                   # a lazy node whose block ran its backward as one kernel (MlpChainFn)
            return (None,) * (8 + len(ctx.saved_tensors) - 3)
        W, y, gidx, *xs = ctx.saved_tensors
        act, slope, cols, gcol, gdim, has_b = ctx.meta
        N, Ktot = W.shape
        M = y.shape[0]
        dy = _rows(dy)
        # the activation backward rides on the first input-gradient product (one pass over dy and y instead of two)
        fuse_at = -1
        if act != ACT_NONE:
            dpre = POOL.at(ctx.slot, "dpre", (M, N), y)
            fuse_at = next((i for i in range(len(xs)) if ctx.needs_input_grad[8 + i]), -1)
            if fuse_at < 0:
                (g, ldg), (yy, ldy), (dp, ldp) = _m(dy), _m(y), _m(dpre)
                _check(_lib().dagnet_op_act_bwd(M, N, g, ldg, yy, ldy, dp, ldp, act, slope, _stream(y.device)))
        else:
            dpre = dy
        dxs = [None] * len(xs)
        if fuse_at >= 0:
            x, c0 = xs[fuse_at], cols[fuse_at]
            gacc = ctx.gaccs[fuse_at]
            if gacc is not None:
                gemm_dact(dy, y, dpre, act, slope, W[:, c0:c0 + x.shape[1]], gacc.buffer(x), accumulate=gacc.started)
                gacc.done()
            else:
                dxs[fuse_at] = POOL.at(ctx.slot, "dx%d" % fuse_at, (M, x.shape[1]), y)
                gemm_dact(dy, y, dpre, act, slope, W[:, c0:c0 + x.shape[1]], dxs[fuse_at])
        dW = db = None
        if ctx.needs_input_grad[0]:
            sink = _sink(ctx.W_param)
            dW = sink if sink is not None else torch.zeros_like(W)
            acc = sink is not None
            # the bias gradient rides on the first weight-gradient GEMM (same pass over dpre) when both accumulate alike
            bsink = _sink(ctx.b_param) if has_b and ctx.needs_input_grad[1] else None
            fused_b = None
            if has_b and ctx.needs_input_grad[1] and (bsink is not None) == acc:
                fused_b = bsink if acc else torch.empty(N, dtype=torch.float32, device=y.device)
            defer = _defer(ctx.W_param) if acc else None
            for s_, (x, c0) in enumerate(zip(xs, cols)):
                dst = dW[:, c0:c0 + x.shape[1]]
                if defer is not None:
                    defer.add((id(ctx.W_param), c0), dpre, x, dst, bias_dst=fused_b if s_ == 0 else None)
                else:
                    gemm_tn(dpre, x, dst, accumulate=acc, dbias=fused_b if s_ == 0 else None)
            if gidx is not None and defer is not None:
                defer.add((id(ctx.W_param), gcol), dpre, None, dW[:, gcol:gcol + gdim], gidx=gidx)
            elif gidx is not None:
                (dp, ldp) = _m(dpre)
                ws = _ws(y.device, _lib().dagnet_op_gemm_tn_workspace(M, N, gdim))
                _check(_lib().dagnet_op_onehot_grad(M, N, gdim, dp, ldp, _p(gidx), C.c_void_p(dW.data_ptr() + 4 * gcol), Ktot, _p(ws),
                                                    int(acc), _stream(y.device)))
            if acc:
                dW = None
        if has_b and ctx.needs_input_grad[1] and ctx.needs_input_grad[0] and fused_b is not None:
            db = None if acc else fused_b
        elif has_b and ctx.needs_input_grad[1]:
            sink = _sink(ctx.b_param)
            if sink is not None:
                colsum(dpre, sink, accumulate=True)
            else:
                db = torch.empty(N, dtype=torch.float32, device=y.device)
                colsum(dpre, db)
        for i, (x, c0) in enumerate(zip(xs, cols)):
            if getattr(ctx, "dx_done", False):      # the consumer's backward kernel produced the input gradients (GruLayerFn)
                break
            if ctx.needs_input_grad[8 + i] and i != fuse_at:
                gacc = ctx.gaccs[i]
                if gacc is not None:            # a fan-out input (FanoutFn): add into the shared gradient buffer of the tensor
                    gemm(dpre, W[:, c0:c0 + x.shape[1]], gacc.buffer(x), transb=True, K=N, accumulate=gacc.started)
                    gacc.done()
                    continue
                dxs[i] = POOL.at(ctx.slot, "dx%d" % i, (M, x.shape[1]), y)
                gemm(dpre, W[:, c0:c0 + x.shape[1]], dxs[i], transb=True, K=N)
        return (dW, db, None, None, None, None, None, None, *dxs)


class MlpChainFn(torch.autograd.Function):
    """Values of a block of dense layers computed from the block's INPUTS by dagnet_op_mlp_chain_fwd (one kernel for the models'
    64-wide blocks: csrc/mlp_fused.hip) into the outputs of the layers' LinearFn nodes, which were created lazily (allocated,
    not computed) and keep the autograd graph - the backward pass is the unfused one.  Layers: first (y1), optional second (y2),
    optional pair of heads on y2 (mu, lv).  -> aliases of (the last of y1 / y2[, mu, lv])."""

    @staticmethod
    def forward(ctx, y1, y2, mu, lv, l1, l2, heads, act, slope, cols, gidx, gcol, *xs):
        gcol, ctx.gdim = gcol if isinstance(gcol, tuple) else (gcol, 0)
        M = y1.shape[0]
        n = len(xs)
        for x in xs:
            if x.dim() != 2 or x.stride(1) != 1 or x.dtype != torch.float32:
                raise DagnetHipError("input pieces must be fp32 matrices with unit column stride")
        ptr = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        ldx = (C.c_int64 * n)(*[(x.stride(0) if M > 1 else x.shape[1]) for x in xs])
        ks = (C.c_int32 * n)(*[x.shape[1] for x in xs])
        cs = (C.c_int32 * n)(*[int(c) for c in cols])
        (W1, b1) = l1
        (w1, ldw1), (p1, ld1) = _m(W1, "W1"), _m(y1, "y1")
        (W2, b2) = l2 if l2 is not None else (None, None)
        (w2, ldw2), (p2, ld2) = (_m(W2, "W2"), _m(y2, "y2")) if l2 is not None else ((None, 0), (None, 0))
        if heads is not None:
            (Wm, bm), (Wl, bl) = heads
            (wm, ldwm), (wl, ldwl), (pm, ldm), (pl, ldl) = _m(Wm, "W_mean"), _m(Wl, "W_logvar"), _m(mu, "mu"), _m(lv, "lv")
            Z = Wm.shape[0]
        else:
            wm = wl = pm = pl = bm = bl = None
            ldwm = ldwl = ldm = ldl = Z = 0
        _check(_lib().dagnet_op_mlp_chain_fwd(M, n, ptr, ldx, ks, cs, w1, ldw1, _p(b1), W1.shape[0], _p(gidx), int(gcol), act[0], slope, w2, ldw2,
                                              _p(b2), W2.shape[0] if W2 is not None else 0, act[1], slope, p1, ld1, p2, ld2, wm, ldwm, _p(bm),
                                              wl, ldwl, _p(bl), Z, pm, ldm, pl, ldl, _stream(y1.device)))
        ctx.n_in = 12 + n
        ctx.has = (y2 is not None, heads is not None)
        ctx.set_materialize_grads(False)
        if y2 is None and act[0] == ACT_NONE and gidx is None:      # one Linear without activation: its pieces' gradients in one kernel
            ctx.slot = POOL.site()
            ctx.one = (l1, tuple(cols))
            ctx.xs = xs
            ctx.gaccs = [_gacc_of(x) for x in xs]
        if y2 is not None:               # what the one-kernel backward needs (the unfused one keeps its own records in the lazy nodes)
            ctx.slot = POOL.site()
            ctx.blk = (l1, l2, heads, act, slope, tuple(cols), gidx, gcol, mu.shape[1] if heads is not None else 0)
            ctx.xs, ctx.y12 = xs, (y1, y2)
            ctx.gaccs = [_gacc_of(x) for x in xs]
        last = y2 if y2 is not None else y1
        if heads is not None:
            return last.view_as(last), mu.view_as(mu), lv.view_as(lv)
        return last.view_as(last)

    @staticmethod
    def backward(ctx, *g):
        two, heads = ctx.has
        out = [None] * ctx.n_in
        if two and MlpChainFn._fused_backward(ctx, g, out):
            return tuple(out)
        if not two and getattr(ctx, "one", None) is not None and MlpChainFn._fused_backward_one(ctx, g[0], out):
            return tuple(out)
        out[1 if two else 0] = g[0]
        if heads:
            out[2], out[3] = g[1], g[2]
        return tuple(out)

    @staticmethod
    def _fused_backward_one(ctx, dy, out):
        """Single Linear without activation over its pieces: both input gradients from one pass over dy (dagnet_op_linear_pieces_bwd);
        the weight gradient is recorded as LinearFn.backward records it (dpre = dy)."""
        (W, b), cols = ctx.one
        xs = ctx.xs
        if dy is None or _sink(W) is None or _defer(W) is None or (b is not None and _sink(b) is None):
            return False
        want = [i for i in range(len(xs)) if ctx.needs_input_grad[12 + i]]
        if not 1 <= len(want) <= 2 or any(xs[i].shape[1] % 4 or xs[i].shape[1] > 64 for i in want):
            return False
        dy = _rows(dy)
        M = dy.shape[0]
        dst = []
        for i in want:
            gacc = ctx.gaccs[i]
            if gacc is not None:
                dst.append((gacc.buffer(xs[i]), int(gacc.started), gacc))
            else:
                dst.append((POOL.at(ctx.slot, "dx%d" % i, tuple(xs[i].shape), dy), 0, None))
        pa = [(None, 0, 0, 0, 0), (None, 0, 0, 0, 0)]
        for j, i in enumerate(want):
            (ptr, ld) = _m(dst[j][0], "dx")
            pa[j] = (ptr, ld, int(cols[i]), xs[i].shape[1], dst[j][1])
        (pd, ldd), (w, ldw) = _m(dy, "dy"), _m(W, "W")
        rc = _lib().dagnet_op_linear_pieces_bwd(M, pd, ldd, w, ldw, *pa[0], *pa[1], _stream(dy.device))
        if rc == 1:
            return False
        _check(rc)
        for j, i in enumerate(want):
            if dst[j][2] is not None:
                dst[j][2].done()
            else:
                out[12 + i] = dst[j][0]
        sink, bs = _sink(W), (_sink(b) if b is not None else None)
        for s_, (x, c0) in enumerate(zip(xs, cols)):
            _defer(W).add((id(W), c0), dy, x, sink[:, c0:c0 + x.shape[1]], bias_dst=bs if s_ == 0 else None)
        return True

    @staticmethod
    def _fused_backward(ctx, g, out):
        """One kernel from (dmu, dlv | dy2) to the pieces' input gradients; the weight gradients are recorded for the deferred
        products exactly as LinearFn.backward records them.  -> False (nothing done) when a piece of the bookkeeping is missing
        (no gradient arena, an input piece whose gradient has nowhere to go, an unsupported shape)."""
        (W1, b1), (W2, b2), hd, act, slope, cols, gidx, gcol, Z = ctx.blk
        xs, (y1, y2) = ctx.xs, ctx.y12
        heads = hd is not None
        layers = [(W1, b1), (W2, b2)] + ([hd[0], hd[1]] if heads else [])
        for (W, b) in layers:
            if _sink(W) is None or _defer(W) is None or (b is not None and _sink(b) is None):
                return False
        if heads:
            dmu, dlv = g[1], g[2]
            if dmu is None or dlv is None or g[0] is not None:
                return False
            dmu, dlv = _rows(dmu), _rows(dlv)
        else:
            if g[0] is None:
                return False
            dy2 = _rows(g[0])
        M = y1.shape[0]
        want = [i for i in range(len(xs)) if ctx.needs_input_grad[12 + i]]
        if len(want) > 2 or any(xs[i].shape[1] % 4 or xs[i].shape[1] > 64 for i in want):
            return False
        dp2, dp1 = POOL.at(ctx.slot, "dpre2", (M, 64), y1), POOL.at(ctx.slot, "dpre1", (M, 64), y1)
        dst = []
        for i in want:
            gacc = ctx.gaccs[i]
            if gacc is not None:
                dst.append((gacc.buffer(xs[i]), int(gacc.started), gacc))
            else:
                dst.append((POOL.at(ctx.slot, "dx%d" % i, tuple(xs[i].shape), y1), 0, None))
        pa = [(None, 0, 0, 0, 0), (None, 0, 0, 0, 0)]
        for j, i in enumerate(want):
            (ptr, ld) = _m(dst[j][0], "dx")
            pa[j] = (ptr, ld, int(cols[i]), xs[i].shape[1], dst[j][1])
        (w1, ldw1), (w2, ldw2), (py1, ldy1), (py2, ldy2), (pd2, ld2), (pd1, ld1) = _m(W1), _m(W2), _m(y1), _m(y2), _m(dp2), _m(dp1)
        if heads:
            (wm, ldwm), (wl, ldwl), (pmu, ldmu), (plv, ldlv) = _m(hd[0][0]), _m(hd[1][0]), _m(dmu), _m(dlv)
            pdy, lddy = None, 0
        else:
            wm = wl = pmu = plv = None
            ldwm = ldwl = ldmu = ldlv = 0
            (pdy, lddy) = _m(dy2)
        rc = _lib().dagnet_op_mlp2_bwd(M, pdy, lddy, pmu, ldmu, plv, ldlv, Z, wm, ldwm, wl, ldwl, py2, ldy2, py1, ldy1, w2, ldw2, act[1], slope,
                                       w1, ldw1, act[0], slope, pd2, ld2, pd1, ld1, *pa[0], *pa[1], _stream(y1.device))
        if rc == 1:
            return False
        _check(rc)
        for j, i in enumerate(want):
            if dst[j][2] is not None:
                dst[j][2].done()
            else:
                out[12 + i] = dst[j][0]

        sink1, defer1 = _sink(W1), _defer(W1)
        bs1 = _sink(b1) if b1 is not None else None
        for s_, (x, c0) in enumerate(zip(xs, cols)):
            defer1.add((id(W1), c0), dp1, x, sink1[:, c0:c0 + x.shape[1]], bias_dst=bs1 if s_ == 0 else None)
        if gidx is not None:
            gdim = ctx.gdim
            defer1.add((id(W1), gcol), dp1, None, sink1[:, gcol:gcol + gdim], gidx=gidx)
        _defer(W2).add((id(W2), 0), dp2, y1, _sink(W2), bias_dst=_sink(b2) if b2 is not None else None)
        if heads:
            for (W, b), d in ((hd[0], dmu), (hd[1], dlv)):
                _defer(W).add((id(W), 0), d, y2, _sink(W), bias_dst=_sink(b) if b is not None else None)
        return True


def mlp2(seq, xs, cols, act=ACT_LRELU, slope=0.01, gidx=None, gcol=0, gdim=0, heads=None):
    """seq = nn.Sequential(Linear, act, Linear, act) applied to [x_0 | ... | one-hot(gidx)]: the LinearFn nodes are only recorded,
    their values come from one dagnet_op_mlp_chain_fwd call.  heads = (mean, logvar) nn.Linear modules that read the block's output:
    -> (fan-out alias of the output, mean, logvar), all three from the same call; else -> the output."""
    global _LAZY_LINEAR
    l1, l2 = seq[0], seq[2]
    _LAZY_LINEAR = True
    try:
        y1 = LinearFn.apply(l1.weight, l1.bias, act, slope, tuple(cols), gidx, gcol, gdim, *xs)
        y2 = LinearFn.apply(l2.weight, l2.bias, act, slope, (0,), None, 0, 0, y1)
        if heads is not None:
            out = fanout(y2)
            mu = LinearFn.apply(heads[0].weight, heads[0].bias, ACT_NONE, 0.0, (0,), None, 0, 0, out)
            lv = LinearFn.apply(heads[1].weight, heads[1].bias, ACT_NONE, 0.0, (0,), None, 0, 0, out)
    finally:
        _LAZY_LINEAR = False
    w = ((l1.weight, l1.bias), (l2.weight, l2.bias))
    if heads is None:
        return MlpChainFn.apply(y1, y2, None, None, w[0], w[1], None, (act, act), slope, tuple(cols), gidx, (gcol, gdim), *xs)
    hw = ((heads[0].weight, heads[0].bias), (heads[1].weight, heads[1].bias))
    _, mu_a, lv_a = MlpChainFn.apply(y1, y2, mu, lv, w[0], w[1], hw, (act, act), slope, tuple(cols), gidx, (gcol, gdim), *xs)
    return out, mu_a, lv_a


def linear_pieces(mod, xs, cols, act=ACT_NONE, slope=0.01):
    """nn.Linear over [x_0 | x_1 | ...] as ONE kernel (instead of one accumulating product per piece)."""
    global _LAZY_LINEAR
    _LAZY_LINEAR = True
    try:
        y1 = LinearFn.apply(mod.weight, mod.bias, act, slope, tuple(cols), None, 0, 0, *xs)
    finally:
        _LAZY_LINEAR = False
    return MlpChainFn.apply(y1, None, None, None, (mod.weight, mod.bias), None, None, (act, ACT_NONE), slope, tuple(cols), None, 0, *xs)


def linear(mod, xs: Sequence[torch.Tensor], cols: Sequence[int], act=ACT_NONE, slope=0.01, gidx=None, gcol=0, gdim=0):
    """nn.Linear ``mod`` applied to [x_0 | ... | one-hot(gidx)] with x_s at input columns cols[s] and the one-hot block (gdim wide)
    at column gcol."""
    return LinearFn.apply(mod.weight, mod.bias, act, slope, tuple(cols), gidx, gcol, gdim, *xs)


class MatmulFn(torch.autograd.Function):
    """x @ W with W [K,N] (the graph layers store their weights input-major)."""

    @staticmethod
    def forward(ctx, x, W):
        ctx.slot = POOL.site()
        y = POOL.at(ctx.slot, "y", (x.shape[0], W.shape[1]), x)
        gemm(x, W, y, transb=True)
        ctx.save_for_backward(x, W)
        ctx.W_param = W
        ctx.gacc = _gacc_of(x)
        ctx.wacc = getattr(W, "_hip_wacc", None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = _rows(dy)
        dx = dW = None
        if ctx.needs_input_grad[0]:
            if ctx.gacc is not None:
                gemm(dy, W, ctx.gacc.buffer(x), accumulate=ctx.gacc.started)
                ctx.gacc.done()
            else:
                dx = POOL.at(ctx.slot, "dx", tuple(x.shape), x)
                gemm(dy, W, dx)                       # dy [M,N] . W[K,N]^T
        if ctx.needs_input_grad[1]:
            sink = _sink(ctx.W_param)
            if ctx.wacc is not None:            # a stacked (non-leaf) weight: StackFn owns the accumulation (one GEMM over all steps)
                ctx.wacc.add(x, dy)
            elif sink is not None and _defer(ctx.W_param) is not None:
                _defer(ctx.W_param).add((id(ctx.W_param), 0), x, dy, sink)
            elif sink is not None:
                gemm_tn(x, dy, sink, accumulate=True)
            else:
                dW = torch.empty_like(W)
                gemm_tn(x, dy, dW)
        return dx, dW


class StackFn(torch.autograd.Function):
    """[W_0 | W_1 | ...] along the columns (the heads of a GAT evaluated as one GEMM), or flattened row vectors stacked
    along the rows when ``rows`` (the attention vectors a_p [2F,1] -> [P, 2F]).  Inside a training step the stack is built
    ONCE (training.forward_train hoists it out of the time loop) and carries a _WAcc: its consumers record their operand
    pairs instead of returning a gradient per time step, and backward() turns them into one product / one column sum."""

    @staticmethod
    def forward(ctx, rows, *ws):
        slot = POOL.site()
        if rows:
            n = ws[0].numel()
            out = POOL.at(slot, "stack", (len(ws), n), ws[0])
            for i, w in enumerate(ws):
                copy2d(w.view(1, n), out[i:i + 1])
        else:
            K, F = ws[0].shape
            out = POOL.at(slot, "stack", (K, F * len(ws)), ws[0])
            for i, w in enumerate(ws):
                copy2d(w, out[:, i * F:(i + 1) * F])
        ctx.rows = rows
        ctx.shapes = [tuple(w.shape) for w in ws]
        ctx.wacc = _WAcc()
        ctx.slot = slot
        ctx.oshape = tuple(out.shape)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, dout):
        wacc = ctx.wacc
        if wacc.products or wacc.colsums:
            BR.join()            # the consumers that recorded these operands ran on the branches' streams
            total = POOL.at(ctx.slot, "dstack", ctx.oshape, wacc.products[0][0] if wacc.products else wacc.colsums[0])
            if wacc.products:
                flush_products(wacc.products, total, None, False)
            else:
                flush_colsums(wacc.colsums, total.view(-1), False)
            wacc.products, wacc.colsums = [], []
            if dout is not None:
                copy2d(_rows(dout), total, accumulate=True)
            dout = total
        if dout is None:
            return (None,) * (1 + len(ctx.shapes))
        dout = _rows(dout)
        grads = []
        for i, shp in enumerate(ctx.shapes):
            g = torch.empty(shp, dtype=torch.float32, device=dout.device)
            if ctx.rows:
                copy2d(dout[i:i + 1], g.view(1, -1))
            else:
                F = shp[1]
                copy2d(dout[:, i * F:(i + 1) * F], g)
            grads.append(g)
        return (None, *grads)


def stack(rows, *ws):
    out = StackFn.apply(rows, *ws)
    if POOL.active and out.grad_fn is not None and hasattr(out.grad_fn, "wacc"):
        out._hip_wacc = out.grad_fn.wacc
    return out


class SoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.slot = POOL.site()
        y = POOL.at(ctx.slot, "y", tuple(x.shape), x)
        (xp, ldx), (yp, ldy) = _m(x), _m(y)
        _check(_lib().dagnet_op_softmax_fwd(x.shape[0], x.shape[1], xp, ldx, yp, ldy, _stream(x.device)))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _rows(dy)
        dx = POOL.at(ctx.slot, "dx", tuple(y.shape), y)
        (yp, ldy), (gp, ldg), (xp, ldx) = _m(y), _m(dy), _m(dx)
        _check(_lib().dagnet_op_softmax_bwd(y.shape[0], y.shape[1], yp, ldy, gp, ldg, xp, ldx, _stream(y.device)))
        return dx


class GruLayerFn(torch.autograd.Function):
    """h' of one GRU layer step computed from the layer's INPUTS by dagnet_op_gru_layer_fwd (one kernel for rnn_dim 64:
    csrc/gru_fused.hip).  gi and gh are the outputs of the two gate products' LinearFn nodes, created lazily (allocated, not
    computed): they keep the autograd graph - and with it the whole backward pass (GruCellFn's gate gradients, then the
    products' input and weight gradients) - exactly as in the unfused form."""

    @staticmethod
    def forward(ctx, gi, gh, h, Wih, bih, Whh, bhh, pack, packb, *xs):
        M, R = h.shape
        ctx.slot = POOL.site()
        ctx.packb = packb
        ctx.widths = [x.shape[1] for x in xs]
        ctx.gaccs = [_gacc_of(h)] + [_gacc_of(x) for x in xs]       # fan-out aliases among (h, *xs): shared gradient buffers
        hn = POOL.at(ctx.slot, "hn", (M, R), h)
        gates = POOL.at(ctx.slot, "gates", (M, 3 * R), h)
        (x0, ldx0) = _m(xs[0], "x0")
        (x1, ldx1) = _m(xs[1], "x1") if len(xs) > 1 else (None, 0)
        (hp, ldh), (wi, ldwi), (wh, ldwh) = _m(h, "h"), _m(Wih, "W_ih"), _m(Whh, "W_hh")
        if not (gi.is_contiguous() and gh.is_contiguous()):
            raise DagnetHipError("gate buffers must be contiguous")
        _check(_lib().dagnet_op_gru_layer_fwd(M, R, x0, ldx0, xs[0].shape[1], x1, ldx1, xs[1].shape[1] if len(xs) > 1 else 0, hp, ldh,
                                              wi, ldwi, _p(bih), wh, ldwh, _p(bhh), _p(gi), _p(gh), _p(hn), _p(gates), _p(pack), _stream(h.device)))
        ctx.save_for_backward(gates, gh, h)
        ctx.n_extra = 6 + len(xs)
        return hn

    @staticmethod
    def backward(ctx, dhn):
        """One kernel from dh' to the gate gradients AND the input gradients of both products (dagnet_op_gru_layer_bwd); the two
        LinearFn nodes then only record their weight-gradient operands (ctx.dx_done).  Without a fused kernel for the shape:
        the gate-gradient kernel here, the products in the LinearFn nodes."""
        nodes = getattr(ctx, "lin_nodes", None)
        if ctx.packb is not None and nodes is not None and dhn is not None:
            gates, gh, h = ctx.saved_tensors
            M, R = h.shape
            dhn = _rows(dhn)
            dgi, dgh = POOL.at(ctx.slot, "dgi", (M, 3 * R), h), POOL.at(ctx.slot, "dgh", (M, 3 * R), h)
            want = [ctx.needs_input_grad[2]] + [ctx.needs_input_grad[9 + i] for i in range(len(ctx.widths))]
            dst, ret = [], []
            for j, (w, gacc) in enumerate(zip(want, ctx.gaccs)):
                if not w:
                    dst.append((None, 0, 0))
                    ret.append(None)
                elif gacc is not None:
                    buf = gacc.buffer(h)                 # every piece is [M, R] here
                    dst.append((_p(buf), buf.stride(0), int(gacc.started)))
                    ret.append(None)
                else:
                    buf = POOL.at(ctx.slot, "dh" if j == 0 else "dx%d" % (j - 1), (M, R), h)
                    dst.append((_p(buf), buf.stride(0), 0))
                    ret.append(buf)
            while len(dst) < 3:
                dst.append((None, 0, 0))
            (pd, ldd), (ph, ldh) = _m(dhn, "dhn"), _m(h, "h")
            K1 = ctx.widths[1] if len(ctx.widths) > 1 else 0
            rc = _lib().dagnet_op_gru_layer_bwd(M, R, ctx.widths[0], K1, pd, ldd, _p(gates), _p(gh), ph, ldh, _p(ctx.packb), _p(dgi), _p(dgh),
                                                *dst[1], *dst[2], *dst[0], _stream(h.device))
            if rc != 1:
                _check(rc)
                for j, (w, gacc) in enumerate(zip(want, ctx.gaccs)):
                    if w and gacc is not None:
                        gacc.done()
                for nd in nodes:
                    nd.dx_done = True
                return (dgi, dgh, ret[0], None, None, None, None, None, None, *ret[1:])
        return (*GruCellFn.backward(ctx, dhn), *([None] * ctx.n_extra))


class GruCellFn(torch.autograd.Function):
    """h' from gi = x W_ih^T + b_ih and gh = h W_hh^T + b_hh (nn.GRU gate order r, z, n)."""

    @staticmethod
    def forward(ctx, gi, gh, h):
        M, R = h.shape
        ctx.slot = POOL.site()
        hn = POOL.at(ctx.slot, "hn", (M, R), h)
        gates = POOL.at(ctx.slot, "gates", (M, 3 * R), h)
        _check(_lib().dagnet_op_gru_fwd(M, R, _p(gi), _p(gh), _p(h), _p(hn), _p(gates), _stream(h.device)))
        ctx.save_for_backward(gates, gh, h)
        return hn

    @staticmethod
    def backward(ctx, dhn):
        gates, gh, h = ctx.saved_tensors
        M, R = h.shape
        dhn = dhn.contiguous()
        dgi, dgh, dh = (POOL.at(ctx.slot, "dgi", (M, 3 * R), h), POOL.at(ctx.slot, "dgh", (M, 3 * R), h),
                        POOL.at(ctx.slot, "dh", (M, R), h))
        _check(_lib().dagnet_op_gru_bwd(M, R, _p(dhn), _p(gates), _p(gh), _p(h), _p(dgi), _p(dgh), _p(dh), _stream(h.device)))
        return dgi, dgh, dh


def _sum_rows(rows: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    out = torch.empty(1, dtype=torch.float32, device=rows.device)
    _check(_lib().dagnet_op_sum(rows.numel(), _p(rows), scale, _p(out), 0, _stream(rows.device)))
    return out


class LossSumFn(torch.autograd.Function):
    """scale * sum over all the per-row loss vectors of a step's time steps.  The vectors are StepPool slices, i.e. one
    contiguous [T*M] array: ONE reduction kernel per loss and step instead of one per time step plus a chain of scalar adds.
    Backward: every vector gets the incoming scalar gradient (times scale) as a stride-0 view - no kernel."""

    @staticmethod
    def forward(ctx, scale, *rows):
        ctx.scale, ctx.shapes = float(scale), [tuple(r.shape) for r in rows]
        n = rows[0].numel()
        ptrs = [r.data_ptr() for r in rows]
        lo = min(ptrs)
        if all(r.numel() == n and r.is_contiguous() for r in rows) and sorted(ptrs) == [lo + 4 * n * i for i in range(len(rows))]:
            # one contiguous array of total floats: column sums of its [total / w, w] view (parallel, deterministic), then
            # the w partial sums in one workgroup (a single workgroup over millions of floats takes milliseconds)
            total = n * len(rows)
            w = next(c for c in (64, 32, 16, 8, 4, 2, 1) if total % c == 0)
            lib = _lib()
            dev = rows[0].device
            part = torch.empty(w, dtype=torch.float32, device=dev)
            ws = _ws(dev, lib.dagnet_op_col_workspace(total // w, w))
            _check(lib.dagnet_op_colsum(total // w, w, C.c_void_p(lo), w, _p(part), 0, _p(ws), _stream(dev)))
            out = torch.empty(1, dtype=torch.float32, device=dev)
            _check(lib.dagnet_op_sum(w, _p(part), ctx.scale, _p(out), 0, _stream(dev)))
            return out
        out = torch.empty(1, dtype=torch.float32, device=rows[0].device)
        for i, r in enumerate(rows):
            _check(_lib().dagnet_op_sum(r.numel(), _p(r.contiguous()), ctx.scale, _p(out), int(i > 0), _stream(r.device)))
        return out

    @staticmethod
    def backward(ctx, g):
        gs = g.reshape(1) if ctx.scale == 1.0 else g.reshape(1) * ctx.scale
        return (None, *[gs.expand(shp) for shp in ctx.shapes])


class VaeFn(torch.autograd.Function):
    """z = eps * exp(0.5 logvar_enc) + mean_enc and the per-row KLD(enc || prior) (models/dagnet/model.py:118-128; the mean
    over the rows and the sum over the steps are LossSumFn's)."""

    @staticmethod
    def forward(ctx, mu_e, lv_e, mu_p, lv_p, eps):
        M, Z = mu_e.shape
        ctx.slot = POOL.site()
        z = POOL.at(ctx.slot, "z", (M, Z), mu_e)
        rows = POOL.at(ctx.slot, "rows", (M,), mu_e)
        _check(_lib().dagnet_op_vae_fwd(M, Z, _p(mu_e), _p(lv_e), _p(mu_p), _p(lv_p), _p(eps), _p(z), _p(rows), _stream(mu_e.device)))
        ctx.save_for_backward(mu_e, lv_e, mu_p, lv_p, eps)
        return z, rows

    @staticmethod
    def backward(ctx, dz, dk):
        mu_e, lv_e, mu_p, lv_p, eps = ctx.saved_tensors
        M, Z = mu_e.shape
        outs = [POOL.at(ctx.slot, "d%d" % i, (M, Z), mu_e) for i in range(4)]
        dz = None if dz is None else dz.contiguous()
        # dk: the gradient of every row's KLD - one scalar for all rows (LossSumFn hands over a stride-0 view of it)
        _check(_lib().dagnet_op_vae_bwd(M, Z, _p(mu_e), _p(lv_e), _p(mu_p), _p(lv_p), _p(eps), _p(dz), _scalar_grad(dk), 1.0, *[_p(o) for o in outs],
                                        _stream(mu_e.device)))
        return (*outs, None)


def _scalar_grad(g):
    """Pointer to the one value of a per-row loss gradient that is the same for every row (None -> NULL)."""
    if g is None:
        return None
    if g.numel() > 1 and any(st != 0 for st in g.stride()):
        raise DagnetHipError("per-row loss vectors must be reduced with LossSumFn (a uniform gradient per row)")
    return C.c_void_p(g.data_ptr())


class NllFn(torch.autograd.Function):
    """Per-row Gaussian negative log-likelihood (models/dagnet/model.py:130-135)."""

    @staticmethod
    def forward(ctx, mu, lv, x):
        M, X = mu.shape
        ctx.slot = POOL.site()
        rows = POOL.at(ctx.slot, "rows", (M,), mu)
        _check(_lib().dagnet_op_nll_fwd(M, X, _p(mu), _p(lv), _p(x), _p(rows), _stream(mu.device)))
        ctx.save_for_backward(mu, lv, x)
        return rows

    @staticmethod
    def backward(ctx, g):
        mu, lv, x = ctx.saved_tensors
        dmu, dlv = POOL.at(ctx.slot, "dmu", tuple(mu.shape), mu), POOL.at(ctx.slot, "dlv", tuple(lv.shape), lv)
        _check(_lib().dagnet_op_nll_bwd(mu.shape[0], mu.shape[1], _p(mu), _p(lv), _p(x), _scalar_grad(g), 1.0, _p(dmu), _p(dlv),
                                        _stream(mu.device)))
        return dmu, dlv, None


class CeFn(torch.autograd.Function):
    """Per row -p[row, goal[row]]  (the reference's "cross entropy" on probabilities, models/dagnet/model.py:162)."""

    @staticmethod
    def forward(ctx, p, goal):
        ctx.slot = POOL.site()
        rows = POOL.at(ctx.slot, "rows", (p.shape[0],), p)
        (pp, ldp) = _m(p)
        _check(_lib().dagnet_op_ce_fwd(p.shape[0], pp, ldp, _p(goal), _p(rows), _stream(p.device)))
        ctx.save_for_backward(goal)
        ctx.shape = tuple(p.shape)
        return rows

    @staticmethod
    def backward(ctx, g):
        (goal,) = ctx.saved_tensors
        M, G = ctx.shape
        dp = POOL.at(ctx.slot, "dp", (M, G), goal, dtype=torch.float32)
        _check(_lib().dagnet_op_ce_bwd(M, G, _p(goal), _scalar_grad(g), _p(dp), dp.stride(0), _stream(goal.device)))
        return dp, None


class SoftmaxCeFn(torch.autograd.Function):
    """Per row -softmax(x)[row, goal[row]]: nn.Softmax followed by the reference's probability "cross entropy"
    (models/dagnet/model.py:100,162) with a one-kernel backward (SoftmaxFn + CeFn: a [M, G] one-hot gradient and a softmax backward
    over it)."""

    @staticmethod
    def forward(ctx, x, goal):
        ctx.slot = POOL.site()
        M, G = x.shape
        y = POOL.at(ctx.slot, "y", (M, G), x)
        rows = POOL.at(ctx.slot, "rows", (M,), x)
        (xp, ldx), (yp, ldy) = _m(x), _m(y)
        _check(_lib().dagnet_op_softmax_fwd(M, G, xp, ldx, yp, ldy, _stream(x.device)))
        _check(_lib().dagnet_op_ce_fwd(M, yp, ldy, _p(goal), _p(rows), _stream(x.device)))
        ctx.save_for_backward(y, goal)
        return rows

    @staticmethod
    def backward(ctx, g):
        y, goal = ctx.saved_tensors
        M, G = y.shape
        dx = POOL.at(ctx.slot, "dx", (M, G), y)
        (yp, ldy), (xp, ldx) = _m(y), _m(dx)
        _check(_lib().dagnet_op_softmax_ce_bwd(M, G, yp, ldy, _p(goal), _scalar_grad(g), xp, ldx, _stream(y.device)))
        return dx, None


class _GradAcc:
    """Shared gradient buffer of a tensor with several consumers (FanoutFn)."""

    def __init__(self, slot, device):
        self.slot, self.started, self.buf = slot, False, None
        # The buffer is a side channel that autograd does not see, so only consumers on the stream of the alias itself may use
        # it (kernels of one stream run in order); a consumer on another branch's stream (Branches) returns its input gradient
        # the ordinary way and autograd orders and adds it.  (An event chain between the writers' streams was tried first:
        # hipStreamEndCapture does not return on the resulting graph.)
        self.stream = torch.cuda.current_stream(device).cuda_stream

    def buffer(self, like):
        if self.buf is None:
            self.buf = POOL.at(self.slot, "gacc", tuple(like.shape), like)
        return self.buf

    def done(self):
        self.started = True


def _gacc_of(x):
    """The shared gradient buffer of a fan-out alias, for a consumer that runs on the alias's own stream; else None."""
    g = getattr(x, "_hip_gacc", None)
    if g is not None and g.stream != torch.cuda.current_stream(x.device).cuda_stream:
        return None
    return g


class FanoutFn(torch.autograd.Function):
    """Identity on a tensor that several layers read.  Autograd would add up the consumers' input gradients one torch add
    at a time (3 x the tensor's bytes per add); instead the dense layers that read the alias write their input gradient
    straight into ONE shared buffer (the GEMM epilogue accumulates) and hand autograd nothing; gradients of other consumers
    arrive the usual way and are added here with one kernel."""

    @staticmethod
    def forward(ctx, x):
        out = x.view_as(x)
        ctx.gacc = _GradAcc(POOL.site(), x.device)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, g):
        acc = ctx.gacc
        if not acc.started:
            return g
        buf = acc.buf                    # its writers ran on this stream
        if g is not None:
            copy2d(_rows(g), buf, accumulate=True)
        return buf


class UnstackFn(torch.autograd.Function):
    """[T * M, W] -> T views [M, W] of a matrix computed for all time steps at once (a block that does not depend on the recurrence:
    phi_x of the whole observed sequence).  Backward: the steps' gradients are StepPool slices of one stacked buffer (the fan-out
    buffers of the per-step aliases), so the gradient of the whole matrix is a view of that buffer - no copy; anything else is
    concatenated."""

    @staticmethod
    def forward(ctx, big, T):
        M = big.shape[0] // T
        ctx.shape = tuple(big.shape)
        ctx.set_materialize_grads(False)
        return tuple(big[t * M:(t + 1) * M] for t in range(T))

    @staticmethod
    def backward(ctx, *gs):
        if all(g is None for g in gs):
            return None, None
        R, W = ctx.shape
        M = R // len(gs)
        ref = next(g for g in gs if g is not None)
        gs = [g if g is not None else torch.zeros((M, W), dtype=ref.dtype, device=ref.device) for g in gs]
        first = gs[0]
        if all(g.is_contiguous() and g.shape == (M, W) and g.data_ptr() == first.data_ptr() + 4 * i * M * W for i, g in enumerate(gs)):
            whole = torch.empty(0, dtype=first.dtype, device=first.device)
            whole.set_(first.untyped_storage(), first.storage_offset(), (R, W), (W, 1))
            return whole, None
        return torch.cat([_rows(g) for g in gs], 0), None


def fanout(x):
    """-> an alias of x whose consumers share one gradient buffer (only worth it inside a training step)."""
    if not POOL.active or not x.requires_grad:
        return x
    out = FanoutFn.apply(x)
    out._hip_gacc = out.grad_fn.gacc if hasattr(out.grad_fn, "gacc") else None
    return out


# ---------------------------------------------------------------------------------------------
# graph layers on the compact block-diagonal adjacency of a plan
# ---------------------------------------------------------------------------------------------
class GatAttnFn(torch.autograd.Function):
    """Attention of one GAT layer for H heads of width F stacked along the columns of Wh (gat_layers.py:25-40)."""

    @staticmethod
    def forward(ctx, Wh, a, adj, plan, H, F, alpha, elu):
        ctx.slot = POOL.site()
        out = POOL.at(ctx.slot, "out", tuple(Wh.shape), Wh)
        ctx.wacc = getattr(a, "_hip_wacc", None)
        (w, ldw), (o, ldo) = _m(Wh), _m(out)
        _check(_lib().dagnet_op_gat_attn_fwd(plan._h, _p(adj), _adj_ld(adj), w, ldw, _p(a), H, F, alpha, int(elu), o, ldo, _stream(Wh.device)))
        ctx.save_for_backward(Wh, a, adj, out)
        ctx.meta = (plan, H, F, alpha, elu)
        return out

    @staticmethod
    def backward(ctx, dout):
        Wh, a, adj, out = ctx.saved_tensors
        plan, H, F, alpha, elu = ctx.meta
        dout = _rows(dout)
        dWh = POOL.at(ctx.slot, "dWh", tuple(Wh.shape), Wh)
        part = POOL.at(ctx.slot, "da_part", (plan.S, H * 2 * F), Wh)      # per-scene contributions to da
        (w, ldw), (o, ldo), (g, ldg), (dw, ldd) = _m(Wh), _m(out), _m(dout), _m(dWh)
        _check(_lib().dagnet_op_gat_attn_bwd(plan._h, _p(adj), _adj_ld(adj), w, ldw, _p(a), H, F, alpha, int(elu), o, ldo, g, ldg, dw, ldd,
                                             _p(part), _stream(Wh.device)))
        if ctx.wacc is not None:                # stacked attention vectors: one column sum over all time steps (StackFn.backward)
            ctx.wacc.add_rows(part)
            return dWh, None, None, None, None, None, None, None
        da = torch.empty_like(a)
        colsum(part, da)
        return dWh, da, None, None, None, None, None, None


class BnFn(torch.autograd.Function):
    """nn.BatchNorm1d in train mode (batch statistics over all rows, running statistics updated in place) + optional tanh."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, eps, momentum, tanh_out):
        M, N = x.shape
        ctx.slot = POOL.site()
        y = POOL.at(ctx.slot, "y", (M, N), x)
        stat = POOL.at(ctx.slot, "stat", (2 * N,), x)
        tmp = POOL.at(ctx.slot, "tmp", (2 * N,), x)
        ws = _ws(x.device, _lib().dagnet_op_col_workspace(M, N))
        (xp, ldx), (yp, ldy) = _m(x), _m(y)
        _check(_lib().dagnet_op_bn_fwd(M, N, xp, ldx, _p(gamma), _p(beta), eps, momentum, int(tanh_out), yp, ldy, _p(stat),
                                       _p(run_mean), _p(run_var), _p(tmp), _p(ws), _stream(x.device)))
        ctx.save_for_backward(x, gamma, stat, y)
        ctx.tanh_out = tanh_out
        ctx.gamma_param, ctx.beta_param = gamma, beta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, stat, y = ctx.saved_tensors
        M, N = x.shape
        dy = _rows(dy)
        dx = POOL.at(ctx.slot, "dx", (M, N), x)
        # gradient sinks (training.attach_grad_arena): the reduction's second stage adds into p.grad, autograd gets None - one
        # AccumulateGrad add per parameter and call less (4 per time step of the DAG-Net step)
        gsink, bsink = _sink(ctx.gamma_param), _sink(ctx.beta_param)
        acc = gsink is not None and bsink is not None
        dgamma, dbeta = (gsink, bsink) if acc else (torch.empty_like(gamma), torch.empty_like(gamma))
        sums = POOL.at(ctx.slot, "sums", (2 * N,), x)
        ws = _ws(x.device, _lib().dagnet_op_col_workspace(M, N))
        (xp, ldx), (gp, ldg), (yp, ldy), (dp, ldd) = _m(x), _m(dy), _m(y), _m(dx)
        _check(_lib().dagnet_op_bn_bwd(M, N, xp, ldx, _p(stat), _p(gamma), gp, ldg, yp, ldy, int(ctx.tanh_out), dp, ldd, _p(dgamma),
                                       _p(dbeta), int(acc), _p(sums), _p(ws), _stream(x.device)))
        if acc:
            return dx, None, None, None, None, None, None, None
        return dx, dgamma, dbeta, None, None, None, None, None


class AdjMatmulFn(torch.autograd.Function):
    """adj @ x (+ bias) per scene (gcn_layers.py:20-26)."""

    @staticmethod
    def forward(ctx, x, bias, adj, plan):
        ctx.slot = POOL.site()
        y = POOL.at(ctx.slot, "y", tuple(x.shape), x)
        (xp, ldx), (yp, ldy) = _m(x), _m(y)
        _check(_lib().dagnet_op_adj_matmul(plan._h, x.shape[1], _p(adj), _adj_ld(adj), xp, ldx, _p(bias), 0, yp, ldy, _stream(x.device)))
        ctx.save_for_backward(adj)
        ctx.meta = (plan, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (adj,) = ctx.saved_tensors
        plan, has_b = ctx.meta
        dy = _rows(dy)
        dx = POOL.at(ctx.slot, "dx", tuple(dy.shape), dy)
        (gp, ldg), (dp, ldd) = _m(dy), _m(dx)
        _check(_lib().dagnet_op_adj_matmul(plan._h, dy.shape[1], _p(adj), _adj_ld(adj), gp, ldg, None, 1, dp, ldd, _stream(dy.device)))
        db = None
        if has_b and ctx.needs_input_grad[1]:
            db = torch.empty(dy.shape[1], dtype=torch.float32, device=dy.device)
            colsum(dy, db)
        return dx, db, None, None


def disposition(plan, pos: torch.Tensor, D: int) -> torch.Tensor:
    """d_t [N, D]: every row carries the flattened xy of its scene's agents (models/dagnet/model.py:141-147); no gradient."""
    out = POOL.at(POOL.site(), "d", (pos.shape[0], D), pos)
    _check(_lib().dagnet_op_disposition(plan._h, D, _p(pos.contiguous()), _p(out), D, _stream(pos.device)))
    return out


class BnSync:
    """Who shares BatchNorm statistics in train mode.  The reference's nn.BatchNorm1d normalises with the mean / variance of ALL
    rows of the batch (gat_model.py:21, gcn_model.py:16,18); with the scenes sharded over ranks that takes an exchange:
      * ``mode = "dist"``: the ranks of the default torch.distributed group - one all-gather of [3N] moments in the forward pass and
        one all-reduce of [2N] sums in the backward pass per BatchNorm call (RCCL; collectives cannot be captured into the training
        step's hipGraph by this code: run the step eagerly);
      * ``mode = "virtual"``: ``bounds`` cuts the rows of ONE process's batch into shards that go through the same staged kernels
        (moments per shard -> merge -> apply; sums per shard -> add -> apply) - how tests check the staged path against the
        whole-batch goldens on one GPU;
      * ``mode = None`` (default): per-process statistics (BnFn), the documented deviation under data parallelism."""
    mode = None
    bounds = None
    rows_all = None      # "dist": rows of the whole batch over all ranks when the caller knows it (else one blocking all-reduce per call)


BN_SYNC = BnSync()


def set_sync_batchnorm(mode, bounds=None):
    """mode: None | "dist" | "virtual" (bounds: row boundaries [0, m1, ..., M] of the virtual shards).  -> the previous (mode, bounds)."""
    prev = (BN_SYNC.mode, BN_SYNC.bounds)
    if mode not in (None, "dist", "virtual"):
        raise ValueError("sync BatchNorm mode must be None, 'dist' or 'virtual'")
    BN_SYNC.mode, BN_SYNC.bounds = mode, (list(bounds) if bounds is not None else None)
    return prev


class SyncBnFn(torch.autograd.Function):
    """nn.BatchNorm1d in train mode with statistics over the rows of all shards (BnSync), + optional tanh."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, eps, momentum, tanh_out):
        lib, st = _lib(), _stream(x.device)
        M, N = x.shape
        ctx.slot = POOL.site()
        y = POOL.at(ctx.slot, "y", (M, N), x)
        stat = POOL.at(ctx.slot, "stat", (2 * N,), x)
        tmp = POOL.at(ctx.slot, "tmp", (2 * N,), x)
        total = POOL.at(ctx.slot, "total", (1,), x)
        (xp, ldx), (yp, ldy) = _m(x), _m(y)
        if BN_SYNC.mode == "virtual":
            b = BN_SYNC.bounds
            assert b[0] == 0 and b[-1] == M, "virtual shard bounds must cover the batch"
            parts = POOL.at(ctx.slot, "parts", ((len(b) - 1) * 3 * N,), x).view(len(b) - 1, 3 * N)      # (1-D: the pool pads matrix rows)
            for k in range(len(b) - 1):
                rows = b[k + 1] - b[k]
                ws = _ws(x.device, lib.dagnet_op_col_workspace(max(rows, 1), N))
                _check(lib.dagnet_op_bn_moments(rows, N, C.c_void_p(x.data_ptr() + 4 * b[k] * ldx), ldx, _p(parts[k]), _p(tmp), _p(ws), st))
            ctx.shards = [(b[k], b[k + 1]) for k in range(len(b) - 1)]
        else:
            import torch.distributed as dist
            world = dist.get_world_size() if dist.is_initialized() else 1
            mom = POOL.at(ctx.slot, "mom", (3 * N,), x)
            ws = _ws(x.device, lib.dagnet_op_col_workspace(max(M, 1), N))
            _check(lib.dagnet_op_bn_moments(M, N, xp, ldx, _p(mom), _p(tmp), _p(ws), st))
            parts = POOL.at(ctx.slot, "parts", (world * 3 * N,), x).view(world, 3 * N)
            ctx.Mtot = M
            if world > 1:
                dist.all_gather_into_tensor(parts.view(-1), mom)
                if BN_SYNC.rows_all is not None:
                    ctx.Mtot = int(BN_SYNC.rows_all)
                else:
                    cnt = torch.tensor([float(M)], dtype=torch.float64, device=x.device)
                    dist.all_reduce(cnt)
                    ctx.Mtot = int(cnt.item())
            else:
                parts[0].copy_(mom)
            ctx.shards = [(0, M)]
        _check(lib.dagnet_op_bn_merge(N, parts.shape[0], _p(parts), eps, momentum, _p(stat), _p(run_mean), _p(run_var), _p(total), st))
        _check(lib.dagnet_op_bn_apply(M, N, xp, ldx, _p(stat), _p(gamma), _p(beta), int(tanh_out), yp, ldy, st))
        # (parts is NOT saved: it is a view of the step pool's stacked buffer, and the collective / copy_ that fills the next time
        # step's view bumps the version counter they all share - autograd would then refuse this step's backward; nothing in the
        # backward pass reads it)
        ctx.save_for_backward(x, gamma, stat, y)
        ctx.tanh_out, ctx.virtual = tanh_out, BN_SYNC.mode == "virtual"
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, stat, y = ctx.saved_tensors
        lib, st = _lib(), _stream(x.device)
        M, N = x.shape
        dy = _rows(dy)
        dx = POOL.at(ctx.slot, "dx", (M, N), x)
        (xp, ldx), (gp, ldg), (yp, ldy), (dp, ldd) = _m(x), _m(dy), _m(y), _m(dx)
        sums = POOL.at(ctx.slot, "sums", (len(ctx.shards) * 2 * N,), x).view(len(ctx.shards), 2 * N)
        for k, (r0, r1) in enumerate(ctx.shards):
            ws = _ws(x.device, lib.dagnet_op_col_workspace(max(r1 - r0, 1), N))
            _check(lib.dagnet_op_bn_bwd_sums(r1 - r0, N, C.c_void_p(x.data_ptr() + 4 * r0 * ldx), ldx, _p(stat),
                                             C.c_void_p(dy.data_ptr() + 4 * r0 * ldg), ldg, C.c_void_p(y.data_ptr() + 4 * r0 * ldy), ldy,
                                             int(ctx.tanh_out), _p(sums[k]), _p(ws), st))
        local = sums.sum(0) if len(ctx.shards) > 1 else sums[0]          # this process's share of dbeta | dgamma
        dbeta, dgamma = local[:N].clone(), local[N:].clone()
        tot_sums, Mtot = local, M
        if not ctx.virtual:
            import torch.distributed as dist
            Mtot = ctx.Mtot
            if dist.is_initialized() and dist.get_world_size() > 1:
                tot_sums = local.clone()
                dist.all_reduce(tot_sums)
        tot_sums = tot_sums.contiguous()
        for (r0, r1) in ctx.shards:
            _check(lib.dagnet_op_bn_bwd_apply(r1 - r0, Mtot, N, C.c_void_p(x.data_ptr() + 4 * r0 * ldx), ldx, _p(stat), _p(gamma),
                                              C.c_void_p(dy.data_ptr() + 4 * r0 * ldg), ldg, C.c_void_p(y.data_ptr() + 4 * r0 * ldy), ldy,
                                              int(ctx.tanh_out), _p(tot_sums), C.c_void_p(dx.data_ptr() + 4 * r0 * ldd), ldd, st))
        return dx, dgamma, dbeta, None, None, None, None, None


def bn_train(bn: torch.nn.BatchNorm1d, x: torch.Tensor, tanh_out: bool) -> torch.Tensor:
    momentum = 0.1 if bn.momentum is None else bn.momentum
    fn = SyncBnFn if BN_SYNC.mode is not None else BnFn
    y = fn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, momentum, tanh_out)
    if bn.num_batches_tracked is not None:
        if POOL.active:             # inside a training step: one add per module at the end of the pass (StepPool.end), not one per call
            POOL.bn_calls[bn] = POOL.bn_calls.get(bn, 0) + 1
        else:
            bn.num_batches_tracked += 1
    return y


GAT_WIDE_OUT = 128      # out heads wider than this aggregate BEFORE their dense layer (gat_train)


def gat_stacks(gat):
    """The stacked weights of a GAT: all heads' W as one [in, P*F] matrix, their attention vectors as [P, 2F], the out head's
    as [1, 2*nout] - and, for a wide out head (g_dim in the hundreds), its attention vectors folded through its weights:
    [1, 2*P*F] = [W a_src | W a_dst].  Built once per training step, outside the time loop."""
    Wcat, acat = stack(False, *[att.W for att in gat.attentions]), stack(True, *[att.a for att in gat.attentions])
    a2 = stack(True, gat.out_att.a)
    fold = None
    if gat.nout > GAT_WIDE_OUT:
        # (x W) . a = x . (W a): [2, nout] x W[P*F, nout]^T -> [2, P*F]; differentiable in W and in a (LinearFn: y = x W^T)
        f2 = LinearFn.apply(gat.out_att.W, None, ACT_NONE, 0.0, (0,), None, 0, 0, a2.view(2, gat.nout))
        fold = stack(True, f2)
    return Wcat, acat, a2, fold


def gat_train(gat, x, adj, plan, stacks=None):
    """GAT.forward in train mode (gat_model.py:18-22): P heads -> concat -> out head -> BatchNorm (batch statistics) -> tanh.
    A wide out head (nout > GAT_WIDE_OUT: the g_dim = 960 goal graph of the SDD models, models/dagnet/train_sdd.py:334) runs in
    the order of the evaluation path's goal head (csrc/wide_head.hip): its logits from the folded vectors, (x2 W) . a = x2 . (W a),
    and its aggregation on the P*F-wide head outputs BEFORE the dense layer, alpha (x2 W) = (alpha x2) W - the attention kernel
    then works on P*F = 32 columns per row instead of nout (whose scene tile would not fit the LDS), and the nout-wide matrix
    is written once."""
    P, F = gat.nheads, gat.nhid
    Wcat, acat, a2, fold = stacks if stacks is not None else gat_stacks(gat)
    Wh = MatmulFn.apply(x, Wcat)
    x2 = GatAttnFn.apply(Wh, acat, adj, plan, P, F, float(gat.alpha), True)
    if fold is not None:
        x3 = GatAttnFn.apply(x2, fold, adj, plan, 1, P * F, float(gat.alpha), False)
        o = MatmulFn.apply(x3, gat.out_att.W)
    else:
        Wh2 = MatmulFn.apply(x2, gat.out_att.W)
        o = GatAttnFn.apply(Wh2, a2, adj, plan, 1, gat.nout, float(gat.alpha), False)
    return bn_train(gat.bn1, o, True)


def gcn_train(gcn, x, adj, plan):
    """GCN.forward in train mode (gcn_model.py:14-19)."""
    x = AdjMatmulFn.apply(MatmulFn.apply(x, gcn.gc1.weight), gcn.gc1.bias, adj, plan)
    x = bn_train(gcn.bn1, x, False)
    x = AdjMatmulFn.apply(MatmulFn.apply(x, gcn.gc2.weight), gcn.gc2.bias, adj, plan)
    return bn_train(gcn.bn2, x, True)


def gru_packs(rnn: torch.nn.GRU, widths):
    """Per layer: the weights in the fused layer kernel's LDS order (csrc/gru_fused.hip), or None where the shape has no fused
    kernel.  Valid until the weights change: training.forward_train builds them once per call, ahead of the time loop."""
    lib, R, out = _lib(), rnn.hidden_size, []
    for l in range(rnn.num_layers):
        Wih, Whh = getattr(rnn, f"weight_ih_l{l}"), getattr(rnn, f"weight_hh_l{l}")
        bih, bhh = getattr(rnn, f"bias_ih_l{l}", None), getattr(rnn, f"bias_hh_l{l}", None)
        w = tuple(widths) if l == 0 else (R,)
        n = lib.dagnet_op_gru_pack_floats(R, w[0], w[1] if len(w) > 1 else 0) if len(w) <= 2 and bih is not None and Wih.is_cuda else 0
        if n <= 0:
            out.append(None)
            continue
        pack = POOL.at(POOL.site(), "grupack", (n,), Wih)
        (wi, ldwi), (wh, ldwh) = _m(Wih, "W_ih"), _m(Whh, "W_hh")
        _check(lib.dagnet_op_gru_pack(R, w[0], w[1] if len(w) > 1 else 0, wi, ldwi, _p(bih), wh, ldwh, _p(bhh), _p(pack), _stream(Wih.device)))
        packb = None
        nb = lib.dagnet_op_gru_pack_bwd_floats(R, w[0], w[1] if len(w) > 1 else 0) if torch.is_grad_enabled() else 0
        if nb > 0:        # the transposed images of the layer's backward kernel (GruLayerFn.backward)
            packb = POOL.at(POOL.site(), "grupackb", (nb,), Wih)
            _check(lib.dagnet_op_gru_pack_bwd(R, w[0], w[1] if len(w) > 1 else 0, wi, ldwi, wh, ldwh, _p(packb), _stream(Wih.device)))
        out.append((pack, packb))
    return out


def gru_step(rnn: torch.nn.GRU, xs, cols, h_layers, packs=None):
    """One seq_len = 1 call of nn.GRU; xs/cols: the pieces of the layer-0 input.  -> list of new per-layer states."""
    out = []
    for l in range(rnn.num_layers):
        Wih, Whh = getattr(rnn, f"weight_ih_l{l}"), getattr(rnn, f"weight_hh_l{l}")
        bih, bhh = getattr(rnn, f"bias_ih_l{l}"), getattr(rnn, f"bias_hh_l{l}")
        pieces, pcols = (list(xs), tuple(cols)) if l == 0 else ([out[-1]], (0,))
        contiguous_cols = all(c == sum(x.shape[1] for x in pieces[:i]) for i, c in enumerate(pcols))
        if len(pieces) <= 2 and contiguous_cols:
            global _LAZY_LINEAR
            _LAZY_LINEAR = True       # the two products are only recorded; GruLayerFn computes their values (and the cell) itself
            try:
                gi = LinearFn.apply(Wih, bih, ACT_NONE, 0.0, pcols, None, 0, 0, *pieces)
                gh = LinearFn.apply(Whh, bhh, ACT_NONE, 0.0, (0,), None, 0, 0, h_layers[l])
            finally:
                _LAZY_LINEAR = False
            pk, pkb = (packs[l] if packs and packs[l] is not None else (None, None))
            hn = GruLayerFn.apply(gi, gh, h_layers[l], Wih, bih, Whh, bhh, pk, pkb, *pieces)
            if hn.grad_fn is not None and gi.grad_fn is not None and gh.grad_fn is not None:
                hn.grad_fn.lin_nodes = (gi.grad_fn, gh.grad_fn)
        else:
            gi = LinearFn.apply(Wih, bih, ACT_NONE, 0.0, pcols, None, 0, 0, *pieces)
            gh = LinearFn.apply(Whh, bhh, ACT_NONE, 0.0, (0,), None, 0, 0, h_layers[l])
            hn = GruCellFn.apply(gi, gh, h_layers[l])
        out.append(fanout(hn))      # read by the next layer, the next step, the graph, ...
    return out


# ---------------------------------------------------------------------------------------------
# optimizer step: clip_grad_norm_ + Adam (train_bsk.py:289-290), two launches per parameter, no host sync
# ---------------------------------------------------------------------------------------------
class ClipAdam:
    """torch.nn.utils.clip_grad_norm_(params, clip) followed by torch.optim.Adam(params, lr).step() (defaults:
    betas (0.9, 0.999), eps 1e-8, no weight decay), as HIP kernels.  total_norm of the last step is kept on the device."""

    def __init__(self, params, lr=1e-3, clip=10.0, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        self.lr, self.clip, self.betas, self.eps = lr, clip, betas, eps
        self.step_count = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        dev = self.params[0].device
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=dev)

    def zero_grad(self):
        for d in {id(d): d for d in (getattr(p, "_hip_defer", None) for p in self.params) if d is not None}.values():
            d.reset()                           # a backward pass that raised must not poison this step
        arena = getattr(self, "arena", None)
        if arena is not None and all(getattr(p, "_hip_sink", False) and p.grad is not None for p in self.params):
            arena.zero_()                       # the parameters' .grad are views of one flat buffer
            return
        for p in self.params:
            p.grad = None

    def step(self):
        lib = _lib()
        live = [(p, m, v) for p, m, v in zip(self.params, self.m, self.v) if p.grad is not None]
        if not live:
            return
        st = _stream(live[0][0].device)
        self.sumsq.zero_()
        arena = getattr(self, "arena", None)
        if arena is not None and len(live) == len(self.params):
            _check(lib.dagnet_op_sumsq(arena.numel(), _p(arena), _p(self.sumsq), st))      # all gradients are one flat buffer
        else:
            for p, _, _ in live:
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                _check(lib.dagnet_op_sumsq(p.grad.numel(), _p(p.grad), _p(self.sumsq), st))
        self.step_count += 1
        flat = getattr(self, "flat", None)
        if flat is not None and arena is not None and len(live) == len(self.params) and self._flat_ok(flat, arena):
            # parameters, gradients and moments are views of four flat buffers in one layout (training.attach_grad_arena): one launch
            _check(lib.dagnet_op_adam(arena.numel(), _p(flat["p"]), _p(arena), _p(flat["m"]), _p(flat["v"]), _p(self.sumsq),
                                      float(self.clip or 0.0), self.lr, self.betas[0], self.betas[1], self.eps, self.step_count, st))
            for p in self.params:
                torch.autograd.graph.increment_version(p)
            return
        for p, m, v in live:
            _check(lib.dagnet_op_adam(p.numel(), _p(p.data), _p(p.grad), _p(m), _p(v), _p(self.sumsq), float(self.clip or 0.0), self.lr,
                                      self.betas[0], self.betas[1], self.eps, self.step_count, st))
            # the kernel wrote through the raw pointer: tell autograd, and the modules' packed-weight cache
            # (_runtime.HipBacked._fingerprint keys on the version counters), that the parameter changed
            torch.autograd.graph.increment_version(p)

    def _flat_ok(self, flat, arena):
        """Every parameter, gradient and moment still IS its slice of the flat buffers (something may have re-allocated one)."""
        bp, bg, bm, bv = flat["p"].data_ptr(), arena.data_ptr(), flat["m"].data_ptr(), flat["v"].data_ptr()
        for p, m, v, o in zip(self.params, self.m, self.v, flat["offs"]):
            if p.data_ptr() != bp + 4 * o or p.grad.data_ptr() != bg + 4 * o or m.data_ptr() != bm + 4 * o or v.data_ptr() != bv + 4 * o:
                return False
        return True

    def total_norm(self) -> torch.Tensor:
        return self.sumsq.sqrt().float()
