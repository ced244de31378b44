"""Host-side plumbing shared by the drop-in modules: weight-pack cache, scene plans, layout helpers.

Nothing here computes on the CPU; tensors are only inspected to derive layouts (scene pointers).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import numpy as np
import torch

from . import _capi

_ADJ = {0: 0, 1: 1, 2: 2}
_GRAPH = {None: 0, "gat": _capi.GRAPH_GAT, "gcn": _capi.GRAPH_GCN}


_sse_cache: dict = {}
_goal_cache: dict = {}


def scene_ptr_from_sse(seq_start_end, n_rows: Optional[int] = None) -> np.ndarray:
    """The reference's seq_start_end [[start,end],...] -> CSR pointer int32[S+1].

    Scenes must tile the rows contiguously (they do for every loader of the reference:
    models/dagnet/data/loader_bsk.py:12-35, loader_sdd.py:16-42)."""
    if isinstance(seq_start_end, torch.Tensor):
        if seq_start_end.is_cuda:
            # a device tensor costs a blocking copy per call, which would also serialise the launches of a loop of sample()
            # calls (_capi._overlap_stream): the last one is remembered by storage and version counter
            key = (seq_start_end.data_ptr(), seq_start_end._version, tuple(seq_start_end.shape), n_rows)
            if _sse_cache.get("key") == key:
                return _sse_cache["ptr"]
            if torch.cuda.is_current_stream_capturing():
                raise ValueError("under hipGraph capture pass seq_start_end as a host tensor or list "
                                 "(a device tensor would need a blocking copy, which a capturing stream refuses)")
            ptr = scene_ptr_from_sse(seq_start_end.detach().cpu().numpy(), n_rows)
            _sse_cache.update(key=key, ptr=ptr, ref=seq_start_end)      # the reference keeps the address from being reused
            return ptr
        sse = seq_start_end.detach().numpy()
    else:
        sse = np.asarray(seq_start_end)
    sse = sse.reshape(-1, 2).astype(np.int64)
    if sse.shape[0] == 0:
        raise ValueError("seq_start_end is empty")
    if sse[0, 0] != 0 or np.any(sse[1:, 0] != sse[:-1, 1]) or np.any(sse[:, 1] <= sse[:, 0]):
        raise ValueError("seq_start_end must list contiguous, non-empty scenes starting at row 0")
    if n_rows is not None and sse[-1, 1] != n_rows:
        raise ValueError(f"seq_start_end covers {int(sse[-1, 1])} rows but the batch has {n_rows}")
    return np.concatenate([sse[:, 0], sse[-1:, 1]]).astype(np.int32)


def infer_scene_ptr_from_adj(adj: torch.Tensor) -> np.ndarray:
    """Derive the block-diagonal scene layout from the support of a dense adjacency
    ([N,N] or [T,N,N]).  Used when the reference signature carries no seq_start_end
    (AVRNN.forward(x, adj), GAT.forward(x, adj))."""
    a = adj if adj.dim() == 2 else (adj != 0).any(0)
    nz = (a != 0)
    n = nz.shape[0]
    idx = torch.arange(n, device=nz.device)
    big = torch.full_like(idx, n)
    first = torch.where(nz, idx[None, :], big[None, :]).min(1).values
    last = torch.where(nz, idx[None, :], torch.full_like(idx, -1)[None, :]).max(1).values
    first = torch.minimum(first, idx).cpu().numpy()       # a row always belongs to its own block
    last = torch.maximum(last, idx).cpu().numpy()
    # a block boundary before row b is valid iff no row's [first,last] interval crosses it
    reach = np.maximum.accumulate(last)                    # max last over rows < b  -> reach[b-1]
    back = np.minimum.accumulate(first[::-1])[::-1]        # min first over rows >= b -> back[b]
    cuts = [b for b in range(1, n) if reach[b - 1] < b and back[b] >= b]
    return np.asarray([0] + cuts + [n], dtype=np.int32)


def draw_seed() -> int:
    """Seed for the in-kernel Philox stream, drawn from torch's global generator so that
    torch.manual_seed() (the reference's set_random_seed) makes runs repeatable."""
    return int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())


class HipBacked:
    """Mixin for nn.Modules whose compute runs in libdagnet_hip.so."""

    _PLAN_CACHE = 8

    def _hip_cfg(self) -> dict:   # pragma: no cover - overridden
        raise NotImplementedError

    def _hip_state(self) -> dict:
        return self.state_dict()

    def _fingerprint(self, device):
        fp = [str(device)]
        for t in list(self.parameters()) + list(self.buffers()):
            fp.append((t.data_ptr(), t._version))
        return tuple(fp)

    def hip_model(self, device) -> _capi.HipModel:
        device = torch.device(device)
        if device.type != "cuda":
            raise _capi.DagnetHipError(
                f"inputs live on {device}; the HIP path needs a ROCm device and has no CPU fallback")
        fp = self._fingerprint(device)
        cache = self.__dict__.setdefault("_hip_cache", {})
        if cache.get("fp") != fp:
            cache.clear()
            cache["model"] = _capi.HipModel(self._hip_cfg(), self._hip_state(), device)
            cache["fp"] = fp
            cache["plans"] = OrderedDict()
        return cache["model"]

    def hip_plan(self, device, scene_ptr: np.ndarray, n_samples: int = 1) -> _capi.HipPlan:
        model = self.hip_model(device)
        plans = self.__dict__["_hip_cache"]["plans"]
        key = (scene_ptr.tobytes(), int(n_samples))
        plan = plans.get(key)
        if plan is None:
            plan = _capi.HipPlan(model, scene_ptr, n_samples)
            plans[key] = plan
            while len(plans) > self._PLAN_CACHE:
                plans.popitem(last=False)
        else:
            plans.move_to_end(key)
        return plan

    def _require_eval(self, what: str):
        if self.training and getattr(self, "_has_batchnorm", False):
            raise NotImplementedError(
                f"{what}: train-mode BatchNorm (batch statistics over all rows) is not implemented in the HIP path yet; "
                "call model.eval() (see DESIGN.md, row T)")


def as_goal_index(g: torch.Tensor, g_dim: int) -> torch.Tensor:
    """one-hot [.., G] float (the reference's goals_ohe) or integer indices [..] -> int32 indices."""
    if g.dtype == torch.int32:
        return g
    # the conversion makes a new tensor; a loop that passes the same one-hot tensor again gets the same index tensor back, so that
    # consecutive sample() calls are seen to start from the same inputs (_capi._overlap_stream)
    key = (g.data_ptr(), g._version, tuple(g.shape), g.dtype, g_dim)
    if _goal_cache.get("key") == key:
        return _goal_cache["idx"]
    if g.dtype in (torch.int64, torch.int16, torch.uint8):
        idx = g.to(torch.int32)
    else:
        if g.shape[-1] != g_dim:
            raise ValueError(f"goal tensor must be one-hot with last dim {g_dim}, got {tuple(g.shape)}")
        idx = g.argmax(-1).to(torch.int32)
    if g.is_cuda and g.dim() <= 2 and not torch.cuda.is_current_stream_capturing():     # g_start of sample(), not forward()'s [T,N,G]
        _goal_cache.update(key=key, idx=idx, ref=g)
    return idx
