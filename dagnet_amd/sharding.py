"""Scene sharding for multi-GPU evaluation: scenes are independent in eval mode, so each rank
takes a contiguous range of scenes balanced by agent count and the only data-path collective is
the all-reduce of (ade_sum, fde_sum, total_agents) in evaluate()."""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_scenes(sizes, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) scene range of ``rank``; ranges partition all scenes and balance agents."""
    sizes = np.asarray(sizes, dtype=np.int64)
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world size")
    cum = np.concatenate([[0], np.cumsum(sizes)])
    total = cum[-1]
    bounds = [int(np.searchsorted(cum, total * r / world_size, side="left")) for r in range(world_size)] + [len(sizes)]
    for r in range(1, world_size + 1):
        bounds[r] = max(bounds[r], bounds[r - 1])
    return bounds[rank], bounds[rank + 1]


def shard_batch(batch, sizes, world_size: int, rank: int):
    """Slice a seq_collate-style batch tuple down to the scenes of ``rank``."""
    import torch
    lo, hi = shard_scenes(sizes, world_size, rank)
    sizes = np.asarray(sizes, dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(sizes)])
    a, b = int(cum[lo]), int(cum[hi])
    out = []
    for t in batch[:-1]:
        out.append(t[:, a:b].contiguous() if t.dim() >= 2 else t[a:b].contiguous())
    local = np.concatenate([[0], np.cumsum(sizes[lo:hi])])
    out.append(torch.from_numpy(np.stack([local[:-1], local[1:]], 1)))
    return tuple(out), sizes[lo:hi]
