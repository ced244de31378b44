"""Synthetic inputs of the reference's batch layout and deterministic
random weights.  Pure numpy/torch-CPU data generation; nothing here is on the compute path.

Batch tuple layouts mirror ``seq_collate`` (models/dagnet/data/loader_bsk.py:12-35,
loader_sdd.py:16-42): time-major [T,N,2] trajectories, relative coords with rel[0] = 0,
goal ids, seq_start_end [S,2].
"""
from __future__ import annotations

import zlib
from types import SimpleNamespace
from typing import Dict, Sequence

import numpy as np
import torch


def make_args(**kw) -> SimpleNamespace:
    """An ``args`` namespace with the attribute names the reference constructors read."""
    d = dict(x_dim=2, h_dim=64, z_dim=32, rnn_dim=64, n_layers=2, g_dim=90, graph_model="gat", graph_hid=8,
             adjacency_type=1, top_k_neigh=None, sigma=1.2, alpha=0.2, n_heads=4, obs_len=10, pred_len=40)
    d.update(kw)
    return SimpleNamespace(**d)


def _rng_for(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def random_state_dict(module_or_shapes, seed: int) -> Dict[str, torch.Tensor]:
    """Deterministic, order-independent weights for every entry of a state dict.

    Each tensor is drawn from its own generator keyed by (seed, crc32(name)), so the reference's
    module and ours get bit-identical values whatever their parameter registration order."""
    if isinstance(module_or_shapes, torch.nn.Module):
        shapes = {k: tuple(v.shape) for k, v in module_or_shapes.state_dict().items()}
    else:
        shapes = dict(module_or_shapes)
    out = {}
    for name, shape in shapes.items():
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros((), dtype=torch.int64)
            continue
        rng = _rng_for(seed, name)
        leaf = name.rsplit(".", 1)[-1]
        is_bn = ".bn" in name or name.startswith("bn")
        if is_bn and leaf in ("weight", "running_var"):
            a = rng.uniform(0.5, 1.5, size=shape)
        elif is_bn:
            a = rng.uniform(-0.2, 0.2, size=shape)
        elif leaf in ("W", "a") or ".gc" in name and leaf == "weight":
            fan = shape[0] + shape[-1]
            a = rng.uniform(-1.0, 1.0, size=shape) * 1.414 * np.sqrt(6.0 / fan)
        elif len(shape) >= 2:
            a = rng.uniform(-1.0, 1.0, size=shape) / np.sqrt(shape[-1])
        else:
            a = rng.uniform(-1.0, 1.0, size=shape) * 0.1
        out[name] = torch.from_numpy(np.asarray(a, dtype=np.float32))
    return out


def uniform_noise(seed: int, key: str, shape) -> torch.Tensor:
    """U[0,1) fp32 (the reference's rand_like noise)."""
    return torch.from_numpy(_rng_for(seed, key).random(size=shape, dtype=np.float32))


def exp1_noise(seed: int, key: str, shape) -> torch.Tensor:
    """Exp(1) fp32 (what torch.multinomial divides by)."""
    u = _rng_for(seed, key).random(size=shape, dtype=np.float64)
    return torch.from_numpy((-np.log1p(-u) + 1e-30).astype(np.float32))


def scene_ptr(sizes: Sequence[int]) -> np.ndarray:
    return np.concatenate([[0], np.cumsum(np.asarray(sizes, dtype=np.int64))]).astype(np.int32)


def seq_start_end(sizes: Sequence[int]) -> torch.Tensor:
    p = scene_ptr(sizes).astype(np.int64)
    return torch.from_numpy(np.stack([p[:-1], p[1:]], 1))


def ragged_sizes(n_scenes: int, seed: int, n_min=2, n_max=32, mean=6.0) -> np.ndarray:
    """SDD-like agent counts: truncated geometric on [n_min, n_max]."""
    rng = np.random.default_rng([seed, 7])
    p = 1.0 / (mean - n_min + 1.0)
    return np.minimum(n_min + rng.geometric(p, size=n_scenes) - 1, n_max).astype(np.int64)


def basketball_goals(pos: np.ndarray) -> np.ndarray:
    """9 x 10 grid of 5 ft cells on the 45 x 50 ft half court (datasets/scripts/preprocess_bsk.py:16-41)."""
    gx = np.clip((pos[..., 0] // 5).astype(np.int64), 0, 8)
    gy = np.clip((pos[..., 1] // 5).astype(np.int64), 0, 9)
    return gx * 10 + gy


def grid_goals(pos: np.ndarray, lo, hi, nx, ny) -> np.ndarray:
    fx = np.clip(((pos[..., 0] - lo[0]) / (hi[0] - lo[0]) * nx).astype(np.int64), 0, nx - 1)
    fy = np.clip(((pos[..., 1] - lo[1]) / (hi[1] - lo[1]) * ny).astype(np.int64), 0, ny - 1)
    return fx * ny + fy


def make_batch(sizes: Sequence[int], obs_len: int, pred_len: int, seed: int, kind: str = "bsk", g_cells=(9, 10)):
    """-> (obs_traj, pred_traj_gt, obs_traj_rel, pred_traj_rel_gt, obs_goals, pred_goals_gt, seq_start_end)
    as CPU tensors, the tuple ``seq_collate`` emits for the basketball loader."""
    rng = np.random.default_rng([seed, 11])
    n = int(np.sum(sizes))
    T = obs_len + pred_len
    if kind == "bsk":
        start = rng.uniform([0.0, 0.0], [45.0, 50.0], size=(n, 2))
        steps = rng.normal(0.0, 0.5, size=(T, n, 2))
    else:
        start = rng.uniform([0.0, 0.0], [60.0, 60.0], size=(n, 2))
        steps = rng.normal(0.0, 0.4, size=(T, n, 2))
    steps[0] = 0.0
    traj = (start[None] + np.cumsum(steps, 0)).astype(np.float32)
    rel = np.zeros_like(traj)
    rel[1:] = traj[1:] - traj[:-1]
    if kind == "bsk":
        goals = basketball_goals(traj)
    else:
        goals = grid_goals(traj, (0.0, 0.0), (60.0, 60.0), g_cells[0], g_cells[1])
    t = torch.from_numpy
    return (t(traj[:obs_len]), t(traj[obs_len:]), t(rel[:obs_len]), t(rel[obs_len:]),
            t(goals[:obs_len].astype(np.float32)), t(goals[obs_len:].astype(np.float32)), seq_start_end(sizes))
