"""Batch layout of the reference's loaders: models/dagnet/data/loader_bsk.py:12-35,38-84
(``seq_collate``, ``BasketDataset``), models/dagnet/data/loader_sdd.py:16-42 (``seq_collate`` of the SDD loader) and the
goal grid of models/utils/utils.py:210-233 (``get_goal``, ``compute_goals_fixed``).

The reference collates a Python list of per-scene tensors with ``torch.cat`` for every batch.  Scenes of a split are
stored back to back here (time-major ``[T, N_all, C]`` plus a CSR ``scene_ptr``), so a batch of consecutive scenes is a
slice and a shuffled batch one ``index_select`` of rows - on the device, with no per-scene Python work
(``SceneBatcher``).  The tuples handed to the models are the reference's, field by field.

Reading the datasets' files (``trajectories.npy`` / ``goals.npy``, the SDD ``.npy`` windows) is plain numpy and not
part of the hot path; ``BasketDataset`` accepts the arrays directly or the directory the reference reads them from.
"""
from __future__ import annotations

import os

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------
# goal grid (models/utils/utils.py:210-233)
# ------------------------------------------------------------------------------------------------
def _cell_index(v, lo, hi, n_cells):
    """Index of the last grid line <= v among linspace(lo, hi, n_cells + 1) (np.max(np.where(steps <= v)) in the
    reference): n_cells for v >= hi, ValueError below lo - as the reference's np.max of an empty selection."""
    steps = np.linspace(lo, hi, num=n_cells + 1)
    idx = np.searchsorted(steps, np.asarray(v, dtype=np.float64), side="right") - 1
    if np.any(idx < 0):
        raise ValueError("zero-size array to reduction operation maximum which has no identity")   # numpy's message
    return idx


def get_goal(position, min_x, max_x, min_y, max_y, n_cells_x, n_cells_y):
    """Cell of a grid of n_cells_x x n_cells_y over the scene rectangle that holds ``position`` (x, y)."""
    gx = _cell_index(position[0], min_x, max_x, n_cells_x)
    gy = _cell_index(position[1], min_y, max_y, n_cells_y)
    return (gx * n_cells_y) + gy


def compute_goals_fixed(tj, min_x, max_x, min_y, max_y, n_cells_x, n_cells_y, window=1):
    """Goals of one trajectory ``tj [T, 2]``: the cell of the position at the end of each window of ``window`` steps
    (and of the last step), held constant over the window.  -> float64 [T] like the reference.
    Accepts ``[..., T, 2]`` as well (all trajectories of a scene at once)."""
    tj = np.asarray(tj)
    T = tj.shape[-2]
    t = np.arange(T)
    anchor = np.minimum(((t // window) + 1) * window - 1, T - 1)      # the step whose position labels step t
    cells = (_cell_index(tj[..., 0], min_x, max_x, n_cells_x) * n_cells_y
             + _cell_index(tj[..., 1], min_y, max_y, n_cells_y))
    return np.take(cells, anchor, axis=-1).astype(np.float64)


# ------------------------------------------------------------------------------------------------
# collate functions (tuple layouts of the two loaders)
# ------------------------------------------------------------------------------------------------
def seq_collate_bsk(data):
    """loader_bsk.py:12-35.  data: list of (obs [To,n,2], pred [Tp,n,2], obs_rel, pred_rel, obs_goals [To,n],
    pred_goals [Tp,n]) with the same n everywhere -> the 7-tuple ending in seq_start_end [S,2] (int64)."""
    obs, pred, obs_rel, pred_rel, obs_g, pred_g = zip(*data)
    n = obs[0].shape[1]
    cat = [torch.cat(x, dim=1) for x in (obs, pred, obs_rel, pred_rel, obs_g, pred_g)]
    starts = torch.arange(0, len(obs) * n, n, dtype=torch.int64)
    return tuple(cat + [torch.stack((starts, starts + n), dim=1)])


def seq_collate_sdd(data):
    """loader_sdd.py:16-42.  data: list of (obs [n,2,To], pred [n,2,Tp], obs_rel, pred_rel, obs_goals [n,To],
    pred_goals [n,Tp], non_linear_ped [n], loss_mask [n,T]) -> the 9-tuple, trajectories as [T,N,2], goals [T,N]."""
    obs, pred, obs_rel, pred_rel, obs_g, pred_g, nl, mask = zip(*data)
    lens = np.array([len(s) for s in obs], dtype=np.int64)
    ends = np.cumsum(lens)
    sse = torch.from_numpy(np.stack((ends - lens, ends), axis=1))
    tm = [torch.cat(x, dim=0).permute(2, 0, 1) for x in (obs, pred, obs_rel, pred_rel)]
    gl = [torch.cat(x, dim=0).permute(1, 0) for x in (obs_g, pred_g)]
    return tuple(tm + gl + [torch.cat(nl), torch.cat(mask, dim=0), sse])


# ------------------------------------------------------------------------------------------------
# datasets
# ------------------------------------------------------------------------------------------------
class BasketDataset(torch.utils.data.Dataset):
    """loader_bsk.py:38-84: ``trajectories [T, S*n, 2]`` (absolute xy) and ``goals [T, S*n]`` of one split, n agents
    per scene.  ``source`` is the directory holding ``trajectories.npy`` / ``goals.npy`` or the pair of arrays."""

    def __init__(self, source, n_agents, obs_len=10, pred_len=40):
        super().__init__()
        if isinstance(source, (str, os.PathLike)):
            traj_abs = np.load(os.path.join(source, "trajectories.npy"))
            goals = np.load(os.path.join(source, "goals.npy"))
        else:
            traj_abs, goals = (np.asarray(a) for a in source)
        self.obs_len, self.pred_len, self.seq_len, self.n_agents = obs_len, pred_len, obs_len + pred_len, n_agents
        assert traj_abs.shape[0] == self.seq_len and goals.shape[0] == self.seq_len
        self.num_samples = traj_abs.shape[1] // n_agents
        traj_rel = np.zeros(traj_abs.shape)
        traj_rel[1:] = traj_abs[1:] - traj_abs[:-1]                    # rel[0] = 0 (loader_bsk.py:63-64)
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).type(torch.float)     # noqa: E731
        self.obs_traj, self.pred_traj = f(traj_abs[:obs_len]), f(traj_abs[obs_len:])
        self.obs_traj_rel, self.pred_traj_rel = f(traj_rel[:obs_len]), f(traj_rel[obs_len:])
        self.obs_goals, self.pred_goals = f(goals[:obs_len]), f(goals[obs_len:])
        self.seq_start_end = [[i * n_agents, (i + 1) * n_agents] for i in range(self.num_samples)]
        self.scene_ptr = np.arange(0, (self.num_samples + 1) * n_agents, n_agents, dtype=np.int64)

    def __len__(self):
        return self.num_samples

    def __max_agents__(self):
        return self.n_agents

    def __getitem__(self, idx):
        a, b = self.seq_start_end[idx]
        return [self.obs_traj[:, a:b], self.pred_traj[:, a:b], self.obs_traj_rel[:, a:b], self.pred_traj_rel[:, a:b],
                self.obs_goals[:, a:b], self.pred_goals[:, a:b]]

    def fields(self):
        """The six time-major tensors of the whole split, in the order of the collated tuple."""
        return [self.obs_traj, self.pred_traj, self.obs_traj_rel, self.pred_traj_rel, self.obs_goals, self.pred_goals]


class SceneBatcher:
    """Iterates over batches of whole scenes of one split held as time-major tensors ``[T, N_all, ...]`` plus a CSR
    ``scene_ptr`` - the reference's DataLoader + ``seq_collate`` without per-scene Python work:

      * the tensors are moved to ``device`` once;
      * ``shuffle=False``: a batch of consecutive scenes is a slice of rows (no copy);
      * ``shuffle=True``: the scene order is permuted per epoch (torch generator, like the DataLoader's sampler) and a
        batch is one ``index_select`` of rows per field on the device.

    Every batch is ``fields... + (seq_start_end [S,2] int64,)`` with rows renumbered from 0, i.e. exactly what
    ``seq_collate`` returns for those scenes.  ``rank`` / ``world`` split the scenes of every batch for data-parallel runs:
    the same batch boundaries on every rank, scene i of the batch goes to the rank with ``lo <= i < hi``,
    ``lo = len * rank // world`` (even shares, sizes differ by at most one).  Every rank yields EVERY batch, so the ranks
    stay in lock-step for the collectives of the step; a rank whose share is empty (a short last batch) gets a batch with
    zero rows and an empty ``seq_start_end`` and must still take part in them (``loop.train`` does).  ``global_rows`` /
    ``global_scenes`` describe the whole batch that was just yielded (all ranks), e.g. for ``row_fraction``."""

    def __init__(self, fields, scene_ptr, batch_size, shuffle=False, device=None, generator=None, rank=0, world=1):
        self.ptr = np.asarray(scene_ptr, dtype=np.int64)
        self.fields = [f.to(device) if device is not None else f for f in fields]
        self.batch_size, self.shuffle, self.gen = int(batch_size), bool(shuffle), generator
        self.rank, self.world = int(rank), int(world)
        self.dataset = self                      # len(loader.dataset) as the reference's epoch functions use it
        if not (0 <= self.rank < self.world):
            raise ValueError(f"rank {rank} outside world {world}")
        self.global_rows = self.global_scenes = 0

    def __len__(self):                           # scenes (as len(dataset)); number of batches: n_batches()
        return len(self.ptr) - 1

    def n_batches(self):
        return (len(self) + self.batch_size - 1) // self.batch_size

    def __max_agents__(self):
        return int(np.max(np.diff(self.ptr))) if len(self) else 0

    def __iter__(self):
        S = len(self)
        order = torch.randperm(S, generator=self.gen).numpy() if self.shuffle else np.arange(S)
        dev = self.fields[0].device
        for b0 in range(0, S, self.batch_size):
            scenes = order[b0:b0 + self.batch_size]
            self.global_scenes = len(scenes)
            self.global_rows = int(np.sum(self.ptr[scenes + 1] - self.ptr[scenes]))
            if self.world > 1:
                scenes = scenes[len(scenes) * self.rank // self.world:len(scenes) * (self.rank + 1) // self.world]
                if len(scenes) == 0:             # stay in lock-step: an empty share is still a batch
                    yield tuple([f[:, 0:0] for f in self.fields] + [torch.zeros((0, 2), dtype=torch.int64)])
                    continue
            sizes = self.ptr[scenes + 1] - self.ptr[scenes]
            ends = np.cumsum(sizes)
            sse = torch.from_numpy(np.stack((ends - sizes, ends), axis=1))
            if not self.shuffle:
                a, b = int(self.ptr[scenes[0]]), int(self.ptr[scenes[-1] + 1])
                out = [f[:, a:b] for f in self.fields]
            else:
                rows = np.repeat(self.ptr[scenes] - (ends - sizes), sizes) + np.arange(int(ends[-1]))   # CSR gather, vectorised
                idx = torch.from_numpy(rows).to(dev)
                out = [f.index_select(1, idx) for f in self.fields]
            yield tuple(out + [sse])
