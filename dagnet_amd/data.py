"""Batch layout of the reference's loaders: models/dagnet/data/loader_bsk.py:12-35,38-84
(``seq_collate``, ``BasketDataset``), models/dagnet/data/loader_sdd.py:16-42 (``seq_collate`` of the SDD loader) and the
goal grid of models/utils/utils.py:210-233 (``get_goal``, ``compute_goals_fixed``).

The reference collates a Python list of per-scene tensors with ``torch.cat`` for every batch.  Scenes of a split are
stored back to back here (time-major ``[T, N_all, C]`` plus a CSR ``scene_ptr``), so a batch of consecutive scenes is a
slice and a shuffled batch one ``index_select`` of rows - on the device, with no per-scene Python work
(``SceneBatcher``).  The tuples handed to the models are the reference's, field by field.

Reading the datasets' files (``trajectories.npy`` / ``goals.npy``, the SDD tracks) is plain numpy and not part of the hot
path; ``BasketDataset`` accepts the arrays directly or the directory the reference reads them from, ``TrajectoryDataset``
(loader_sdd.py:65-227, the SDD window extraction) the reference's ``.npy`` file or the list of tracks.
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


def poly_fit(traj, traj_len, threshold):
    """loader_sdd.py:45-62: 1.0 when the residuals of a quadratic fit to the last ``traj_len`` steps of ``traj [2, T]``
    reach ``threshold`` (non-linear trajectory), else 0.0.  Same numpy call as the reference: same bits."""
    t = np.linspace(0, traj_len - 1, traj_len)
    res_x = np.polyfit(t, traj[0, -traj_len:], 2, full=True)[1]
    res_y = np.polyfit(t, traj[1, -traj_len:], 2, full=True)[1]
    return 1.0 if res_x + res_y >= threshold else 0.0


class TrajectoryDataset(torch.utils.data.Dataset):
    """loader_sdd.py:65-227: sliding windows of ``obs_len + pred_len`` frames over the Stanford Drone Dataset tracks.

    ``source``: the ``.npy`` file the reference reads (an object array of ``(scene_name, data [rows, 4])`` with rows
    ``frame, pedestrian, x, y``; ``limits.json`` two directories above it), or a pair ``(files, scene_limits)`` with
    ``files`` a list of such tuples and ``scene_limits`` the dictionary of that json.
    Behaviour kept from the reference, quirks included: positions rounded to 4 decimals; a pedestrian counts only if it spans the
    whole window; and because the reference tests ``num_peds_considered > min_ped`` INSIDE its pedestrian loop
    (loader_sdd.py:166-172), a window with m accepted pedestrians contributes the sequences of its first
    ``min_ped + 1, min_ped + 2, ..., m`` pedestrians - growing prefixes of one window - and ``non_linear_ped`` repeats
    the flags of every prefix; its length therefore differs from the number of rows, as in the reference.
    The per-window work is numpy on sorted frame / pedestrian indices (one lexsort per file) instead of a boolean scan of the
    file per frame and per pedestrian."""

    def __init__(self, source, obs_len, pred_len, set="train", skip=1, n_cells_x=16, n_cells_y=15, goals_window=4, disable=True,
                 threshold=0.002, min_ped=1):
        super().__init__()
        import json
        import math
        import pathlib
        self.obs_len, self.pred_len, self.skip = obs_len, pred_len, skip
        self.seq_len = S = obs_len + pred_len
        self.n_cells_x, self.n_cells_y, self.goals_window = n_cells_x, n_cells_y, goals_window
        if isinstance(source, (str, os.PathLike)):
            self.data_dir = source
            root = pathlib.Path(source).parent.parent.absolute()
            with open(os.path.join(root, "limits.json")) as fh:
                scene_limits = json.load(fh)
            all_files = np.load(source, allow_pickle=True)        # the reference's own container format (the caller's file)
        else:
            all_files, scene_limits = source
        num_peds_in_seq, loss_masks, non_linear, seqs, seqs_rel, goals = [], [], [], [], [], []
        for file in all_files:
            scene_name, data = file[0], np.asarray(file[1], dtype=np.float64)
            frames = np.unique(data[:, 0])
            fidx = np.searchsorted(frames, data[:, 0])                  # frame number -> position in the frame list
            # rows grouped by frame in their original order (the reference concatenates per-frame selections)
            order = np.argsort(fidx, kind="stable")
            data, fidx = data[order], fidx[order]
            fstart = np.searchsorted(fidx, np.arange(len(frames) + 1))
            in_file = []
            num_sequences = int(math.ceil((len(frames) - S + 1) / skip))
            for idx in range(0, num_sequences * skip + 1, skip):
                lo, hi = fstart[min(idx, len(frames))], fstart[min(idx + S, len(frames))]
                if hi <= lo:
                    continue            # (the reference's np.concatenate of an empty list raises here; cannot happen for skip = 1)
                w, wf = data[lo:hi], fidx[lo:hi]
                # rows of the window grouped by pedestrian (ids ascending = np.unique's order), frames ascending inside
                po = np.lexsort((np.arange(hi - lo), w[:, 1]))
                w, wf = w[po], wf[po]
                pstart = np.flatnonzero(np.r_[True, w[1:, 1] != w[:-1, 1], True])
                m_all = len(pstart) - 1
                curr_seq, curr_rel, curr_mask = np.zeros((m_all, 2, S)), np.zeros((m_all, 2, S)), np.zeros((m_all, S))
                cnt, nl = 0, []
                for a, b in zip(pstart[:-1], pstart[1:]):
                    pad_front, pad_end = int(wf[a]) - idx, int(wf[b - 1]) - idx + 1
                    if pad_end - pad_front != S:
                        continue
                    xy = np.transpose(np.around(w[a:b], decimals=4)[:, 2:])
                    if xy.shape[1] != S:
                        raise ValueError("could not broadcast input array: a track with missing frames spans the window "
                                         "(the reference fails on this input too)")
                    rel = np.zeros(xy.shape)
                    rel[:, 1:] = xy[:, 1:] - xy[:, :-1]
                    curr_seq[cnt, :, pad_front:pad_end] = xy
                    curr_rel[cnt, :, pad_front:pad_end] = rel
                    nl.append(poly_fit(xy, pred_len, threshold))
                    curr_mask[cnt, pad_front:pad_end] = 1
                    cnt += 1
                    if cnt > min_ped:                       # inside the loop, as in the reference: one entry per prefix
                        non_linear += nl
                        num_peds_in_seq.append(cnt)
                        loss_masks.append(curr_mask[:cnt])
                        in_file.append(curr_seq[:cnt])
                        seqs.append(curr_seq[:cnt])
                        seqs_rel.append(curr_rel[:cnt])
            if in_file:
                all_traj = np.concatenate(in_file, axis=0)                                  # [rows, 2, S]
                x_min, x_max, y_min, y_max = scene_limits[scene_name].values()
                goals.append(compute_goals_fixed(all_traj.swapaxes(1, 2), x_min, x_max, y_min, y_max, n_cells_x, n_cells_y,
                                                 window=goals_window))
        if not seqs:
            raise ValueError("need at least one array to concatenate")          # numpy's message in the reference
        self.num_seq = len(seqs)
        self.seq_list, self.seq_list_rel = np.concatenate(seqs, axis=0), np.concatenate(seqs_rel, axis=0)
        goals = np.concatenate(goals, axis=0)
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).type(torch.float)     # noqa: E731
        self.obs_traj, self.pred_traj = f(self.seq_list[:, :, :obs_len]), f(self.seq_list[:, :, obs_len:])
        self.obs_traj_rel, self.pred_traj_rel = f(self.seq_list_rel[:, :, :obs_len]), f(self.seq_list_rel[:, :, obs_len:])
        self.obs_goals, self.pred_goals = f(goals[:, :obs_len]), f(goals[:, obs_len:])
        self.loss_mask = f(np.concatenate(loss_masks, axis=0))
        self.non_linear_ped = f(np.asarray(non_linear))
        cum = [0] + np.cumsum(num_peds_in_seq).tolist()
        self.seq_start_end = list(zip(cum, cum[1:]))
        self.scene_ptr = np.asarray(cum, dtype=np.int64)
        self.max_agents = max(num_peds_in_seq)

    def __len__(self):
        return self.num_seq

    def __max_agents__(self):
        return self.max_agents

    def __getitem__(self, index):
        a, b = self.seq_start_end[index]
        return [self.obs_traj[a:b, :], self.pred_traj[a:b, :], self.obs_traj_rel[a:b, :], self.pred_traj_rel[a:b, :],
                self.obs_goals[a:b], self.pred_goals[a:b], self.non_linear_ped[a:b], self.loss_mask[a:b, :]]

    def fields(self):
        """Time-major tensors of the whole split for ``SceneBatcher``, in the order of the SDD tuple's first six fields plus the
        loss mask: obs / pred ``[T, N, 2]``, their relative versions, goals ``[T, N]``, loss mask ``[T_all, N]``."""
        tm = [t.permute(2, 0, 1).contiguous() for t in (self.obs_traj, self.pred_traj, self.obs_traj_rel, self.pred_traj_rel)]
        return tm + [self.obs_goals.t().contiguous(), self.pred_goals.t().contiguous()]


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
