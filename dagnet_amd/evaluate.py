"""Best-of-K evaluation with the reference's semantics (models/dagnet/evaluate_bsk.py:71-129,
models/avrnn/evaluate_bsk.py:70-123, models/baseline/evaluate_bsk.py:54-99): forward over the observed
steps once, K rollouts from the same h, per scene the minimum over K of the scene-summed ADE and
(independently) FDE, summed over scenes; ADE = sum / (total_agents * pred_len), FDE = sum / total_agents.

All K rollouts are one kernel launch (rows = scenes x agents x K) and the metric reduction is a
second one; with ``torch.distributed`` initialised the three partial sums are all-reduced (RCCL)."""
from __future__ import annotations

import torch

from . import _capi
from ._runtime import scene_ptr_from_sse
from .adjacency_matrix import compute_adjs, compute_adjs_distsim, compute_adjs_knnsim
from .avrnn import AVRNN
from .baseline import VRNN
from .dagnet import DAGNet
from .utils import to_goals_one_hot


def evaluate_helper(error, seq_start_end):
    """error: list of K tensors [N] -> sum over scenes of min over K of the scene's summed error."""
    err = torch.stack(list(error), dim=0).float().contiguous()          # [K,N]
    sp = scene_ptr_from_sse(seq_start_end, err.shape[1])
    plan = _capi.HipPlan(None, sp, 1, device=err.device)
    return _capi.scene_min_sum(plan, err)


def rollout_best_of_k(model, args, batch, num_samples, seed=None):
    """One batch of ``evaluate``: -> dict(ade_sum, fde_sum (fp64 device scalars), samples_rel [K,T,N,2], n_agents)."""
    (obs_traj, pred_traj_gt, obs_traj_rel, pred_traj_rel_gt, obs_goals, pred_goals_gt, seq_start_end) = batch[:7]
    sp = scene_ptr_from_sse(seq_start_end, obs_traj.shape[1])
    if isinstance(model, DAGNet):
        adj = _obs_adjacency(args, seq_start_end, obs_traj, pred_traj_gt)
        goals = obs_goals.to(torch.int32)
        _, _, _, h = model(obs_traj, obs_traj_rel, goals, seq_start_end, adj, seed=seed)
        samples = model.sample_k(num_samples, args.pred_len, h, obs_traj[-1], goals[-1], seq_start_end, seed=seed)
    elif isinstance(model, AVRNN):
        adj = _obs_adjacency(args, seq_start_end, obs_traj, pred_traj_gt)
        _, _, h = model(obs_traj_rel, adj, seq_start_end=seq_start_end, seed=seed)
        samples = model.sample_k(num_samples, args.pred_len, h, obs_traj[-1], seq_start_end, seed=seed)
    elif isinstance(model, VRNN):
        _, _, h = model(obs_traj_rel, seed=seed)
        samples = model.sample_k(num_samples, args.pred_len, h, seed=seed)
    else:
        raise TypeError(f"unsupported model {type(model).__name__}")
    plan = _capi.HipPlan(None, sp, 1, device=obs_traj.device)
    r = _capi.best_of_k(plan, samples, obs_traj[-1].float(), pred_traj_gt.float())
    return dict(ade_sum=r["sums"][0], fde_sum=r["sums"][1], samples_rel=samples, n_agents=int(obs_traj.shape[1]),
                ade_raw=r["ade_raw"], fde_raw=r["fde_raw"], scene_best=r["scene_best"])


def _obs_adjacency(args, seq_start_end, obs_traj, pred_traj_gt):
    # compact per-scene blocks: the dense [T,N,N] tensor of the reference is 5 GB at N=5120
    if args.adjacency_type == 0:
        return compute_adjs(args, seq_start_end, compact=True, device=obs_traj.device)
    if args.adjacency_type == 1:
        return compute_adjs_distsim(args, seq_start_end, obs_traj, pred_traj_gt, compact=True)
    return compute_adjs_knnsim(args, seq_start_end, obs_traj, pred_traj_gt, compact=True)


def evaluate(args, model, loader, num_samples, seed=None):
    """-> (ade, fde) python floats, as the reference's evaluate()."""
    dev = next(model.parameters()).device
    tot = torch.zeros(3, dtype=torch.float64, device=dev)     # ade_sum, fde_sum, total_traj
    for data in loader:
        data = [t.to(dev) for t in data]
        r = rollout_best_of_k(model, args, data, num_samples, seed=seed)
        tot[0] += r["ade_sum"]
        tot[1] += r["fde_sum"]
        tot[2] += r["n_agents"]
    return finalize_metrics(tot, args.pred_len)


def finalize_metrics(tot, pred_len):
    """tot = fp64 [ade_sum, fde_sum, n_agents] of this rank's scenes -> (ADE, FDE) over all ranks.
    With torch.distributed initialised this is the path's single collective (RCCL all-reduce of 24 bytes)."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.all_reduce(tot, op=torch.distributed.ReduceOp.SUM)
    tot = tot.cpu()
    return float(tot[0] / (tot[2] * pred_len)), float(tot[1] / tot[2])
