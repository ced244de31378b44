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


def rollout_best_of_k(model, args, batch, num_samples, seed=None, batch_index=0):
    """One batch of ``evaluate``: -> dict(ade_sum, fde_sum (fp64 device scalars), samples_rel [K,T,N,2], n_agents).
    ``seed``: the teacher-forced pass and the rollout draw from Philox streams of their own (loop.sub_seed: a hash of seed, batch
    index, phase and rank), so the rollout noise is independent of the noise that produced h and differs from rank to rank."""
    obs_traj, pred_traj_gt, obs_traj_rel, pred_traj_rel_gt, obs_goals, pred_goals_gt = batch[:6]
    seq_start_end = batch[-1]                     # 7-tuple (basketball loader) or 9-tuple (SDD loader): last field
    sp = scene_ptr_from_sse(seq_start_end, obs_traj.shape[1])
    from .loop import sub_seed
    seed_f, seed_s = sub_seed(seed, batch_index, 0), sub_seed(seed, batch_index, 1)
    if isinstance(model, DAGNet):
        adj = _obs_adjacency(args, seq_start_end, obs_traj, pred_traj_gt)
        goals = obs_goals.to(torch.int32)
        _, _, _, h = model(obs_traj, obs_traj_rel, goals, seq_start_end, adj, seed=seed_f)
        samples = model.sample_k(num_samples, args.pred_len, h, obs_traj[-1], goals[-1], seq_start_end, seed=seed_s)
    elif isinstance(model, AVRNN):
        adj = _obs_adjacency(args, seq_start_end, obs_traj, pred_traj_gt)
        _, _, h = model(obs_traj_rel, adj, seq_start_end=seq_start_end, seed=seed_f)
        samples = model.sample_k(num_samples, args.pred_len, h, obs_traj[-1], seq_start_end, seed=seed_s)
    elif isinstance(model, VRNN):
        _, _, h = model(obs_traj_rel, seed=seed_f)
        samples = model.sample_k(num_samples, args.pred_len, h, seed=seed_s)
    else:
        raise TypeError(f"unsupported model {type(model).__name__}")
    plan = _capi.HipPlan(None, sp, 1, device=obs_traj.device)
    r = _capi.best_of_k(plan, samples, obs_traj[-1].float(), pred_traj_gt.float())
    return dict(ade_sum=r["sums"][0], fde_sum=r["sums"][1], samples_rel=samples, n_agents=int(obs_traj.shape[1]),
                ade_raw=r["ade_raw"], fde_raw=r["fde_raw"], scene_best=r["scene_best"])


def best_of_k_chunked(model, args, num_samples, h, x_abs_start, g_start, pred_traj_gt, seq_start_end, chunk_scenes, eps=None, q=None,
                      seed=None):
    """The K rollouts + min-over-K metrics of a batch too large for one launch (rows x K >= 2^31, or the [K,T,N,2] output
    too large to keep): scenes are walked in chunks of ``chunk_scenes``, each chunk is one ``sample_k`` launch and one
    metric launch, the partial sums are added on the device (no host synchronisation).  Scenes are independent and all K
    samples of a scene stay in one chunk, so the result is the un-chunked one (bit for bit with injected ``eps`` / ``q``;
    with ``seed`` every chunk draws from a stream of its own, seed + chunk index).
    h [L,N,R], x_abs_start [N,2], g_start [N] goal indices (DAG-Net), pred_traj_gt [T,N,2]
    -> fp64 device tensor [ade_sum, fde_sum, n_agents], ready for ``finalize_metrics``."""
    sp = scene_ptr_from_sse(seq_start_end, h.shape[1])
    S, dev = len(sp) - 1, h.device
    tot = torch.zeros(3, dtype=torch.float64, device=dev)
    sse = torch.as_tensor(seq_start_end).reshape(-1, 2)
    for c, s0 in enumerate(range(0, S, int(chunk_scenes))):
        s1 = min(S, s0 + int(chunk_scenes))
        a, b = int(sp[s0]), int(sp[s1])
        sse_c = sse[s0:s1] - a
        hc, gt_c = h[:, a:b].contiguous(), pred_traj_gt[:, a:b].contiguous()
        e = None if eps is None else eps[:, :, a:b].contiguous()
        kw = dict(eps=e, seed=None if seed is None else int(seed) + c)
        if isinstance(model, DAGNet):
            qq = None if q is None else q[:, :, a:b].contiguous()
            s = model.sample_k(num_samples, args.pred_len, hc, x_abs_start[a:b], g_start[a:b], sse_c, q=qq, **kw)
        elif isinstance(model, AVRNN):
            s = model.sample_k(num_samples, args.pred_len, hc, x_abs_start[a:b], sse_c, **kw)
        else:
            s = model.sample_k(num_samples, args.pred_len, hc, **kw)
        plan = _metric_plan(sp[s0:s1 + 1] - a, dev)
        r = _capi.best_of_k(plan, s, x_abs_start[a:b].float().contiguous(), gt_c.float())
        tot[:2] += r["sums"]
        tot[2] += b - a
    return tot


_METRIC_PLANS = {}


def _metric_plan(scene_ptr, device):
    """Tables-only plan per scene layout (chunks of equal layout share one)."""
    key = (str(device), scene_ptr.tobytes())
    p = _METRIC_PLANS.get(key)
    if p is None:
        if len(_METRIC_PLANS) >= 16:
            _METRIC_PLANS.clear()
        p = _METRIC_PLANS[key] = _capi.HipPlan(None, scene_ptr, 1, device=device)
    return p


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
    for i, data in enumerate(loader):
        data = [t.to(dev) for t in data]
        if data[0].shape[1] == 0:                 # a data-parallel loader's empty share of a short batch
            continue
        r = rollout_best_of_k(model, args, data, num_samples, seed=seed, batch_index=i)
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
