"""GPU: one training step through the HIP operators (dagnet_amd.training / dagnet_amd._ops, C ABI dagnet_op_*) against
the reference's own step captured in tests/golden/train_*.npz: losses, every parameter's gradient, the clipped-gradient
total norm, the parameters after one Adam step and the BatchNorm running statistics."""
import numpy as np
import pytest
import torch

from tests import cases
from tests.test_oracle_train import check_step

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def hip_train_step(meta, model, arena=False):
    from dagnet_amd import _ops
    from dagnet_amd.adjacency_matrix import compute_adjs, compute_adjs_distsim, compute_adjs_knnsim
    from dagnet_amd.utils import to_goals_one_hot
    batch, eps = cases.build_train_inputs(meta)
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = [t.to(DEV) for t in batch]
    eps = eps.to(DEV)
    args = cases.syn.make_args(obs_len=meta["obs_len"], pred_len=meta["pred_len"], **meta["args"])
    kind = meta["kind"]
    model.train()
    opt = _ops.ClipAdam(list(model.parameters()), lr=meta["lr"], clip=meta["clip"])
    if arena:
        from dagnet_amd.training import attach_grad_arena
        attach_grad_arena(model, opt)
    opt.zero_grad()
    if kind != "vrnn":
        if args.adjacency_type == 0:
            adj = compute_adjs(args, sse, device=DEV)
        elif args.adjacency_type == 1:
            adj = compute_adjs_distsim(args, sse, obs, gt)
        else:
            adj = compute_adjs_knnsim(args, sse, obs, gt)
    if kind == "vrnn":
        kld, nll, _ = model(obs_rel, eps=eps)
        ce = torch.zeros_like(kld)
        loss = meta["kld_weight"] * kld + nll
    elif kind == "avrnn":
        kld, nll, _ = model(obs_rel, adj, seq_start_end=sse, eps=eps)
        ce = torch.zeros_like(kld)
        loss = meta["kld_weight"] * kld + nll
    else:
        traj, rel = torch.cat((obs, gt), 0), torch.cat((obs_rel, gt_rel), 0)
        g_ohe = to_goals_one_hot(torch.cat((obs_goals, gt_goals), 0), args.g_dim)
        kld, nll, ce, _ = model(traj, rel, g_ohe, sse, adj, eps=eps)
        loss = meta["kld_weight"] * kld + nll + meta["ce_weight"] * ce
    loss.backward()
    grads = {k: (p.grad.detach().cpu().clone() if p.grad is not None else torch.zeros_like(p).cpu()) for k, p in model.named_parameters()}
    opt.step()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    return dict(kld=kld.detach().cpu(), nll=nll.detach().cpu(), ce=ce.detach().cpu(), loss=loss.detach().cpu(), grads=grads,
                total_norm=opt.total_norm().cpu(), params={k: sd[k] for k, _ in model.named_parameters()},
                buffers={k: v for k, v in sd.items()})


@pytest.mark.parametrize("name", cases.list_train_cases())
def test_train_step_matches_reference(name):
    meta, gold = cases.load_golden(name)
    _, model, _ = cases.build_model(meta)
    model = model.to(DEV)
    out = hip_train_step(meta, model)
    check_step(out, gold, meta["lr"], meta["clip"])


@pytest.mark.parametrize("name", ["train_dagnet_tiny_gat", "train_dagnet_tiny_gcn", "train_vrnn_tiny"])
def test_train_step_with_gradient_arena(name):
    """Flat gradient buffer + in-place accumulation by the backward kernels: same step as the plain autograd path."""
    meta, gold = cases.load_golden(name)
    _, model, _ = cases.build_model(meta)
    out = hip_train_step(meta, model.to(DEV), arena=True)
    check_step(out, gold, meta["lr"], meta["clip"])


def test_train_mode_forward_without_grad_uses_batch_statistics():
    """model.train() under no_grad: same losses as the differentiable call (batch-statistics BatchNorm), no tape."""
    meta, gold = cases.load_golden("train_avrnn_tiny_gat")
    _, model, _ = cases.build_model(meta)
    model = model.to(DEV).train()
    from dagnet_amd.adjacency_matrix import compute_adjs_distsim
    batch, eps = cases.build_train_inputs(meta)
    obs, gt, obs_rel, gt_rel, obs_goals, gt_goals, sse = [t.to(DEV) for t in batch]
    args = cases.syn.make_args(obs_len=meta["obs_len"], pred_len=meta["pred_len"], **meta["args"])
    with torch.no_grad():
        kld, nll, h = model(obs_rel, compute_adjs_distsim(args, sse, obs, gt), seq_start_end=sse, eps=eps.to(DEV))
    assert not kld.requires_grad
    np.testing.assert_allclose(kld.cpu().numpy(), gold["kld"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(nll.cpu().numpy(), gold["nll"], rtol=1e-4, atol=1e-5)
