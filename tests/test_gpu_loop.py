"""GPU: the epoch functions of dagnet_amd.loop (validate / test / train / fit) on synthetic
splits.  ``validate`` must equal the composition of the individually parity-tested pieces (teacher-forced forward,
one sample(), relative_to_abs, 'sum'-mode ADE / FDE: models/dagnet/train_bsk.py:177-248); training must make the next
eval-mode call see the new weights; a checkpoint must reproduce the model."""
import numpy as np
import pytest
import torch

import dagnet_amd
from dagnet_amd import data as D, loop, synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TINY = dict(h_dim=16, z_dim=8, rnn_dim=16, g_dim=12)


def _split(kind, sizes, obs_len, pred_len, seed, batch_size, args_kw, g_cells=(3, 4)):
    args = syn.make_args(obs_len=obs_len, pred_len=pred_len, CE_weight=1e-2, learning_rate=1e-3, clip=10.0, num_epochs=2,
                         warmup=True, lr_scheduler=True, eval_every=2, num_samples=3, **args_kw)
    batch = syn.make_batch(sizes, obs_len, pred_len, seed, kind="sdd", g_cells=g_cells)
    ptr = syn.scene_ptr(sizes)
    loader = D.SceneBatcher(list(batch[:6]), ptr, batch_size=batch_size, device=DEV)
    if kind == "dagnet":
        model = dagnet_amd.DAGNet(args, int(max(sizes)))
    elif kind == "avrnn":
        model = dagnet_amd.AVRNN(args)
    else:
        model = dagnet_amd.VRNN(args)
    model.load_state_dict(syn.random_state_dict(model, seed))
    return args, model.to(DEV).eval(), loader, batch


@pytest.mark.parametrize("kind,kw", [("vrnn", dict(TINY, n_layers=2)),
                                     ("avrnn", dict(h_dim=64, z_dim=32, rnn_dim=64, n_layers=1)),
                                     ("dagnet", dict(h_dim=64, z_dim=32, rnn_dim=64, g_dim=90, n_layers=2))])
def test_validate_is_the_composition_of_the_pieces(kind, kw):
    sizes = [3, 5, 2, 4, 6]
    args, model, loader, _ = _split(kind, sizes, 5, 7, 21, 2, kw, g_cells=(9, 10))
    warm = loop.kld_warmup(4, True, warmup_epochs=4)
    epch, seed = 2, 100
    ade, fde = loop.validate(args, epch, loader, model, warm, seed=seed)
    r = loop.validate.last
    acc = np.zeros(7)
    with torch.no_grad():
        for i, b in enumerate(loader):
            obs, gt, obs_rel, gt_rel, og, gg, sse = b
            if kind == "dagnet":
                adj = dagnet_amd.adjacency_matrix.compute_adjs_distsim(args, sse, obs, gt)          # dense, as the reference
                goh = dagnet_amd.utils.to_goals_one_hot(og, args.g_dim)
                kld, nll, ce, h = model(obs, obs_rel, goh, sse, adj, seed=loop.sub_seed(seed, i, 0))
                s = model.sample(args.pred_len, h, obs[-1], goh[-1], sse, seed=loop.sub_seed(seed, i, 1))
            elif kind == "avrnn":
                adj = dagnet_amd.adjacency_matrix.compute_adjs_distsim(args, sse, obs, gt)
                kld, nll, h = model(obs_rel, adj, seed=loop.sub_seed(seed, i, 0))
                ce = torch.zeros(1)
                s = model.sample(args.pred_len, h, obs[-1], sse, seed=loop.sub_seed(seed, i, 1))
            else:
                kld, nll, h = model(obs_rel, seed=loop.sub_seed(seed, i, 0))
                ce = torch.zeros(1)
                s = model.sample(args.pred_len, h, seed=loop.sub_seed(seed, i, 1))
            pred = (obs[-1][None] + torch.cumsum(s, 0)).cpu().double()
            err = (pred - gt.cpu().double()).norm(dim=2)
            w = warm[epch - 1]
            acc += [float(nll) + w * float(kld) + float(ce) * (args.CE_weight if kind == "dagnet" else 0.0), float(kld), float(nll), float(ce),
                    float(err.sum()), float(err[-1].sum()), pred.shape[1]]
    S = len(sizes)
    want = dict(loss=acc[0] / S, kld=acc[1] / S, nll=acc[2] / S, ce=acc[3] / S, ade=acc[4] / (acc[6] * args.pred_len), fde=acc[5] / acc[6])
    for k, v in want.items():
        assert r[k] == pytest.approx(v, rel=2e-5, abs=1e-6), k
    assert r["total_traj"] == sum(sizes) and ade == r["ade"] and fde == r["fde"]
    t = loop.test(args, epch, loader, model, warm, seed=seed)
    assert t["ade"] == ade and t["loss"] == r["loss"]                                                  # same seeds, same path


def test_training_invalidates_the_packed_weights():
    """VRNN has no BatchNorm buffer whose version changes: only the optimizer's kernels touch the parameters."""
    args, model, loader, batch = _split("vrnn", [4, 4], 5, 4, 3, 2, dict(TINY, n_layers=1))
    x = batch[2].to(DEV)
    eps = syn.uniform_noise(1, "e", (4, 8, 8)).to(DEV)
    h0 = model(x, eps=eps)[2].clone()
    opt = dagnet_amd._ops.ClipAdam(list(model.parameters()), lr=1e-2, clip=10.0)
    dagnet_amd.training.train_step(model, opt, (x,))
    model.eval()
    h1 = model(x, eps=eps)[2]
    assert (h1 - h0).abs().max() > 1e-4


def test_fit_and_checkpoint_round_trip(tmp_path):
    sizes = [2, 5, 3, 4, 2, 6]
    args, model, loader, _ = _split("dagnet", sizes, 4, 5, 9, 3, dict(TINY, n_layers=2))
    warm = loop.kld_warmup(args.num_epochs, True)
    before = loop.test(args, 1, loader, model, warm, seed=7)
    best = []
    saved = []
    hist = loop.fit(args, model, loader, loader, loader, on_best=lambda e, a, ctx: best.append((e, a)),
                    on_epoch=lambda e, ctx: (ctx.save(tmp_path / f"periodic_{e}.pth"), saved.append(e)), seed=7)
    assert saved == [1, 2] and torch.load(tmp_path / "periodic_1.pth", weights_only=True)["epoch"] == 1
    assert len(hist["train"]) == len(hist["val"]) == len(hist["test"]) == 2 and len(hist["ade_list_test"]) == 1
    assert all(np.isfinite(list(h.values())).all() for h in hist["train"] + hist["val"])
    assert hist["train"][0]["kld"] > 0 and best and hist["best_ade"] == min(a for _, a in best)
    ade_bok = hist["ade_list_test"][0][0]
    assert 0 < ade_bok <= hist["test"][-1]["ade"] * 1.5                       # best of 3 is not (much) worse than one sample
    after = loop.test(args, 1, loader, model, warm, seed=7)
    assert abs(after["nll"] - before["nll"]) > 1e-6                           # eval after training runs on the trained weights
    fn = tmp_path / "checkpoint_epoch_2.pth"
    loop.save_checkpoint(fn, args, 2, model, hist["optimizer"], hist["lr_scheduler"], hist["best_ade"], hist["ade_list_val"],
                         hist["fde_list_val"], hist["ade_list_test"], hist["fde_list_test"])
    fresh = dagnet_amd.DAGNet(args, int(max(sizes))).to(DEV)
    opt = dagnet_amd._ops.ClipAdam(list(fresh.parameters()), lr=1.0, clip=10.0)
    ck = loop.load_checkpoint(fn, fresh, opt, loop.PlateauScheduler(opt))
    assert ck["epoch"] == 2 and ck["args"]["g_dim"] == 12 and opt.lr == hist["optimizer"].lr
    assert opt.step_count == hist["optimizer"].step_count == 4 and torch.equal(opt.m[0], hist["optimizer"].m[0])
    again = loop.test(args, 1, loader, fresh.eval(), warm, seed=7)
    assert again == after
    # resuming: the restored optimizer / scheduler go back into fit and carry on (step count, moments, learning rate)
    args.num_epochs = 3
    h2 = loop.fit(args, fresh, loader, loader, loader, start_epoch=2, best_ade=ck["best_ade"], seed=7, optimizer=opt,
                  lr_scheduler=loop.PlateauScheduler(opt))
    assert len(h2["train"]) == 1 and h2["optimizer"] is opt and opt.step_count == 6

