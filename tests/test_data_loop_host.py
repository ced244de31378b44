"""CPU: batch layout and goal grid (dagnet_amd.data) against outputs of the reference's own loader code
(tests/golden/data_utils.npz, written by oracle/make_golden.py data), and the host logic of dagnet_amd.loop
(KLD warm-up, plateau scheduler, checkpoint dictionary layout)."""
import os

import numpy as np
import pytest
import torch

from dagnet_amd import data as D

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_utils.npz"))


def test_goal_grid_matches_reference():
    tj, lim = GOLD["goal_tj"], GOLD["goal_limits"]
    for w in (1, 3, 4, 7):
        want = GOLD[f"goals_w{w}"]
        got = np.stack([D.compute_goals_fixed(t, *lim, 32, 30, window=w) for t in tj])
        assert got.dtype == want.dtype and np.array_equal(got, want)
        assert np.array_equal(D.compute_goals_fixed(tj, *lim, 32, 30, window=w), want)      # all trajectories at once
    got = np.asarray([D.get_goal(p, *lim, 32, 30) for p in tj[:, 0]])
    assert np.array_equal(got, GOLD["get_goal"])
    assert D.get_goal([lim[1], lim[3]], *lim, 32, 30) == 32 * 30 + 30                       # upper border: index n_cells (reference quirk)
    with pytest.raises(ValueError):                                                        # below the rectangle: np.max of nothing
        D.get_goal([lim[0] - 1.0, 5.0], *lim, 32, 30)


def _items(prefix, n_items, n_fields):
    return [[torch.from_numpy(GOLD[f"{prefix}_in_{i}_{j}"]) for j in range(n_fields)] for i in range(n_items)]


def test_collate_layouts_match_reference():
    for prefix, fn, nf in (("bsk", D.seq_collate_bsk, 6), ("sdd", D.seq_collate_sdd, 8)):
        got = fn(_items(prefix, 3, nf))
        assert len(got) == nf + 1
        for j, t in enumerate(got):
            want = GOLD[f"{prefix}_out_{j}"]
            assert tuple(t.shape) == want.shape and np.array_equal(t.numpy(), want), (prefix, j)
        assert got[-1].dtype == torch.int64


def test_basket_dataset_and_scene_batcher():
    traj, goals = GOLD["ds_traj"], GOLD["ds_goals"]
    ds = D.BasketDataset((traj, goals), n_agents=5, obs_len=4, pred_len=6)
    assert len(ds) == int(GOLD["ds_len"]) and ds.__max_agents__() == 5
    for j, t in enumerate(ds[2]):
        assert np.array_equal(t.numpy(), GOLD[f"ds_item2_{j}"]), j
    # consecutive scenes: slices equal the collated list; shuffled: the same scenes, rows renumbered
    want = D.seq_collate_bsk([ds[i] for i in range(len(ds))])
    (got,) = list(D.SceneBatcher(ds.fields(), ds.scene_ptr, batch_size=8))
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    g = torch.Generator().manual_seed(3)
    order = torch.randperm(len(ds), generator=torch.Generator().manual_seed(3)).tolist()
    batches = list(D.SceneBatcher(ds.fields(), ds.scene_ptr, batch_size=3, shuffle=True, generator=g))
    assert [b[-1].shape[0] for b in batches] == [3, 1]
    want = D.seq_collate_bsk([ds[i] for i in order[:3]])
    for a, b in zip(batches[0], want):
        assert torch.equal(a, b)
    # data parallel: the ranks' shares of a batch partition its scenes
    parts = [list(D.SceneBatcher(ds.fields(), ds.scene_ptr, batch_size=4, rank=r, world=2))[0] for r in range(2)]
    assert torch.equal(torch.cat([p[0] for p in parts], 1), ds.obs_traj)


def test_ragged_scene_batcher():
    ptr = np.array([0, 2, 7, 10, 11])
    x = torch.arange(11, dtype=torch.float32)[None, :, None].expand(3, 11, 2).contiguous()
    g = torch.Generator().manual_seed(1)
    order = torch.randperm(4, generator=torch.Generator().manual_seed(1)).tolist()
    (b,) = list(D.SceneBatcher([x], ptr, batch_size=4, shuffle=True, generator=g))
    rows = np.concatenate([np.arange(ptr[s], ptr[s + 1]) for s in order])
    assert np.array_equal(b[0][0, :, 0].numpy(), rows)
    sizes = [ptr[s + 1] - ptr[s] for s in order]
    assert b[-1][:, 1].tolist() == np.cumsum(sizes).tolist() and b[-1][0, 0] == 0


def test_kld_warmup_schedule():
    from dagnet_amd.loop import kld_warmup
    w = kld_warmup(120, True)
    ref = np.ones(120)
    ref[:50] = np.linspace(0, 1, num=50)                 # models/dagnet/train_bsk.py:377-380
    assert np.array_equal(w, ref)
    assert np.array_equal(kld_warmup(7, False), np.ones(7))


class _Opt:
    def __init__(self, lr):
        self.lr = lr


def test_plateau_scheduler_follows_torch():
    from dagnet_amd.loop import PlateauScheduler
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.Adam([p], lr=1e-3)
    tsch = torch.optim.lr_scheduler.ReduceLROnPlateau(topt, threshold=1e-2, patience=10, factor=5e-1)
    mine = PlateauScheduler(_Opt(1e-3))
    rng = np.random.default_rng(0)
    metric = 10.0
    for e in range(200):
        metric = metric * (0.97 if e < 40 else 1.0) + 0.01 * rng.normal()
        tsch.step(metric)
        mine.step(metric)
        assert mine.optimizer.lr == pytest.approx(topt.param_groups[0]["lr"], rel=0, abs=0), e
        assert mine.num_bad_epochs == tsch.num_bad_epochs
    assert mine.optimizer.lr < 1e-3
    sd = mine.state_dict()
    assert set(sd) >= {"best", "num_bad_epochs", "last_epoch", "patience", "factor", "threshold"}
    again = PlateauScheduler(_Opt(1e-3))
    again.load_state_dict(sd)
    assert again.best == mine.best and again.optimizer.lr == mine.optimizer.lr


def test_checkpoint_dictionary_layout(tmp_path):
    """Keys and nesting of train_bsk.py:80-93; Adam's state_dict layout inside."""
    from dagnet_amd import loop

    class FakeAdam:                                    # the attributes of _ops.ClipAdam that the checkpoint code reads
        def __init__(self, params):
            self.params, self.lr, self.betas, self.eps, self.step_count = params, 1e-3, (0.9, 0.999), 1e-8, 3
            self.m = [torch.full_like(p, 0.5) for p in params]
            self.v = [torch.full_like(p, 0.25) for p in params]

    model = torch.nn.Linear(3, 2)
    opt = FakeAdam(list(model.parameters()))
    sch = loop.PlateauScheduler(opt)

    class A:
        pass
    args = A()
    args.g_dim, args.run = 90, "x"
    fn = tmp_path / "ck.pth"
    loop.save_checkpoint(fn, args, 7, model, opt, sch, 1.5, [(1.0, 5)], [(2.0, 5)], [(3.0, 5)], [(4.0, 5)])
    ck = torch.load(fn, weights_only=True)
    w_saved = model.weight.detach().clone()
    assert set(ck) == {"args", "epoch", "best_ade", "model_state_dict", "optimizer_state_dict", "lr_scheduler_state_dict",
                       "ade_list_val", "fde_list_val", "ade_list_test", "fde_list_test"}
    ref = torch.optim.Adam(model.parameters(), lr=1e-3)
    model(torch.ones(1, 3)).sum().backward()
    ref.step()
    rsd = ref.state_dict()
    osd = ck["optimizer_state_dict"]
    assert set(osd) == set(rsd) and set(osd["state"][0]) >= {"step", "exp_avg", "exp_avg_sq"}
    assert set(osd["param_groups"][0]) >= {"lr", "betas", "eps", "weight_decay", "amsgrad", "params"}
    ref.load_state_dict(osd)                           # torch's own optimizer accepts it
    opt2 = FakeAdam(list(model.parameters()))
    opt2.step_count = 0
    model2 = torch.nn.Linear(3, 2)
    got = loop.load_checkpoint(fn, model2, opt2, loop.PlateauScheduler(opt2))
    assert got["epoch"] == 7 and opt2.step_count == 3 and torch.equal(opt2.m[0], opt.m[0])
    assert torch.equal(model2.weight, w_saved)


# ------------------------------------------------------------------------------------------------
# SDD window extraction (models/dagnet/data/loader_sdd.py:65-227) against the reference's own output
# ------------------------------------------------------------------------------------------------
SDD_GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "sdd_windows.npz"))


def _sdd_source():
    files, limits = [], {}
    for i in range(int(SDD_GOLD["n_files"])):
        name = f"scene{i}"
        files.append((name, SDD_GOLD[f"file{i}_rows"]))
        limits[name] = dict(zip(("x_min", "x_max", "y_min", "y_max"), SDD_GOLD[f"file{i}_limits"].tolist()))
    return files, limits


@pytest.mark.parametrize("skip", [1, 3])
def test_sdd_trajectory_dataset_matches_the_reference_bit_for_bit(skip):
    """tests/golden/sdd_windows.npz = the reference's TrajectoryDataset on synthetic tracks (oracle/make_golden.py sdd_windows):
    windows, relative coordinates, 4-decimal rounding, goal cells, loss mask, the non-linearity flags and the reference's
    growing-prefix sequences (its `> min_ped` test sits inside the pedestrian loop) - all bit-exact."""
    from dagnet_amd import data
    ds = data.TrajectoryDataset(_sdd_source(), obs_len=4, pred_len=6, skip=skip, n_cells_x=5, n_cells_y=4, goals_window=3)
    for k in ("obs_traj", "pred_traj", "obs_traj_rel", "pred_traj_rel", "obs_goals", "pred_goals", "loss_mask", "non_linear_ped"):
        want = SDD_GOLD[f"skip{skip}_{k}"]
        got = getattr(ds, k).numpy()
        assert got.shape == want.shape, k
        assert np.array_equal(got, want), k
    assert np.array_equal(np.asarray(ds.seq_start_end), SDD_GOLD[f"skip{skip}_seq_start_end"])
    assert [len(ds), ds.__max_agents__()] == SDD_GOLD[f"skip{skip}_len"].tolist()
    for j, t in enumerate(ds[3]):
        assert np.array_equal(t.numpy(), SDD_GOLD[f"skip{skip}_item3_{j}"])
    # the collated tuple of a DataLoader over it is what the models take
    batch = data.seq_collate_sdd([ds[i] for i in range(4)])
    assert batch[0].shape[0] == 4 and batch[4].shape[0] == 4 and batch[-1].shape == (4, 2)
    # and the time-major fields feed SceneBatcher
    sb = data.SceneBatcher(ds.fields(), ds.scene_ptr, 5, shuffle=False, device="cpu")
    first = next(iter(sb))
    assert torch.equal(first[0], ds.obs_traj[:int(ds.scene_ptr[5])].permute(2, 0, 1))


def test_sdd_trajectory_dataset_reads_the_reference_file_layout(tmp_path):
    """<root>/sdd_npy/train.npy (object array of (scene, rows)) + <root>/limits.json, as loader_sdd.py:97-98,107 read them."""
    import json
    from dagnet_amd import data
    files, limits = _sdd_source()
    (tmp_path / "sdd_npy").mkdir()
    (tmp_path / "limits.json").write_text(json.dumps(limits))
    arr = np.empty(len(files), dtype=object)
    for i, f in enumerate(files):
        arr[i] = f
    np.save(tmp_path / "sdd_npy" / "train.npy", arr, allow_pickle=True)
    ds = data.TrajectoryDataset(str(tmp_path / "sdd_npy" / "train.npy"), obs_len=4, pred_len=6, skip=1, n_cells_x=5, n_cells_y=4,
                                goals_window=3)
    assert np.array_equal(ds.obs_traj.numpy(), SDD_GOLD["skip1_obs_traj"])
    assert np.array_equal(ds.pred_goals.numpy(), SDD_GOLD["skip1_pred_goals"])


def test_sub_seeds_do_not_collide():
    """seeds of call sites were sums (2 (seed + batch) + phase; epoch functions at +1009 / +2003 / +3001), so
    (seed, batch b + 1) met (seed + 1, batch b) and validate's batch 1009 met test's batch 0; and every rank drew the same
    noise for its row i.  Now a hash of (seed, batch, phase, rank) / (seed, epoch, function)."""
    from dagnet_amd import loop
    seen = {}
    for seed in range(4):
        for b in range(40):
            for phase in (0, 1):
                for rank in range(3):
                    v = loop.sub_seed(seed, b, phase, rank)
                    assert 0 <= v < (1 << 63)
                    assert v not in seen, (seed, b, phase, rank, seen.get(v))
                    seen[v] = (seed, b, phase, rank)
    assert loop.sub_seed(None, 1, 0) is None and loop.sub_seed(5, 1, 0, 0) == loop.sub_seed(5, 1, 0, 0)
    ep = {loop.epoch_seed(s, e, w) for s in range(3) for e in range(1, 50) for w in range(4)}
    assert len(ep) == 3 * 49 * 4 and loop.epoch_seed(None, 1, 0) is None
    # an epoch function's batches never meet another function's: (epoch_seed(.., 0), batch 1009) vs (epoch_seed(.., 1), batch 0)
    assert loop.sub_seed(loop.epoch_seed(7, 3, 0), 1009, 0, 0) != loop.sub_seed(loop.epoch_seed(7, 3, 1), 0, 0, 0)
