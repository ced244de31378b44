"""CPU: `python bench.py --gpus N` started plainly (no torchrun, no WORLD_SIZE) must start N ranks itself, and the line rank 0
prints must carry the number of ranks the process group saw.  --rehearse swaps RCCL for gloo and skips the GPU work, so the
launcher, the environment hand-over, the sharding arithmetic and the reductions run here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                     # ONE JSON line on stdout, whatever the ranks chatter about
    return json.loads(lines[0])


def test_self_launch_two_ranks_weak_scaling():
    line = _run("--gpus", "2", "--rehearse", "--steps", "2", "--warmup", "0")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["rehearsal"] is True
    assert line["config"]["scenes_per_gpu"] == 4096 and line["config"]["rows_all_ranks"] == 2 * 4096 * 10
    assert line["config"]["trajectories_per_step"] == 2 * 4096 * 10 * 20


def test_self_launch_shards_the_one_million_scenes():
    line = _run("--gpus", "2", "--rehearse", "--steps", "1", "--warmup", "0", "--workload", "dagnet_bsk_1Mx20")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["scenes_per_gpu"] == (1 << 20) // 2 and line["config"]["rows_all_ranks"] == (1 << 20) * 10


def test_launcher_environment_wins_over_self_launch():
    """Under torch.distributed.run the ranks already exist: --gpus must match WORLD_SIZE, nothing is spawned."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr


def test_self_launch_parent_makes_no_hip_call(monkeypatch, tmp_path):
    """The NON-rehearse branch up to the spawn: the parent counts GPUs from the KFD topology files (no
    torch.cuda.device_count(), which may fall back to hipGetDeviceCount), starts one child per rank with the rendezvous
    environment, relays rank 0's JSON line and leaves torch.cuda uninitialised."""
    import argparse
    import io
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    # a fake KFD topology: one CPU node, three GPU nodes
    for i, simd in enumerate((0, 1024, 1024, 1024)):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {16 if simd == 0 else 0}\nsimd_count {simd}\nmax_waves_per_simd 8\n")
    assert bench.visible_gpu_count(str(tmp_path), env={}) == 3
    assert bench.visible_gpu_count(str(tmp_path), env={"HIP_VISIBLE_DEVICES": "0,2"}) == 2
    assert bench.visible_gpu_count(str(tmp_path / "missing"), env={}) is None

    def no_hip(*a, **k):
        raise AssertionError("the self-launching parent must not ask torch.cuda for anything")
    monkeypatch.setattr(torch.cuda, "device_count", no_hip)
    monkeypatch.setattr(torch.cuda, "is_available", no_hip)
    monkeypatch.setattr(torch.cuda, "set_device", no_hip)
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            started.append((cmd, env))
            self.stdout = io.StringIO('rank 0 chatter\n{"metric": "x", "n_gpus": 2}\n') if env["RANK"] == "0" else None

        def wait(self):
            return 0

        def poll(self):
            return 0
    import subprocess as sp
    monkeypatch.setattr(sp, "Popen", FakeProc)
    opt = argparse.Namespace(gpus=2, rehearse=False)
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 1)
    with pytest.raises(SystemExit, match="only 1 visible"):
        bench.self_launch(opt, ["--gpus", "2"])
    assert not started
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: None)       # topology unreadable: spawn, the ranks fail by themselves
    out = io.StringIO()
    monkeypatch.setattr(sys, "stdout", out)
    with pytest.raises(SystemExit) as ex:
        bench.self_launch(opt, ["--gpus", "2", "--steps", "1"])
    monkeypatch.undo()
    assert ex.value.code == 0 and len(started) == 2
    assert out.getvalue().strip() == '{"metric": "x", "n_gpus": 2}'
    for r, (cmd, env) in enumerate(started):
        assert cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "2", "--steps", "1"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"]) == (str(r), str(r), "2", "127.0.0.1")
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["MASTER_PORT"] == started[0][1]["MASTER_PORT"]
    assert not torch.cuda.is_initialized()


def test_walk_cap_keeps_the_one_million_scene_workload_inside_the_time_limit():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.walk_cap(256, 20, 5) == (3, 1)          # one GPU: 4 walks of 256 launches
    assert bench.walk_cap(32, 20, 5) == (20, 5)          # eight GPUs: the request fits
    assert bench.walk_cap(256, 1, 0) == (1, 0)
    assert bench.walk_cap(2048, 20, 5) == (1, 1)
