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
