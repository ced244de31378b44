"""GPU: the data-parallel branches of dagnet_amd.loop under a real (one-rank) RCCL process group.  Collected after the other GPU
tests and before tests/test_zz_no_alloc.py; it starts a child process (the process group lives and dies there)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
_DIST_SCRIPT = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
import dagnet_amd
from dagnet_amd import data as D, loop, synthetic as syn
from tests.test_gpu_loop import _split, TINY
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
print("PG_OK", flush=True)
try:
    sizes = [2, 5, 3, 4, 2, 6]
    out = {}
    for sync in (False, True):
        args, model, loader, _ = _split("dagnet", sizes, 4, 5, 9, 3, dict(TINY, n_layers=2))
        os.environ["DAGNET_SYNC_BN"] = "1" if sync else "0"
        torch.manual_seed(0)                  # the latent noise of the training steps comes from torch's generator: same in both runs
        hist = loop.fit(args, model, loader, loader, loader, seed=7)
        out["sync" if sync else "plain"] = dict(train=hist["train"], val=hist["val"], best=hist["best_ade"])
    print("RESULT " + json.dumps(out))
finally:
    dist.destroy_process_group()
"""


def test_fit_under_a_one_rank_process_group_with_synchronised_batchnorm():
    """loop.fit with torch.distributed initialised (RCCL, world size 1 - all this box has): the data-parallel branches of the loop
    (row fractions, gradient all-reduce, all-reduced evaluation sums, synchronised BatchNorm through the collective path) run for
    real, and with one rank synchronised statistics ARE the batch statistics: both runs must give the same history."""
    import json
    import os
    import subprocess
    import sys
    import socket
    with socket.socket() as sk:               # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _DIST_SCRIPT], capture_output=True, text=True, timeout=110, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if "PG_OK" not in r.stdout:
        pytest.skip("no RCCL process group on this box: " + r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(next(l for l in r.stdout.splitlines() if l.startswith("RESULT "))[7:])
    a, b = out["plain"], out["sync"]
    assert len(a["train"]) == len(b["train"]) == 2
    for ha, hb in zip(a["train"] + a["val"], b["train"] + b["val"]):
        for k in ha:
            # (the staged statistics differ from the one-pass ones in the last bits, and four Adam steps amplify that)
            assert np.isfinite(hb[k]) and abs(ha[k] - hb[k]) <= 5e-3 * max(1.0, abs(ha[k])), (k, ha[k], hb[k])
