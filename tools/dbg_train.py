"""Per-parameter gradient error of the HIP training step against a golden case (debug aid)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests import cases
from tests.test_gpu_train import hip_train_step, DEV

name = sys.argv[1] if len(sys.argv) > 1 else "train_dagnet_bsk_gat_n5"
meta, gold = cases.load_golden(name)
_, model, _ = cases.build_model(meta)
out = hip_train_step(meta, model.to(DEV))
print({k: (float(out[k]), float(gold[k])) for k in ("kld", "nll", "ce", "loss")}, float(out["total_norm"]), float(gold["total_norm"]))
rows = []
for key, want in gold.items():
    if key.startswith("grad/"):
        got = out["grads"][key[5:]].numpy()
        rows.append((np.abs(got - want).max() / (np.abs(want).max() + 1e-12), np.abs(got - want).max(), np.abs(want).max(), key))
for r in sorted(rows, reverse=True)[:14]:
    print("%.2e  abs %.2e  max|g| %.2e  %s" % r)
