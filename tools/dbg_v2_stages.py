"""Debug: dump v2 intermediate tiles of step 0 and compare with the oracle (VRNN)."""
import ctypes, os, sys
os.environ["DAGNET_V2_DUMP"] = "1"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from dagnet_amd import _capi
_capi._LIB_PATH = os.path.join(ROOT, "dagnet_amd", "_lib", "libdagnet_hip_prof.so")
from tests import cases
import torch.nn.functional as F
meta, gold = cases.load_golden("vrnn_bsk")
args, model, sd = cases.build_model(meta); model = model.to("cuda:0")
_, _, eps_s, _ = cases.build_inputs(meta)
h = torch.from_numpy(gold["fwd_h"])
dbg = torch.zeros(8 * 4 * 64 * 16, dtype=torch.float32, device="cuda:0")
lib = _capi.load(); lib.dagnet_debug_set_prof.argtypes = [ctypes.c_void_p]
lib.dagnet_debug_set_prof(ctypes.c_void_p(dbg.data_ptr()))
s = model.sample_k(2, 1, h.cuda(), eps=eps_s[:, :1].cuda().contiguous())
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(8, 4, 64, 16)
def untile(stage, ntiles):       # -> [rows 32][features ntiles*32]
    out = np.zeros((32, ntiles * 32), np.float32)
    for nt in range(ntiles):
        for lane in range(64):
            lh, m = lane >> 5, lane & 31
            for r in range(16):
                out[m, nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = d[stage, nt, lane, r]
    return out
lin = lambda n, x: F.linear(x, sd[n + ".weight"], sd[n + ".bias"])
lr = lambda x: F.leaky_relu(x, 0.01)
hl = h[-1][:20]
t0 = lr(lin("prior.0", hl)); t1 = lr(lin("prior.2", t0)); mu = lin("prior_mean", t1); lv = lin("prior_logvar", t1)
z = eps_s[0, 0][:20] * torch.exp(0.5 * lv) + mu
pz = lr(lin("phi_z.2", lr(lin("phi_z.0", z))))
da = lr(lin("dec.0", torch.cat([pz, hl], 1))); db = lr(lin("dec.2", da))
for name, stage, nt, ref in (("h", 7, 2, hl), ("prior0", 0, 2, t0), ("prior2", 1, 2, t1), ("ml", 2, 2, torch.cat([mu, lv], 1)), ("z", 3, 1, z),
                             ("pz", 4, 2, pz), ("dec0", 5, 2, da), ("dec2", 6, 2, db)):
    got = untile(stage, nt)[:20]
    if name == "prior0": print("bias-only lrelu:", lr(sd["prior.0.bias"])[:8].numpy()); print(got[0,:8], ref[0,:8].numpy(), (lin("prior.0", hl))[0,:8].numpy())
    print(f"{name:8s} max|diff| = {np.abs(got - ref.numpy()).max():.3e}   (ref scale {ref.abs().max():.2f})")
