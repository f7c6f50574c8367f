import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import golden_batch
from tests.test_gpu_step import _filled_agent, SEED

def snap(opt):
    from ga_ddpg_amd import hip
    hip.set_option("fwd_stream", opt)
    g32 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ddpg_steps_B32.npz"))
    agent, nets = _filled_agent("ddpg_td3_aux.yaml", SEED)
    agent.update_step = 1
    agent.update_parameters(golden_batch(g32, "a0/"), 1, 0, noise_u=g32["a0/noise_u"])
    torch.cuda.synchronize()
    hip.set_option("fwd_stream", 1)
    rt = agent._rt
    sv = rt.slot_v
    n = [int(rt.geo.rows[s]["n"].item()) for s in range(3)]
    out = {"rows": n, "g_feat": rt.hs_c.g_feat.clone(), "Gfc": sv.Gfc.clone(), "mean": sv.mean.clone(), "istd": sv.istd.clone(),
           "scale": sv.scale.clone(), "shift": sv.shift.clone(), "coef": sv.coef.clone(),
           "bstats": sv.bstats.view(8, 2, -1).sum(0).clone(), "stats": sv.stats.view(8, 2, -1).sum(0).clone(),
           "Zfc1": sv.Zfc[0].clone(), "Zfc2": sv.Zfc[1].clone(), "y": rt.y.clone(), "out9": rt.hs_c.out.clone(),
           "g_out9": rt.hs_c.g_out.clone(), "venc_grad": rt.venc.flat.grad.clone(), "cr_grad": rt.cr.flat.grad.clone(),
           "a_next": rt.a_next.clone(), "tgt9": rt.hs_ct.out.clone()}
    for s in range(3):
        for l in range(3):
            out["Z%d%d" % (s + 1, l + 1)] = sv.Z[s][l][:n[s]].clone()
        out["F%d" % (s + 1)] = sv.F[s].clone(); out["dF%d" % (s + 1)] = sv.dF[s].clone()
    return out

a, b = snap(1), snap(0)
print(a["rows"], b["rows"])
for k in a:
    if k == "rows": continue
    x, y = a[k].double(), b[k].double()
    sc = float(y.abs().max()) + 1e-30
    d = (x - y).abs()
    print("%-10s max %.3e median %.3e (scale %.3e)" % (k, float(d.max()) / sc, float(d.median()) / sc, sc))
d = (a["g_feat"] - b["g_feat"]).abs()
idx = (d > 1e-3 * float(b["g_feat"].abs().max())).nonzero()
print("g_feat entries that differ by > 1e-3 of max:", idx.shape[0], "of", d.numel())
o = 512 * 0
tot = a["scale"].numel()
# fc2 is the last BN: its 512 channels are the last 512 of the per-slot vectors
for (r, c) in idx[:12].tolist():
    for nm, s in (("stream", a), ("tile", b)):
        z = float(s["Zfc2"][r, c]); sc = float(s["scale"][tot - 512 + c]); sh = float(s["shift"][tot - 512 + c])
        print("  row %2d ch %3d %-6s z %.7f scale %.7f shift %.7f  y=fma %.3e  g_feat %.4e" % (r, c, nm, z, sc, sh, np.float32(np.float32(z) * np.float32(sc) + np.float32(sh)), float(s["g_feat"][r, c])))

for nm in ("Gfc", "dF3", "dF2"):
    dd = (a[nm].double() - b[nm].double()).abs()
    print(nm, "entries differing > 1e-3 of max:", int((dd > 1e-3 * float(b[nm].abs().max())).sum()), "of", dd.numel())
