"""Diagnostic: accuracy of the BatchNorm batch statistics the GEMM epilogues accumulate (one-pass sum / sum of squares)
against a float64 two-pass evaluation of the SAME raw layer outputs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import golden_batch
from tests.test_gpu_step import _filled_agent, SEED

g32 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ddpg_steps_B32.npz"))
agent, nets = _filled_agent("ddpg_td3_aux.yaml", SEED)
agent.update_step = 1
agent.update_parameters(golden_batch(g32, "a0/"), 1, 0, noise_u=g32["a0/noise_u"])
torch.cuda.synchronize()
rt = agent._rt
for nm, enc, slot in (("value", rt.venc, rt.slot_v), ("policy", rt.enc, rt.slot_p)):
    geo = slot.geo
    for s in range(3):
        r = geo.rows[s]
        n = int(r["n"].item())
        w = r["w"][:n].double()
        for l, m in enumerate(enc.sa_mats[s]):
            z = slot.Z[s][l][:n].double()
            cnt = w.sum()
            mean = (w[:, None] * z).sum(0) / cnt
            var = (w[:, None] * (z - mean) ** 2).sum(0) / cnt
            istd = 1.0 / torch.sqrt(var + 1e-5)
            o = enc.bn_off[m.bn_index]
            e_m = ((slot.mean[o:o + m.n_out].double() - mean).abs() / (var.sqrt() + 1e-12)).max()
            e_i = ((slot.istd[o:o + m.n_out].double() - istd).abs() / istd).max()
            print("%-6s sa%d.l%d rows %6d: max |mean err|/std %.2e  max rel istd err %.2e  max mean^2/var %.1e" % (
                nm, s + 1, l + 1, n, float(e_m), float(e_i), float((mean ** 2 / (var + 1e-30)).max())))
    for l, m in enumerate(enc.fc_mats):
        z = slot.Zfc[l].double()
        mean, var = z.mean(0), z.var(0, unbiased=False)
        istd = 1.0 / torch.sqrt(var + 1e-5)
        o = enc.bn_off[m.bn_index]
        e_m = ((slot.mean[o:o + m.n_out].double() - mean).abs() / (var.sqrt() + 1e-12)).max()
        e_i = ((slot.istd[o:o + m.n_out].double() - istd).abs() / istd).max()
        print("%-6s fc%d    rows %6d: max |mean err|/std %.2e  max rel istd err %.2e  max mean^2/var %.1e" % (
            nm, l + 1, z.shape[0], float(e_m), float(e_i), float((mean ** 2 / (var + 1e-30)).max())))
