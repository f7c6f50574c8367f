"""configs[3] at B = 128 (train mode, forward + backward): per-tensor difference between route sets.
python tools/diag_config4_routes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ga_ddpg_amd import hip
from tests.test_gpu_config4 import _stacks, _cloud

B, N = 128, 4096
xyz, feats = _cloud(B, N, 7)
xyz_d = xyz.cuda()
probe = torch.tensor(np.random.default_rng(9).normal(size=(B, 256, 128)), dtype=torch.float32).cuda()


def run(opts):
    for k, v in opts.items():
        hip.set_option(k, v)
    try:
        mine, _ = _stacks()
        mine = [m.cuda().train() for m in mine]
        f = feats.cuda().requires_grad_(True)
        x1, f1 = mine[0](xyz_d, f)
        _, out = mine[1](x1, f1)
        (out * probe).sum().backward()
        torch.cuda.synchronize()
        g = {"sa%d.%s" % (i, n): p.grad.clone() for i in range(2) for n, p in mine[i].named_parameters() if p.grad is not None}
        g["out"] = out.detach().clone()
        g["dfeat"] = f.grad.clone()
        g["f1"] = f1.detach().clone()
        return g
    finally:
        for k in opts:
            hip.set_option(k, hip.get_option_default("mfma_split") if k == "mfma_split" else 1)


ALL = ("fwd_stream", "fwd_wide", "dx_stream", "dx_wide", "dw_stream", "dw_wide")
sets = {"default": {}, "f32 specialised": {"mfma_split": 0}, "generic": dict({k: 0 for k in ALL}, mfma_split=0)}
for k in ALL:
    sets["no " + k] = {k: 0}
res = {k: run(v) for k, v in sets.items()}
ref = res["generic"]
for name, r in res.items():
    if name == "generic":
        continue
    print("== %s vs generic" % name)
    for k in ref:
        s = float(ref[k].abs().max())
        e = (r[k] - ref[k]).abs()
        flag = "  <<<" if float(e.median()) > 1e-4 * s and s > 1e-6 else ""
        print("   %-28s scale %.3e  median %.3e  max %.3e%s" % (k, s, float(e.median()), float(e.max()), flag))
