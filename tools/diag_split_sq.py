import sys, os
sys.path.insert(0, "/root/repo")
import torch
from tests import split_cases as sc
from ga_ddpg_amd import hip
case = sc.FwdWide(27240, 128, 128, "act", ragged=False)
case.row_w.fill_(1.0)
bad = 0
worst = 0.0
blocks = set()
for rep in range(40):
    out, routed = case.run_mode(True)
    z = out["z"].double()
    d = (out["stat_sq"] - (z * z).sum(0)).abs()
    if d.max().item() > 1e-3:
        bad += 1
        worst = max(worst, d.max().item())
        blocks |= set(int(i) for i in range(4) if d[i*32:(i+1)*32].max().item() > 1e-3)
print(os.environ.get("GAD_LIB_PATH", "default"), routed, "bad runs %d / 40, worst %.3e, blocks %s" % (bad, worst, sorted(blocks)))
