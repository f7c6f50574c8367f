"""diagnostic: which tensor of the policy step's Q(s, pi(s)) pass differs first in an outlier trial"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.test_gpu_step import _filled_agent
from ga_ddpg_amd import engine
from ga_ddpg_amd.core.replay_memory import BaseMemory
from ga_ddpg_amd.experiments.config import load_cfg
from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch

if os.environ.get("SERIAL"):
    engine.SERIAL = True
c = load_cfg("ddpg_td3_aux.yaml")
mem = BaseMemory(1500, c, point_dtype=np.float32)
fill_synthetic_buffer(mem, 1500, seed=5)
rng = np.random.default_rng(9)
B = 32
batch = sample_valid_batch(mem, B, rng)
u = rng.random((B, 6)).astype(np.float32)
N = int(os.environ.get("N", 60))


def snapshot(agent):
    rt = agent._rt
    sv = rt.slot_v
    n = [int(r["n"].item()) for r in sv.geo.rows]
    t = {"pi": rt.pi, "venc.packed": rt.venc.flat.packed, "cr.packed": rt.cr.flat.packed, "scale": sv.scale, "shift": sv.shift,
         "hs_cpi.out": rt.hs_cpi.out, "g_feat": rt.hs_cpi.g_feat, "daction": sv.daction, "Zfc0": sv.Zfc[0], "Zfc1": sv.Zfc[1]}
    for s in range(3):
        t["F%d" % s] = sv.F[s]
        for l in range(3):
            t["Z%d%d" % (s, l)] = sv.Z[s][l][:n[s]]
    return {k: v.detach().clone() for k, v in t.items()}


ref = None
nbad = 0
for i in range(N):
    agent, nets = _filled_agent("ddpg_td3_aux.yaml", 77)
    agent.update_step = 2
    out = agent.update_parameters(batch, agent.update_step, 0, noise_u=u)
    torch.cuda.synchronize()
    snap = snapshot(agent)
    if ref is None:
        ref, ref_out = snap, out
        continue
    rel = abs(out["actor_critic_loss"] - ref_out["actor_critic_loss"]) / abs(ref_out["actor_critic_loss"])
    if rel > 1e-4:
        nbad += 1
        print("trial", i, "actor_critic_loss", out["actor_critic_loss"], "ref", ref_out["actor_critic_loss"])
        for k in snap:
            a, b = snap[k].double(), ref[k].double()
            fin = torch.isfinite(a) & torch.isfinite(b)
            d = ((a - b).abs() * fin).max().item() / (b.abs() * fin).max().item()
            nn = int((~torch.isfinite(a)).sum()), int((~torch.isfinite(b)).sum())
            if d > 1e-6 or nn[0] != nn[1]:
                print("    %-12s max|d|/max|ref| %.3e   differing entries %d of %d  nonfinite %s" % (k, d, int(((a - b).abs() * fin > 0).sum()), a.numel(), nn))
print("outliers:", nbad, "of", N - 1)
