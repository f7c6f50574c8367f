"""The SAME minibatch stepped N times with learning rate 0 (run-ahead or synchronous): the losses that depend on the online
networks only (bc_loss, policy_grasp_aux_loss, critic_grasp_aux_loss) must repeat to the atomics' rounding; a step that computed
with something wrong shows up as an outlier.   python tools/diag_runahead_repeat.py [steps] [sync|ahead] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ga_ddpg_amd.core.replay_memory import BaseMemory
from ga_ddpg_amd.experiments.config import load_cfg
from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
from tests.test_gpu_step import _filled_agent

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mode = sys.argv[2] if len(sys.argv) > 2 else "ahead"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
c = load_cfg("ddpg_td3_aux.yaml")
mem = BaseMemory(1500, c, point_dtype=np.float32)
fill_synthetic_buffer(mem, 1500, seed=5)
rng = np.random.default_rng(3)
batch = sample_valid_batch(mem, B, rng)
noise = rng.random((B, 6)).astype(np.float32)
agent, nets = _filled_agent("ddpg_td3_aux.yaml", 11)
for opt in (agent.policy_optim, agent.critic_optim, agent.state_feat_encoder_optim, agent.state_feat_val_encoder_optim):
    for g in opt.param_groups:
        g["lr"] = 0.0
logs = []
chk = torch.zeros(steps, 12, dtype=torch.float64, device="cuda")
for i in range(steps):
    logs.append(agent.update_parameters(batch, agent.update_step, 0, noise_u=noise, sync=(mode == "sync")))
    if os.environ.get("NO_CHECKSUMS") == "1":          # (the checksum launches change what overlaps what)
        continue
    rt = agent._rt
    used = rt._sets[rt._set]             # the input / geometry set the step just enqueued read (checksummed behind it on the main stream)
    row = []
    for geo in (used["geo"], used["geo_next"]):
        row += [geo.fps1.sum(dtype=torch.float64), geo.fps2.sum(dtype=torch.float64), geo.cnt1.sum(dtype=torch.float64), geo.cnt2.sum(dtype=torch.float64),
                geo.rows[0]["n"].sum(dtype=torch.float64), geo.xyz.sum(dtype=torch.float64)]
    chk[i] = torch.stack(row)
agent.flush()
chk = chk.cpu().numpy()
names = ["fps1", "fps2", "cnt1", "cnt2", "rows1", "xyz"]
for par in (0, 1):
    c = chk[par::2]
    ref = np.median(c, axis=0)
    for j in range(12):
        badj = np.nonzero(c[:, j] != ref[j])[0]
        if len(badj):
            print("geometry checksum %s (%s set, parity %d) differs in steps %s" % (names[j % 6], "state" if j < 6 else "next-state", par,
                                                                                 " ".join(str(2 * b + par) for b in badj[:12])))
logs = [dict(l) for l in logs]
keys = [k for k in ("bc_loss", "policy_grasp_aux_loss", "critic_grasp_aux_loss") if k in logs[0]]
bad = 0
for k in keys:
    v = np.array([l[k] for l in logs], dtype=np.float64)
    # policy steps and other steps may log different values: compare within each parity class
    for par in (0, 1):
        w = v[par::2]
        med = np.median(w)
        out = np.nonzero(np.abs(w - med) > 2e-5 * abs(med) + 1e-7)[0]
        bad += len(out)
        print("%-24s parity %d: median %.8f, spread %.2e, %d outliers of %d%s" % (
            k, par, med, float(np.max(np.abs(w - med))), len(out), len(w),
            ("  steps " + " ".join("%d(%.6f)" % (2 * i + par, w[i]) for i in out[:8])) if len(out) else ""))
print("mode %s B=%d mfma_split=%s: %d outlier values in %d steps" % (mode, B, os.environ.get("GAD_OPT_mfma_split", "default"), bad, steps))
# every logged key over the last quarter of the run (the target networks have converged onto the online ones by then: with
# learning rate 0 all of them -- gradient statistics included -- repeat up to the atomics' rounding)
tail = logs[3 * steps // 4:]
for k in sorted(tail[0]):
    v = np.array([l[k] for l in tail], dtype=np.float64)
    for par in (0, 1):
        w = v[par::2]
        med = np.median(w)
        dev = np.abs(w - med) / (abs(med) + 1e-12)
        print("   %-28s parity %d median %+.6e  max rel dev %.2e  #(> 1e-4) %d" % (k, par, med, float(dev.max()), int((dev > 1e-4).sum())))
if os.environ.get("OUT"):
    np.savez(os.environ["OUT"], **{k: np.array([l[k] for l in logs], dtype=np.float64) for k in logs[0]})
