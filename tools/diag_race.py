"""diagnostic: the same first update step on fresh twin agents, many times: spread of the step's scalars (a race or a stale
buffer shows up as an outlier; run-to-run atomics order alone gives ~1e-6)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.test_gpu_step import _filled_agent
from ga_ddpg_amd.core.replay_memory import BaseMemory
from ga_ddpg_amd.experiments.config import load_cfg
from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch

c = load_cfg("ddpg_td3_aux.yaml")
mem = BaseMemory(1500, c, point_dtype=np.float32)
fill_synthetic_buffer(mem, 1500, seed=5)
rng = np.random.default_rng(9)
B = int(os.environ.get("B", 32))
batch = sample_valid_batch(mem, B, rng)
u = rng.random((B, 6)).astype(np.float32)
N = int(os.environ.get("N", 30))
keys = ("critic_loss", "bc_loss", "actor_critic_loss", "policy_grasp_aux_loss", "critic_grasp_aux_loss", "critic_grad", "policy_grad")
for start in (2, 1):
    res = []
    for i in range(N):
        agent, nets = _filled_agent("ddpg_td3_aux.yaml", 77)
        agent.update_step = start
        out = agent.update_parameters(batch, agent.update_step, 0, noise_u=u)
        torch.cuda.synchronize()
        res.append([out.get(k, 0.0) for k in keys] + [float(agent.pi.double().abs().sum()), float(agent.qf1.double().abs().sum()),
                   float(agent._rt.venc.flat.master.double().abs().sum()), float(agent._rt.enc.flat.master.double().abs().sum()),
                   float(agent._rt.cr.flat.master.double().abs().sum()), float(agent._rt.pol.flat.master.double().abs().sum())])
        if os.environ.get("JUNK"):
            junk = [torch.randn(1 << 22, device="cuda") * (10.0 ** (i % 5)) for _ in range(16)]
            del junk
    r = np.array(res)
    med = np.nanmedian(r, 0)
    dev = np.abs(r - med) / (np.abs(med) + 1e-30)
    print("start", start, "median", med)
    print("   max rel dev per column", dev.max(0))
    bad = np.nonzero((dev > 1e-6).any(1))[0].tolist()
    print("   trials with any rel dev > 1e-6:", bad)
    for i in bad:
        print("   trial", i, "rel dev", np.array2string(dev[i], precision=2))
