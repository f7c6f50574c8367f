"""Diagnostic: pure host enqueue time of update_parameters(sync=False) (the first HOST_RING - 1 calls after a full
synchronise return as soon as everything is enqueued) against the GPU step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ga_ddpg_amd.api import make_agent
from ga_ddpg_amd.core.replay_memory import BaseMemory
from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
from ga_ddpg_amd.runtime import BATCH_KEYS
from ga_ddpg_amd.parallel import mask_counts
agent, cfg = make_agent("ddpg_td3_aux.yaml")
B = 256
mem = BaseMemory(2000, cfg, point_dtype=np.float32)
fill_synthetic_buffer(mem, 2000, seed=1)
rng = np.random.default_rng(1)
ring = []
for _ in range(4):
    hb = sample_valid_batch(mem, B, rng)
    d = {k: torch.as_tensor(np.ascontiguousarray(hb[k], dtype=np.float32)).cuda() for k in BATCH_KEYS}
    d["mask_counts"] = mask_counts(hb)
    ring.append(d)
torch.cuda.synchronize()
ev = torch.cuda.Event(); ev.record()
for d in ring:
    d["ready_event"] = ev
for i in range(20):
    agent.update_parameters(ring[i % 4], agent.update_step, i, sync=False)
agent.flush(); torch.cuda.synchronize()
host = []
for rep in range(10):
    for k in range(2):                 # two calls after a full sync: nothing to wait for
        t0 = time.perf_counter()
        agent.update_parameters(ring[k], agent.update_step, k, sync=False)
        host.append((time.perf_counter() - t0, agent.update_step % 2))
    agent.flush(); torch.cuda.synchronize()
pol = [h for h, p in host if p == 1]; non = [h for h, p in host if p == 0]
print("host enqueue per call: policy-step %.2f ms, other %.2f ms (median)" % (1e3 * np.median(pol), 1e3 * np.median(non)))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(100):
    agent.update_parameters(ring[i % 4], agent.update_step, i, sync=False)
agent.flush(); torch.cuda.synchronize()
print("steady state: %.2f ms per step" % (1e3 * (time.perf_counter() - t0) / 100))
