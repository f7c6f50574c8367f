"""diagnostic: timeline of consecutive update steps from the in-kernel stamps of every tagged launch (one wall clock for all
streams): start / end relative to the first launch, launch order = host enqueue order"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ga_ddpg_amd import engine
from ga_ddpg_amd.core.replay_memory import BaseMemory
from ga_ddpg_amd.experiments.config import load_cfg
from ga_ddpg_amd.runtime import BATCH_KEYS
from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
from tests.test_gpu_step import _filled_agent

B = 256
c = load_cfg("ddpg_td3_aux.yaml")
mem = BaseMemory(4000, c, point_dtype=np.float32)
fill_synthetic_buffer(mem, 4000, seed=5)
rng = np.random.default_rng(9)
ring = []
for _ in range(4):
    hb = sample_valid_batch(mem, B, rng)
    ring.append({k: torch.as_tensor(np.ascontiguousarray(hb[k], dtype=np.float32)).cuda() for k in BATCH_KEYS})
torch.cuda.synchronize()
ev = torch.cuda.Event(); ev.record()
for r in ring:
    r["ready_event"] = ev
agent, nets = _filled_agent("ddpg_td3_aux.yaml", 77)
sync = os.environ.get("SYNC", "0") == "1"
for i in range(12):
    agent.update_parameters(ring[i % 4], agent.update_step, 0, sync=sync)
NS = 4
engine.timing_start("*", capacity=NS * 220)
for i in range(NS):
    agent.update_parameters(ring[i % 4], agent.update_step, 0, sync=sync)
spans = engine.timing_stop(spans=True)
t0 = min(a for _, a, _ in spans)
print("launches", len(spans), "span %.3f ms for %d steps" % (max(b for _, _, b in spans) - t0, NS))
busy = sorted((a, b) for _, a, b in spans)
# idle time: no tagged kernel running
cur_e = busy[0][1]; idle = 0.0
for a, b in busy[1:]:
    if a > cur_e:
        idle += a - cur_e
    cur_e = max(cur_e, b)
print("time with no tagged kernel in flight: %.3f ms" % idle)
for tag, a, b in spans:
    print("%8.1f %8.1f  %6.1f  %s" % ((a - t0) * 1e3, (b - t0) * 1e3, (b - a) * 1e3, tag))
