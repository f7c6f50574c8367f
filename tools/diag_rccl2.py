"""diagnostic: does the mere existence of an RCCL communicator slow the step?  plain steps, then ncclCommInitRank, then again"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
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


def rate(n=120):
    for i in range(10):
        agent.update_parameters(ring[i % 4], agent.update_step, 0, sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for i in range(n):
        h0 = time.perf_counter()
        agent.update_parameters(ring[i % 4], agent.update_step, 0, sync=False)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0), host / n * 1e3


print("plain                        : %.1f steps/s, host %.2f ms/step" % rate())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29613")
dist.init_process_group("nccl", rank=0, world_size=1)
from ga_ddpg_amd.parallel import DataParallelContext
import ga_ddpg_amd.parallel as par
par.BUCKETED = False
dp = DataParallelContext()
agent._dp = dp
rt = agent._rt
dp.attach(rt)
print("DP attached, direct           : %.1f steps/s, host %.2f ms/step" % rate())
comm = dp._comm
dp._comm, dp._direct = None, False
print("DP attached, torch collectives: %.1f steps/s, host %.2f ms/step" % rate())
dp._comm, dp._direct = comm, True
print("DP attached, direct again     : %.1f steps/s, host %.2f ms/step" % rate())
# direct, but without the per-step count exchange / scalar reduction
sc, rs = dp.set_counts, dp.reduce_scalars
dp.set_counts = lambda batch: None
print("direct, no count exchange     : %.1f steps/s, host %.2f ms/step" % rate())
dp.set_counts = sc
dp.reduce_scalars = lambda scal: None
print("direct, no scalar reduction   : %.1f steps/s, host %.2f ms/step" % rate())
dp.reduce_scalars = rs
ar = rt.allreduce
rt.allreduce = lambda ts: None
print("direct, no gradient exchange  : %.1f steps/s, host %.2f ms/step" % rate())
