"""diagnostic: update-loop rates over a HOST replay buffer: synchronous sample -> update, and PrefetchSampler + run-ahead"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ga_ddpg_amd.core.prefetch import PrefetchSampler
from ga_ddpg_amd.core.replay_memory import BaseMemory
from ga_ddpg_amd.experiments.config import load_cfg
from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
from tests.test_gpu_step import _filled_agent

B = 256
c = load_cfg("ddpg_td3_aux.yaml")
mem = BaseMemory(20000, c, point_dtype=np.float32)
fill_synthetic_buffer(mem, 20000, seed=5)
rng = np.random.default_rng(9)
agent, nets = _filled_agent("ddpg_td3_aux.yaml", 77)
for i in range(10):
    agent.update_parameters(sample_valid_batch(mem, B, rng), agent.update_step, i)
t0 = time.perf_counter()
for i in range(20):
    sample_valid_batch(mem, B, rng)
print("sampling alone: %.2f ms per minibatch" % ((time.perf_counter() - t0) / 20 * 1e3))
n = 60
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(n):
        agent.update_parameters(sample_valid_batch(mem, B, rng), agent.update_step, i)
    torch.cuda.synchronize()
    r_sync = n / (time.perf_counter() - t0)
    for depth in (3,):
        with PrefetchSampler(mem, B, depth=depth, sample=lambda bs: sample_valid_batch(mem, bs, rng)) as s:
            for i in range(5):
                agent.update_parameters(s.next(), agent.update_step, i, sync=False)
            agent.flush()
            t0 = time.perf_counter()
            for i in range(n):
                agent.update_parameters(s.next(), agent.update_step, i, sync=False)
            agent.flush()
            r_pre = n / (time.perf_counter() - t0)
    print("rep %d: synchronous %.1f steps/s   prefetch + run-ahead %.1f steps/s" % (rep, r_sync, r_pre))
