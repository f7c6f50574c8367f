"""Diagnostic: per-launch-tag durations of the DDPG step (every tagged launch stamped by the kernel itself: engine.timing_start)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys
from collections import defaultdict

import numpy as np
import torch


def main(B=256, steps=6):
    from ga_ddpg_amd import engine
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    torch.manual_seed(0)
    agent, cfg = make_agent("ddpg_td3_aux.yaml")
    mem = BaseMemory(4000, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 4000, seed=1)
    rng = np.random.default_rng(0)
    batch = sample_valid_batch(mem, B, rng)
    for i in range(4):
        agent.update_parameters(batch, agent.update_step, i)
    engine.timing_start("*")
    for i in range(steps):
        agent.update_parameters(batch, agent.update_step, i)
    acc = defaultdict(list)
    for tag, v in engine.timing_stop().items():
        acc[tag] = [1e3 * x for x in v]
    rt = agent._rt
    print("rows: sa1 %d  sa2 %d  sa3 %d" % tuple(int(rt.geo.rows[s]["n"].item()) for s in range(3)))
    tot = 0.0
    for tag in sorted(acc, key=lambda t: -sum(acc[t])):
        v = np.array(acc[tag])
        tot += v.sum() / steps
        print("%-14s calls/step %5.1f  avg %8.1f us  min %8.1f  per-step %8.1f us" % (tag, len(v) / steps, v.mean(), v.min(), v.sum() / steps))
    print("tagged GEMM time per step: %.2f ms" % (tot / 1e3))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
