"""Diagnostic: host enqueue time vs GPU time of one update step (B=256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
import torch


def main():
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
    hb = sample_valid_batch(mem, B, rng)
    d = {k: torch.as_tensor(np.ascontiguousarray(hb[k], dtype=np.float32)).cuda() for k in BATCH_KEYS}
    d["mask_counts"] = mask_counts(hb)
    rt = agent.runtime(B, hb["point_state_batch"].shape[2])
    marks = []
    orig = rt._download

    def dl(*a, **k):
        marks.append(time.perf_counter())
        return orig(*a, **k)
    rt._download = dl
    for i in range(10):
        agent.update_parameters(d, agent.update_step, i)
    torch.cuda.synchronize()
    enq, tot = [], []
    for i in range(40):
        t0 = time.perf_counter()
        agent.update_parameters(d, agent.update_step, i)
        t1 = time.perf_counter()
        enq.append(marks[-1] - t0)
        tot.append(t1 - t0)
    print("per step: host enqueue %.2f ms (median), total %.2f ms, wait at the end %.2f ms" % (
        1e3 * np.median(enq), 1e3 * np.median(tot), 1e3 * np.median(np.array(tot) - np.array(enq))))


if __name__ == "__main__":
    main()
