"""Diagnostic: is the run-ahead loop host-bound?  Host time to ENQUEUE one update step (sync=False, no waiting on the GPU: the
pinned rings are deep enough for 3 steps) against the GPU's step time, for policy and non-policy steps.
    python tools/diag_host_runahead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    # (a) pure enqueue cost: two steps at a time from an idle GPU (the host never waits for a ring slot)
    enq = {True: [], False: []}
    for rep in range(30):
        for _ in range(2):
            pol = agent.update_step % agent.policy_update_gap == 0
            t0 = time.perf_counter()
            agent.update_parameters(ring[rep % 4], agent.update_step, rep, sync=False)
            enq[pol].append(time.perf_counter() - t0)
        agent.flush(); torch.cuda.synchronize()
    # (b) the steady-state loop
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        agent.update_parameters(ring[i % 4], agent.update_step, i, sync=False)
    t_host = time.perf_counter() - t0
    agent.flush(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("host enqueue per step from an idle GPU: policy steps %.2f ms, other steps %.2f ms (median)" % (
        1e3 * np.median(enq[True]), 1e3 * np.median(enq[False])))
    print("steady state: %d steps enqueued in %.1f ms (%.2f ms/step of host time incl. waits for ring slots), all complete after %.1f ms (%.2f ms/step)"
          % (n, 1e3 * t_host, 1e3 * t_host / n, 1e3 * t_all, 1e3 * t_all / n))


if __name__ == "__main__":
    main()
