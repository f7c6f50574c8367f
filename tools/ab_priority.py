"""A/B: the step's main stream (the TD-target / critic chain) as a HIGH-PRIORITY HIP stream (side lanes: normal priority).
    PRIO=0 python tools/ab_priority.py; PRIO=1 python tools/ab_priority.py      (result: profiles/r04_negative_results.txt item 7)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch

def main():
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.parallel import mask_counts
    from ga_ddpg_amd.runtime import BATCH_KEYS
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    prio = int(os.environ.get("PRIO", "0"))
    torch.manual_seed(1234)
    agent, cfg = make_agent("ddpg_td3_aux.yaml")
    B = 256
    mem = BaseMemory(6000, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 6000, seed=20260928)
    rng = np.random.default_rng(1)
    ring = []
    for _ in range(8):
        hb = sample_valid_batch(mem, B, rng)
        d = {k: torch.as_tensor(np.ascontiguousarray(hb[k], dtype=np.float32)).cuda() for k in BATCH_KEYS}
        d["mask_counts"] = mask_counts(hb)
        ring.append(d)
    torch.cuda.synchronize()
    main_stream = torch.cuda.Stream(priority=-1) if prio else torch.cuda.current_stream()
    with torch.cuda.stream(main_stream):
        ev = torch.cuda.Event(); ev.record()
        for d in ring: d["ready_event"] = ev
        def rate(n):
            for i in range(30):
                agent.update_parameters(ring[i % 8], agent.update_step, i, sync=False)
            agent.flush(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                agent.update_parameters(ring[i % 8], agent.update_step, i, sync=False)
                agent.step_scheduler(agent.update_step)
            agent.flush(); torch.cuda.synchronize()
            return n / (time.perf_counter() - t0)
        print("PRIO=%d: %.1f %.1f steps/s" % (prio, rate(200), rate(200)))
main()
