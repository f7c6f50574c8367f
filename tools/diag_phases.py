"""Diagnostic: per phase of one update step, WHEN the host enqueues it and WHEN the GPU runs it (untraced: HIP events
recorded on the phase's own stream + perf_counter).  A phase whose GPU start follows its host enqueue by microseconds is
waiting for the host, not for data.
    python tools/diag_phases.py"""
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
    hb = sample_valid_batch(mem, B, rng)
    d = {k: torch.as_tensor(np.ascontiguousarray(hb[k], dtype=np.float32)).cuda() for k in BATCH_KEYS}
    d["mask_counts"] = mask_counts(hb)
    rt = agent.runtime(B, hb["point_state_batch"].shape[2])
    if os.environ.get("GAD_DIAG_DP", "0") == "1":                    # the data-parallel hooks over a one-rank RCCL group
        import torch.distributed as dist
        from ga_ddpg_amd.parallel import DataParallelContext
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1)
        dp = DataParallelContext()
        agent._dp = dp
        dp.attach(rt)
        for a in ("set_counts", "reduce_scalars"):
            pass
    log = []
    on = [False]

    def wrap(obj, attr, name):
        f = getattr(obj, attr)

        def g(*a, **k):
            if not on[0]:
                return f(*a, **k)
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            h0 = time.perf_counter()
            e0.record(st)
            r = f(*a, **k)
            e1.record(st)
            log.append((name, h0, time.perf_counter(), e0, e1))
            return r
        setattr(obj, attr, g)

    for k, p in rt.plans.items():
        if p is not None:
            wrap(p, "run", k)
    wrap(rt.geo, "run", "geo(cur)")
    wrap(rt.geo_next, "run", "geo(next)")
    for a in ("_adam", "_stats", "_target_updates", "_download", "upload", "_reduce"):
        wrap(rt, a, a)
    if rt.dp is not None:
        wrap(rt.dp, "set_counts", "dp.set_counts")
        wrap(rt.dp, "reduce_scalars", "dp.reduce_scalars")
    for i in range(12):
        agent.update_parameters(d, agent.update_step, i)
    torch.cuda.synchronize()
    for rep in range(2):
        del log[:]
        on[0] = True
        es = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        es.record(torch.cuda.current_stream())
        agent.update_parameters(d, agent.update_step, 100 + rep)
        t1 = time.perf_counter()
        on[0] = False
        torch.cuda.synchronize()
        print("step (update_step %d, %s): host total %.2f ms" % (agent.update_step - 1, "policy" if (agent.update_step - 1) % 2 == 0
                                                                 else "non-policy", 1e3 * (t1 - t0)))
        print("  %-16s %21s   %21s" % ("phase", "host enqueue [ms]", "GPU [ms]"))
        for name, h0, h1, e0, e1 in log:
            print("  %-16s %9.3f -> %8.3f   %9.3f -> %8.3f" % (name, 1e3 * (h0 - t0), 1e3 * (h1 - t0), es.elapsed_time(e0),
                                                                es.elapsed_time(e1)))


if __name__ == "__main__":
    main()
