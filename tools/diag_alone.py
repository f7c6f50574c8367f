"""Diagnostic: per-layer GEMM launch durations of the B=256 update step with the step serialised on one stream
(engine.SERIAL: every kernel has the GPU to itself), every tagged launch stamped by the kernel itself (engine.timing_start).
    python tools/diag_alone.py [tag-prefix ...]      e.g.  python tools/diag_alone.py dw. dx."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    from ga_ddpg_amd import engine
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from ga_ddpg_amd.runtime import BATCH_KEYS
    from ga_ddpg_amd.parallel import mask_counts
    agent, cfg = make_agent("ddpg_td3_aux.yaml")
    B = 256
    mem = BaseMemory(2000, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 2000, seed=1)
    hb = sample_valid_batch(mem, B, np.random.default_rng(1))
    d = {k: torch.as_tensor(np.ascontiguousarray(hb[k], dtype=np.float32)).cuda() for k in BATCH_KEYS}
    d["mask_counts"] = mask_counts(hb)
    for i in range(6):
        agent.update_parameters(d, agent.update_step, i)
    engine.SERIAL = True
    engine.timing_start("*")
    for i in range(8):
        agent.update_parameters(d, agent.update_step, i)
    by = {t: [1e3 * x for x in v] for t, v in engine.timing_stop().items()}
    engine.SERIAL = False
    pre = sys.argv[1:] or [""]
    tot = 0.0
    for tag in sorted(by):
        if any(tag.startswith(p) for p in pre):
            v = by[tag]
            tot += float(np.sum(v)) / 8
            print("%-12s calls/step %4.1f  avg %7.1f us  min %7.1f" % (tag, len(v) / 8.0, float(np.mean(v)), float(np.min(v))))
    print("sum per step: %.3f ms" % (tot / 1e3))


if __name__ == "__main__":
    main()
