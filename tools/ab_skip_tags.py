"""Marginal value of a kernel family inside the overlapped step (timing only): the step rate with every tagged launch whose
tag starts with one of the given prefixes SKIPPED (stale buffers are read instead: the numbers of such a step are garbage,
its schedule and every other launch are unchanged).  Answers "how much faster would the step be if this family cost
nothing?" -- the upper bound of any optimisation of that family, which its stand-alone duration does not tell (a launch
on a side lane that fills the chain's gaps is nearly free; one on the dX chain costs its full duration).

    python tools/ab_skip_tags.py "dw." "dx.sa2,dx.sa3" "fwd.sa1.l1" "name:gad_bn_finalize" "untagged:gad_gemm_fwd" ...
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    from ga_ddpg_amd import engine
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.parallel import mask_counts
    from ga_ddpg_amd.runtime import BATCH_KEYS
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
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
    ev = torch.cuda.Event()
    ev.record()
    for d in ring:
        d["ready_event"] = ev
    agent.runtime(B, ring[0]["point_state_batch"].shape[2])
    skip = {"prefixes": ()}
    plain_run = engine.Plan.run

    def run(self):
        pre = skip["prefixes"]
        if not pre:
            return plain_run(self)
        plans = tuple(x.split(":", 1)[1] for x in pre if x.startswith("plan:"))            # whole plans of the runtime ("plan:p_fwd")
        if plans:
            rt = agent._rt
            for st in rt._sets:
                if any(st["plans"].get(k) is self for k in plans):
                    return
        tagp = tuple(x for x in pre if ":" not in x and x not in ("optim", "geometry"))
        names = tuple(x.split(":", 1)[1] for x in pre if x.startswith("name:"))            # every call of that entry point
        untag = tuple(x.split(":", 1)[1] for x in pre if x.startswith("untagged:"))        # its untagged calls (the heads' GEMMs)
        key = (id(self), pre)
        filt = filtered.get(key)
        if filt is None or filt.n_src != len(self.calls):
            filt = engine.Plan()                    # the same items minus the skipped launches (compiled and replayed like any plan)
            filt.calls = [it for it in self.calls if not (it.kind == "call" and (
                (tagp and it.tag is not None and it.tag.startswith(tagp)) or it.name in names or (it.name in untag and it.tag is None)))]
            filt.keep = [self]
            filt.n_src = len(self.calls)
            filtered[key] = filt
        return plain_run(filt)
    filtered = {}
    engine.Plan.run = run
    from ga_ddpg_amd import runtime as rtm
    rtm.STEP_PLAN = False       # the step enqueued phase by phase (each plan still replayed in C): whole phases can be skipped from here
    plain_optim = rtm.FusedRuntime._optim_phase

    def optim(self, which, policy_step):                       # token "optim": the fused optimiser launches (gad_optim_jobs)
        if "optim" in skip["prefixes"]:
            return
        return plain_optim(self, which, policy_step)
    rtm.FusedRuntime._optim_phase = optim
    plain_geo = engine.Geometry.run

    def geo_run(self, pts):                                    # token "geometry": FPS / ball query / row compaction of both cloud sets
        if "geometry" in skip["prefixes"]:
            return
        return plain_geo(self, pts)
    engine.Geometry.run = geo_run

    def rate(n=150):
        for i in range(20):
            agent.update_parameters(ring[i % 8], agent.update_step, i, sync=False)
        agent.flush()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            agent.update_parameters(ring[i % 8], agent.update_step, i, sync=False)
            agent.step_scheduler(agent.update_step)
        agent.flush()
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t0)

    base = rate()
    print("%-44s %.1f steps/s  (%.3f ms/step)" % ("nothing skipped", base, 1e3 / base))
    for arg in sys.argv[1:]:
        skip["prefixes"] = tuple(arg.split(","))
        r = rate()
        print("%-44s %.1f steps/s  (%.3f ms/step, %+.3f ms)" % ("skipped: " + arg, r, 1e3 / r, 1e3 / r - 1e3 / base))
    skip["prefixes"] = ()
    r = rate()
    print("%-44s %.1f steps/s" % ("nothing skipped (again)", r))


if __name__ == "__main__":
    main()
