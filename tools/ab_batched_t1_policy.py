"""A/B (timing emulation): what would ONE launch set over 2 x B rows for the two passes that share the
policy encoder's weights -- t1 = encoder(next state) for the TD target and the actor phase's encoder(state), reference
core/ddpg.py:69-86,160-170 -- cost against today's placement (t1 at the head of the main stream's chain, the policy pass on
the actor stream beside t2 and the critic backward)?

The emulation keeps the step's schedule and swaps two plans of the B = 256 runtime:
  * "t1"    <- the t1 plan of a second FusedRuntime built for 2 x B rows (the same networks; its own static inputs and
               geometry): an encoder pass + policy-target head over 2B rows, ~25 launches -- what the batched launch set costs
               (its two BatchNorm-statistic segments would add a row -> segment select per staged element, not modelled);
  * "p_fwd" <- the policy head only (the encoder part is inside the batched pass).
The numerics of the emulated step are meaningless (the actor backward reads stale activations): ONLY the step rate is read.
Dispatches per step drop by ~25 (one encoder pass), as item 3's "done" criterion asks.

    python tools/ab_batched_t1_policy.py [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    from ga_ddpg_amd import heads
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.parallel import mask_counts
    from ga_ddpg_amd.runtime import BATCH_KEYS, FusedRuntime
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
    rt = agent.runtime(B, ring[0]["point_state_batch"].shape[2])

    def rate(n):
        for i in range(30):
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

    base = [rate(steps)]
    # ---- the 2B-row pass: a second runtime over the same networks, its inputs filled once
    big = FusedRuntime(agent, 2 * B, ring[0]["point_state_batch"].shape[2])
    hb2 = sample_valid_batch(mem, 2 * B, rng)
    big_batch = {k: torch.as_tensor(np.ascontiguousarray(hb2[k], dtype=np.float32)).cuda() for k in BATCH_KEYS}
    for i in range(len(big._sets)):
        big._bind_set(i)
        big.upload(big_batch)
        big.geo_next.run(big.dbuf["next_point_state_batch"])
        big.geo.run(big.dbuf["point_state_batch"])
    torch.cuda.synchronize()
    saved = []
    for i, st in enumerate(rt._sets):
        P = st["plans"]
        saved.append((P["t1"], P["p_fwd"]))
        rt._bind_set(i)
        head_only = heads.plan_policy_forward(rt.pol, rt.hs_p, rt.enc, rt.slot_p, rt.dbuf["time_batch"])
        P["t1"] = big._sets[i % len(big._sets)]["plans"]["t1"]
        P["p_fwd"] = head_only
    emu = [rate(steps)]
    for i, st in enumerate(rt._sets):
        st["plans"]["t1"], st["plans"]["p_fwd"] = saved[i]
    base.append(rate(steps))
    for i, st in enumerate(rt._sets):
        st["plans"]["t1"] = big._sets[i % len(big._sets)]["plans"]["t1"]
        rt._bind_set(i)
        st["plans"]["p_fwd"] = heads.plan_policy_forward(rt.pol, rt.hs_p, rt.enc, rt.slot_p, rt.dbuf["time_batch"])
    emu.append(rate(steps))
    print("two passes on two streams (today):           %.1f  %.1f steps/s" % tuple(base))
    print("one 2B-row pass at the head of the chain:    %.1f  %.1f steps/s   (timing emulation)" % tuple(emu))


if __name__ == "__main__":
    main()
