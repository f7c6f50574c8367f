"""diagnostic: per-layer launch durations (in-kernel stamps) of the mid-size layers under the kernel-selection options,
alone (whole step on one stream) and inside the overlapped step, plus the step rate"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ga_ddpg_amd import engine, hip
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
    torch.cuda.synchronize(); ring[-1]["ready_event"] = torch.cuda.Event(); ring[-1]["ready_event"].record()
agent, nets = _filled_agent("ddpg_td3_aux.yaml", 77)
step_i = [0]


def step(sync=False):
    out = agent.update_parameters(ring[step_i[0] % 4], agent.update_step, 0, sync=sync)
    step_i[0] += 1
    return out


def rate(n=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


def probe(serial, n=6):
    engine.SERIAL = serial
    engine.timing_start("*", capacity=n * 200)
    for _ in range(n):
        step(sync=True)
    acc = engine.timing_stop()
    engine.SERIAL = False
    return {k: 1e3 * float(np.mean(v)) for k, v in acc.items()}, {k: len(v) / n for k, v in acc.items()}


configs = [dict()]            # CONFIGS="fwd_stream=0;dx_stream=0,dw_stream=0": kernel-family switches (gad_set_option), default 1
extra = os.environ.get("CONFIGS")
if extra:
    configs = [dict()] + [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in cfg.split(",")) for cfg in extra.split(";")]
for _ in range(6):
    step(sync=True)
base = None
for opts in configs:
    for k in ("fwd_stream", "dx_stream", "dw_stream", "fwd_skinny", "dx_skinny", "dw_skinny", "fwd_wide", "dx_wide", "dw_wide", "bwd_fused", "bwd_wide"):
        hip.set_option(k, opts.get(k, 0 if k == "bwd_wide" else 1))
    for _ in range(4):
        step(sync=True)
    alone, cnt = probe(True)
    inside, _ = probe(False)
    r = rate()
    print("==== options", opts, " steps/s %.1f" % r)
    tags = sorted(t for t in alone if any(t.startswith(p) for p in ("fwd.sa2", "fwd.sa3", "dx.sa2", "dx.sa3", "dw.sa2", "dw.sa3", "bwd.sa2", "bwd.sa3", "bwd.sa1", "dx.sa1", "dw.sa1")))
    tot_a = tot_i = 0.0
    for t in tags:
        tot_a += alone[t] * cnt[t]
        tot_i += inside.get(t, 0.0) * cnt[t]
        print("   %-12s x%4.1f/step  alone %6.1f us   in-step %6.1f us" % (t, cnt[t], alone[t], inside.get(t, float("nan"))))
    if not opts:
        print("   ---- every tagged launch, by time per step (alone)")
        for t in sorted(alone, key=lambda t: -alone[t] * cnt[t]):
            print("   %-16s x%4.1f/step  alone %6.1f us   in-step %6.1f us   per step alone %6.0f us" % (t, cnt[t], alone[t], inside.get(t, float("nan")), alone[t] * cnt[t]))
    print("   mid-size layers per step: alone %.0f us   in-step %.0f us;  all tagged launches alone %.0f us" %
          (tot_a, tot_i, sum(alone[t] * cnt[t] for t in alone)))
