"""Diagnostic (not a test): 3000 DDPG update steps at B=256 fed by the GPU-resident replay mirror -- finite losses,
learning curves, device memory.  Round-1 run: 265 steps/s including sampling; critic_loss 0.072 -> 0.011, aux losses
0.36 -> 0.25, bc_loss flat (the synthetic expert actions are noise).
    python tools/diag_long_run.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ga_ddpg_amd.api import make_agent
from ga_ddpg_amd.core.replay_memory import BaseMemory
from ga_ddpg_amd.core.device_replay import DeviceReplay
from ga_ddpg_amd.synth_data import fill_synthetic_buffer
torch.manual_seed(0)
agent, cfg = make_agent("ddpg_td3_aux.yaml")
mem = BaseMemory(20000, cfg, point_dtype=np.float32)
fill_synthetic_buffer(mem, 20000, seed=1)
dmem = DeviceReplay(mem)
rng = np.random.default_rng(0)
hist = []
agent.update_parameters(dmem.sample(256, rng), agent.update_step, 0)      # builds the runtime (static buffers)
m0 = torch.cuda.memory_allocated()
t0 = time.time()
SYNC = __import__("os").environ.get("SYNC", "0") == "1"        # 0: run-ahead steps (results read 8 steps late)
logs = []
for i in range(3000):
    b = dmem.sample_lazy(256, rng)
    if b["mask_counts"][1] == 0 or b["mask_counts"][2] == 0:
        continue
    logs.append((i, agent.update_parameters(b, agent.update_step, i, sync=SYNC)))
    agent.step_scheduler(agent.update_step)
    while len(logs) > (0 if SYNC else 8):
        j, r = logs.pop(0)
        hist.append([r["bc_loss"], r["critic_loss"], r["policy_grasp_aux_loss"], r["critic_grasp_aux_loss"]])
        if not all(np.isfinite(v) for v in r.values()):
            print("non-finite at", j, dict(r)); break
agent.flush()
for j, r in logs:
    hist.append([r["bc_loss"], r["critic_loss"], r["policy_grasp_aux_loss"], r["critic_grasp_aux_loss"]])
h = np.array(hist)
assert np.isfinite(h).all()
print("mode:", "sync each step" if SYNC else "run-ahead")
print("steps %d in %.1f s (%.1f steps/s incl. sampling); memory growth %.1f MB" % (len(h), time.time() - t0, len(h) / (time.time() - t0), (torch.cuda.memory_allocated() - m0) / 1e6))
for name, col in zip(("bc_loss", "critic_loss", "policy_aux", "critic_aux"), h.T):
    print("%-12s first 100: %.4f   last 100: %.4f" % (name, col[:100].mean(), col[-100:].mean()))
