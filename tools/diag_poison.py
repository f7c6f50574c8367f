"""diagnostic: every torch.empty* float buffer starts as NaN; a buffer read before it is written shows up as NaN in the step's outputs"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

_empty, _empty_like = torch.empty, torch.empty_like
POISON = os.environ.get("POISON", "nan")
POISON = None if POISON == "none" else float(POISON)
ONLY = os.environ.get("POISON_ONLY", "")          # "file.py:lo-hi": only allocations made from these lines
HITS = {}


def _want():
    f = sys._getframe(2)
    key = "%s:%d" % (os.path.basename(f.f_code.co_filename), f.f_lineno)
    if ONLY:
        fn, rng = ONLY.split(":")
        lo, hi = [int(x) for x in rng.split("-")]
        if not (key.startswith(fn + ":") and lo <= f.f_lineno <= hi):
            return False
    HITS[key] = HITS.get(key, 0) + 1
    return True


def empty(*a, **k):
    t = _empty(*a, **k)
    if POISON is not None and t.is_floating_point() and t.is_cuda and _want():
        t.fill_(POISON)
    return t


def empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if POISON is not None and t.is_floating_point() and t.is_cuda and _want():
        t.fill_(POISON)
    return t


torch.empty, torch.empty_like = empty, empty_like
from tests.test_gpu_step import _filled_agent
from ga_ddpg_amd.core.replay_memory import BaseMemory
from ga_ddpg_amd.experiments.config import load_cfg
from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch

c = load_cfg("ddpg_td3_aux.yaml")
mem = BaseMemory(1500, c, point_dtype=np.float32)
fill_synthetic_buffer(mem, 1500, seed=5)


def run(tag, poison_all=False):
    rng = np.random.default_rng(9)
    agent, nets = _filled_agent("ddpg_td3_aux.yaml", 77)
    res = []
    for s in range(4):
        batch = sample_valid_batch(mem, 32, rng)
        u = rng.random((32, 6)).astype(np.float32)
        out = agent.update_parameters(batch, agent.update_step, 0, noise_u=u)
        torch.cuda.synchronize()
        keys = ("critic_loss", "bc_loss", "actor_critic_loss", "policy_grasp_aux_loss", "critic_grasp_aux_loss")
        for nn, net in nets.items():
            for n, q in net.named_parameters():
                bad_p, bad_g = int(torch.isnan(q).sum()), (int(torch.isnan(q.grad).sum()) if q.grad is not None else -1)
                big = int((q.abs() > 1e20).sum()) + (int((q.grad.abs() > 1e20).sum()) if q.grad is not None else 0)
                if bad_p or bad_g > 0 or big:
                    print("step", s, nn, n, tuple(q.shape), "nan params", bad_p, "nan grads", bad_g, "huge", big)
        res.append([out[k] for k in keys] + [float(agent.pi.abs().sum()), float(agent.qf1.abs().sum())])
    print(tag, np.array(res))
    return np.array(res)


a = run("first ")
print("poisoned sites:", sorted(HITS.items()))
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "poison_%s.npy" % os.environ.get("POISON", "nan")), a)
if os.environ.get("SECOND"):
    b = run("second")
    print("max |first - second|", np.abs(a - b).max(0))
