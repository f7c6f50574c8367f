"""Diagnostic: run-to-run reproducibility of one DDPG update step (same init, same batch), per kernel option."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys
import numpy as np
import torch

from ga_ddpg_amd import hip


def run_once(B=32, policy_step=True):
    from tests.test_gpu_step import _filled_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    agent, nets = _filled_agent("ddpg_td3_aux.yaml", 3)
    agent.update_step = 2 if policy_step else 1
    c = load_cfg("ddpg_td3_aux.yaml")
    mem = BaseMemory(600, c, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 600, seed=5)
    rng = np.random.default_rng(1)
    batch = sample_valid_batch(mem, B, rng)
    u = rng.random((B, 6)).astype(np.float32)
    agent.update_parameters(batch, agent.update_step, 0, noise_u=u)
    torch.cuda.synchronize()
    out = {}
    for nn, net in nets.items():
        for n, p in net.named_parameters():
            out[nn + "/" + n] = p.detach().clone()
            if p.grad is not None:
                out["g:" + nn + "/" + n] = p.grad.clone()
    out["pi"] = agent.pi.clone()
    out["q1"] = agent.qf1.clone()
    return out


def main():
    for opts in ({}, {"dx_skinny": 0}, {"dx_skinny": 0, "fwd_skinny": 0}, {"dx_skinny": 0, "fwd_skinny": 0, "fwd_stream": 0}):
        for k in ("dx_skinny", "fwd_skinny", "fwd_stream"):
            hip.set_option(k, opts.get(k, 1))
        ref = run_once()
        worst = {}
        for rep in range(4):
            cur = run_once()
            for k in ref:
                d = float((cur[k].double() - ref[k].double()).abs().max())
                if d > 0:
                    worst[k] = max(worst.get(k, 0.0), d / (float(ref[k].abs().max()) + 1e-30))
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
        print("options %s: %d of %d tensors differ between runs; worst %s" % (opts, len(worst), len(ref), top))


if __name__ == "__main__":
    main()
