"""Step rate of core.train_test_offline.train_off_policy over a synthetic HBM-mirrored buffer: the reference-shaped loop (a host
synchronisation per update; GAD_TRAIN_LOOKAHEAD=0/1: the next minibatch staged behind the running update) and run_ahead.
    python tools/diag_train_loop.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.core.train_test_offline import train_off_policy
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer
    agent, cfg = make_agent("ddpg_td3_aux.yaml")
    c = cfg.RL_TRAIN
    c.batch_size, c.updates_per_step, c.max_epoch = 256, 50, 10 ** 9
    mem = BaseMemory(20000, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 20000, seed=1)
    train_off_policy(agent, mem, c, max_epochs=2)
    for mode in ("sync loop (default)", "run_ahead"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        train_off_policy(agent, mem, c, max_epochs=6, run_ahead=(mode == "run_ahead"))
        torch.cuda.synchronize()
        print("GAD_TRAIN_LOOKAHEAD=%s  %-22s %.1f steps/s" % (os.environ.get("GAD_TRAIN_LOOKAHEAD", "1"), mode, 6 * 50 / (time.perf_counter() - t0)))


if __name__ == "__main__":
    main()
