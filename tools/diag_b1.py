"""Diagnostic (not a test): golden DDPG run b (steps b0,b1): HIP vs CPU oracle vs golden, per-sample TD target."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os

import numpy as np
import torch


def main():
    from tests.helpers import golden_batch
    from tests.test_gpu_step import _filled_agent, SEED
    from ga_ddpg_amd.experiments.config import load_cfg
    from oracle import ref_step
    from oracle.detfill import fill_module_
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ddpg_steps_B32.npz"))
    agent, nets = _filled_agent("ddpg_td3_aux.yaml", SEED)
    oracle = ref_step.OracleAgent(load_cfg("ddpg_td3_aux.yaml").RL_TRAIN)
    for name, net in oracle.nets().items():
        fill_module_(net, name, SEED)
    agent.update_step = oracle.update_step = 2
    for s in range(2):
        p = "b%d/" % s
        batch = golden_batch(g, p)
        u = g[p + "noise_u"]
        agent.update_parameters(batch, agent.update_step, s, noise_u=u)
        agent.step_scheduler()
        oracle.update_parameters(batch, noise_u=u)
        oracle.step_scheduler()
        yg = g[p + "t/next_q_value"]
        yh = agent.next_q_value.cpu().numpy()
        yo = oracle.dbg["y"].numpy()
        print(p, "y: |hip-golden| %.3e  |oracle-golden| %.3e  |hip-oracle| %.3e" %
              (np.abs(yh - yg).max(), np.abs(yo - yg).max(), np.abs(yh - yo).max()))
        bad = np.argsort(-np.abs(yh - yg))[:4]
        for i in bad:
            print("    sample %2d  hip %.6f oracle %.6f golden %.6f  reward %.3f mask %.1f" %
                  (i, yh[i], yo[i], yg[i], batch["reward_batch"][i], batch["mask_batch"][i]))
        rt = agent._rt
        for nm, mine, theirs in (("pol_t", rt.pol_t.flat, oracle.policy_target), ("cr_t", rt.cr_t.flat, oracle.critic_target),
                                 ("pol", rt.pol.flat, oracle.policy), ("cr", rt.cr.flat, oracle.critic)):
            on = dict(theirs.named_parameters())
            worst = 0.0
            for n, q in zip(mine.names, mine.params):
                worst = max(worst, float((q.detach().cpu() - on[n].detach()).abs().max()))
            # packed vs master consistency
            chk = mine.packed.clone()
            mine.sync_packed()
            print("    %-6s max|param - oracle| %.3e   packed stale by %.3e" % (nm, worst, float((chk - mine.packed).abs().max())))


if __name__ == "__main__":
    main()
