"""Diagnostic (not a test): error of the HIP step vs the CPU oracle as a function of batch size.
    python -m tools.diag_parity 32 128
Prints max-abs and max-norm-relative errors of one DDPG step that starts from identical parameters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys

import numpy as np
import torch


def one(B, policy_step, seed=5, freeze_critic=False):
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from oracle import ref_step
    from oracle.detfill import fill_module_
    agent, cfg = make_agent("ddpg_td3_aux.yaml")
    oracle = ref_step.OracleAgent(load_cfg("ddpg_td3_aux.yaml").RL_TRAIN)
    nets = {"policy": agent.policy, "policy_target": agent.policy_target, "critic": agent.critic,
            "critic_target": agent.critic_target, "state_feature_extractor": agent.state_feature_extractor}
    for name, net in nets.items():
        fill_module_(net, name, 3)
    for name, net in oracle.nets().items():
        fill_module_(net, name, 3)
    if policy_step:
        agent.update_step = oracle.update_step = 2
    if freeze_critic:     # no parameter change between the critic phase and the actor phase's Q(s,pi(s))
        for o in (agent.state_feat_val_encoder_optim, agent.critic_optim, oracle.val_encoder_optim, oracle.critic_optim):
            o.param_groups[0]["lr"] = 0.0
    mem = BaseMemory(3000, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 3000, seed=seed)
    rng = np.random.default_rng(1)
    batch = sample_valid_batch(mem, B, rng)
    u = rng.random((B, 6)).astype(np.float32)
    got = agent.update_parameters(batch, agent.update_step, 0, noise_u=u)
    want = oracle.update_parameters(batch, noise_u=u)
    rows = []

    def add(name, a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        e = np.abs(a - b).max()
        rows.append((name, e, e / max(np.abs(b).max(), 1e-30)))
    d = oracle.dbg
    add("pi", agent.pi.cpu().numpy(), d["pi"].numpy())
    add("aux_pred", agent.aux_pred.cpu().numpy(), d["aux_pred"].numpy())
    add("qf1", agent.qf1.cpu().numpy(), d["q1"].numpy())
    add("qf2", agent.qf2.cpu().numpy(), d["q2"].numpy())
    add("td_target y", agent.next_q_value.cpu().numpy(), d["y"].numpy())
    for k in ("critic_loss", "critic_grasp_aux_loss", "bc_loss", "policy_grasp_aux_loss", "actor_critic_loss"):
        add(k, got[k], want[k])
    for name in ("policy", "critic", "state_feature_extractor"):
        on = dict(oracle.nets()[name].named_parameters())
        worst = 0.0
        per = []
        for n, p in nets[name].named_parameters():
            if on[n].grad is None or ".1.0.bias" in n or ".1.3.bias" in n:
                continue
            if policy_step and "value_encoder" in n:
                continue
            g, w = p.grad.cpu().numpy().astype(np.float64), on[n].grad.numpy().astype(np.float64)
            r = np.abs(g - w).max() / max(np.abs(w).max(), 1e-30)
            per.append((r, n, np.abs(w).max()))
            worst = max(worst, r)
        rows.append(("grad(%s) worst tensor" % name, float("nan"), worst))
        for r, n, m in sorted(per, reverse=True)[:4]:
            rows.append(("    " + n[-24:], m, r))
    print("B=%d policy_step=%s freeze_critic=%s" % (B, policy_step, freeze_critic))
    for n, e, r in rows:
        print("   %-28s max|err| %.3e   /max|ref| %.3e" % (n, e, r))


if __name__ == "__main__":
    for b in [int(x) for x in sys.argv[1:]] or [32]:
        one(b, False)
        one(b, True)
        one(b, True, freeze_critic=True)
