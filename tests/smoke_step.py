"""__graft_entry__.smoke(): one small DDPG update step on cuda:0 through the HIP path, checked
against the CPU oracle on the same seeded batch."""
import numpy as np
import torch


def run(B=32):
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
    mem = BaseMemory(400, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 400, seed=11)
    rng = np.random.default_rng(1)
    for s in range(2):
        batch = sample_valid_batch(mem, B, rng)
        u = rng.random((B, 6)).astype(np.float32)
        got = agent.update_parameters(batch, agent.update_step, s, noise_u=u)
        want = oracle.update_parameters(batch, noise_u=u)
        # step 0 starts from identical parameters: 1e-4 relative (actor_critic_loss is evaluated after the critic's
        # Adam step of the same update: 3e-2); step 1 follows an Adam step: sanity bound
        for k in ("critic_loss", "critic_grasp_aux_loss", "bc_loss", "policy_grasp_aux_loss", "actor_critic_loss"):
            rt = (3e-2 if k == "actor_critic_loss" else 1e-4) if s == 0 else 5e-2
            assert abs(got[k] - want[k]) <= rt * abs(want[k]) + 1e-6, (s, k, got[k], want[k])
    torch.cuda.synchronize()
    print("smoke ok:", {k: round(v, 6) for k, v in got.items()})
