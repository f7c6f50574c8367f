"""Diagnostic: gradient of the actor-critic term wrt pi (value-encoder dX path) in full-step context."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys
import numpy as np
import torch


def main(B=64):
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
    agent.update_step = oracle.update_step = 2
    for o in (agent.state_feat_val_encoder_optim, agent.critic_optim, oracle.val_encoder_optim, oracle.critic_optim):
        o.param_groups[0]["lr"] = 0.0
    mem = BaseMemory(3000, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 3000, seed=5)
    rng = np.random.default_rng(1)
    batch = sample_valid_batch(mem, B, rng)
    u = rng.random((B, 6)).astype(np.float32)
    agent.update_parameters(batch, agent.update_step, 0, noise_u=u)
    rt = agent._rt
    torch.cuda.synchronize()
    # oracle: AC term only, same (unchanged) critic / value encoder, pi as a leaf
    t, m = oracle._load(batch)
    oracle.state_feature_extractor.train()
    pi = rt.pi.cpu().clone().requires_grad_(True)
    vf = oracle.features(t["point_state_batch"], t["time_batch"], pi)
    vf.retain_grad()
    q1, q2, _ = oracle.critic(vf)
    keep = ~m["expert_reward"]
    loss = -0.1 * torch.min(q1.squeeze()[keep], q2.squeeze()[keep]).mean()
    loss.backward()

    def rel(a, b, name):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        print("%-22s max|ref| %.3e  max|err| %.3e  rel %.3e" % (name, np.abs(b).max(), np.abs(a - b).max(),
                                                                np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)))
    rel(rt.hs_cpi.out[:, :2].cpu().numpy(), torch.stack([q1.squeeze(), q2.squeeze()], 1).detach().numpy(), "Q(s,pi)")
    rel(rt.hs_cpi.g_feat.cpu().numpy(), vf.grad[:, :512].numpy(), "dL/dfeature")
    rel(rt.slot_v.daction.cpu().numpy(), pi.grad.numpy(), "dL/dpi")
    print("daction sample", rt.slot_v.daction[:2].cpu().numpy(), "\noracle", pi.grad[:2].numpy())
    # total dLoss/dpi = BC + AC, against what the policy backward consumed (hs_p.g_out / tanh')
    pi2 = rt.pi.cpu().clone().requires_grad_(True)
    em = m["expert"]
    bc = ref_step.pose_bc_loss(pi2[em], t["expert_action_batch"][em]) * 0.9
    bc.backward()
    g_tot = pi2.grad + pi.grad
    scale = torch.tensor(ref_step.ACTION_HIGH, dtype=torch.float32)
    th = rt.pi.cpu() / scale
    mine = rt.hs_p.g_out[:, :6].cpu() / (scale * (1 - th * th))
    rel(mine.numpy(), g_tot.numpy(), "dL/dpi total (from g_out)")
    rel((mine - pi2.grad).numpy(), pi.grad.numpy(), "  minus BC = AC part")
    rel(rt.slot_v.daction.cpu().numpy(), (mine - pi2.grad).numpy(), "  daction vs consumed AC")
    # forward features
    fc2 = rt.venc.fc_mats[1]; o = rt.venc.bn_off[fc2.bn_index]
    z = torch.relu(rt.slot_v.Zfc[1] * rt.slot_v.scale[o:o + 512] + rt.slot_v.shift[o:o + 512]).cpu().numpy()
    rel(z, vf[:, :512].detach().numpy(), "value_pi feature")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 64)
