"""DDPG / TD3-style agent with goal-auxiliary heads (reference core/ddpg.py).  Same constructor,
update_parameters(batch_data, updates, k, test=False) -> dict of the 11 floats, get/load_weight."""
import numpy as np

from .agent import Agent
from .utils import get_critic, get_valid_index


class DDPG(Agent):
    def __init__(self, num_inputs, action_space, args):
        super(DDPG, self).__init__(num_inputs, action_space, args, name="DDPG")
        self.critic_num_input = num_inputs + 1
        self.critic_value_dim = 0
        if not (self.value_model and self.sa_channel_concat):
            raise NotImplementedError("only the shipped sa_channel_concat/value_model critic form is on the path")
        self.critic, self.critic_optim, self.critic_scheduler, self.critic_target = get_critic(self)

    def load_weight(self, weights):
        self.policy.load_state_dict(weights[0])
        self.critic.load_state_dict(weights[1])
        self.goal_feature_extractor.load_state_dict(weights[2])
        self.state_feature_extractor.load_state_dict(weights[3])
        self._after_external_weight_change()

    def get_weight(self):
        return [self.policy.state_dict(), self.critic.state_dict(), self.goal_feature_extractor.state_dict(),
                self.state_feature_extractor.state_dict()]

    def _after_external_weight_change(self):
        rt = self._rt
        if rt is not None:
            for f in (rt.pol.flat, rt.pol_t.flat, rt.enc.flat, rt.venc.flat, rt.cr.flat, rt.cr_t.flat):
                f.sync_packed()

    def get_mix_ratio(self, update_step):
        """reference ddpg.py:108-117"""
        idx = sum(1 for m in self.mix_milestones if self.update_step > m)
        mix_policy_ratio = min(get_valid_index(self.mix_policy_ratio_list, idx), self.ddpg_coefficients[4])
        mix_value_ratio = min(get_valid_index(self.mix_value_ratio_list, idx), self.ddpg_coefficients[3])
        return mix_value_ratio, mix_policy_ratio

    def update_parameters(self, batch_data, updates, k, test=False, noise_u=None, sync=True):
        """One gradient step.  `noise_u` (B,6) optionally injects the uniform draw of the TD3
        target-policy noise (the reference draws it with torch.rand_like, core/utils.py:575).
        sync=False: return as soon as the step is enqueued -- the result dict fills in on first read (PendingLog), the
        host goes on to sample / stage / enqueue the next step while this one runs (agent.flush() waits for all)."""
        self.mix_value_ratio, self.mix_policy_ratio = self.get_mix_ratio(self.update_step)
        # test=True (reference core/agent.py:276-280): the SAME update with the online networks in eval mode -- BatchNorm on its
        # running statistics in every pass, forward and backward (FusedRuntime.ddpg_step(test=True)); no driver of the reference
        # calls it that way, tests/golden/ddpg_steps_test_mode_B32.npz pins it
        self.set_mode(test)
        if self._dp is not None:
            self._dp.step_hook()
        ps = batch_data["point_state_batch"]
        rt = self.runtime(ps.shape[0], ps.shape[2])
        s = rt.ddpg_step(batch_data, noise_u=noise_u, sync=sync, test=bool(test))
        self.update_step += 1
        # tensors the reference leaves on the agent after a step
        self.pi, self.aux_pred = rt.pi, rt.aux_pred
        self.qf1, self.qf2 = rt.hs_c.out[:, 0], rt.hs_c.out[:, 1]
        self.next_q_value, self.critic_grasp_aux = rt.y, rt.critic_aux_norm
        return self._result(s, True) if sync else self._pending_result(s, True)
