"""Behaviour-cloning agent (reference core/bc.py): policy + policy encoder trained on the expert
point-matching loss and the goal-auxiliary loss."""
from .agent import Agent


class BC(Agent):
    def __init__(self, num_inputs, action_space, args):
        super(BC, self).__init__(num_inputs, action_space, args, name="BC")

    def load_weight(self, weights):
        self.policy.load_state_dict(weights[0])
        self.goal_feature_extractor.load_state_dict(weights[1])
        self.state_feature_extractor.load_state_dict(weights[2])
        if self._rt is not None:
            for f in (self._rt.pol.flat, self._rt.enc.flat, self._rt.venc.flat):
                f.sync_packed()

    def get_weight(self):
        return [self.policy.state_dict(), self.goal_feature_extractor.state_dict(),
                self.state_feature_extractor.state_dict()]

    def update_parameters(self, batch_data, updates, k):
        self.set_mode(False)
        ps = batch_data["point_state_batch"]
        rt = self.runtime(ps.shape[0], ps.shape[2])
        s = rt.bc_step(batch_data)
        self.update_step += 1
        self.pi, self.aux_pred = rt.pi, rt.aux_pred
        return self._result(s, False)
