"""Network definitions with the reference's module surface and state-dict keys
(reference core/networks.py: base_network :65-92, GoalFeature :150-178, PointNetFeature :182-250,
QNetwork :253-300, GaussianPolicy :303-377).

The nn.Modules here only OWN parameters/buffers (so `state_dict()`, `load_state_dict()`,
`named_parameters()`, `.train()/.eval()` behave as in the reference); all arithmetic runs in
libgaddpg through ga_ddpg_amd.engine.  `forward()` evaluates the module with the HIP kernels
(inference / feature extraction, no autograd graph); training goes through Agent.update_parameters,
which drives the fused forward+backward plans directly.
The image branch (ResNetFeature, reference :106-147) is out of scope: `use_image` is False in every
shipped config (experiments/config.py:105).
"""
import numpy as np
import torch
from torch import nn

from ..pointnet2_ops import pointnet2_modules as pointnet2

LOG_SIG_MAX = 2
LOG_SIG_MIN = -10
epsilon = 1e-6


def base_network(pointnet_radius, pointnet_nclusters, scale, in_features):
    """3 set-abstraction modules + FC head; key layout `0.{0,1,2}.mlps.0.*` / `1.{0,1,3,4}.*`."""
    sa1_module = pointnet2.PointnetSAModule(npoint=pointnet_nclusters, radius=pointnet_radius, nsample=64,
                                            mlp=[in_features, 64 * scale, 64 * scale, 128 * scale])
    sa2_module = pointnet2.PointnetSAModule(npoint=32, radius=0.04, nsample=128,
                                            mlp=[128 * scale, 128 * scale, 128 * scale, 256 * scale])
    sa3_module = pointnet2.PointnetSAModule(mlp=[256 * scale, 256 * scale, 256 * scale, 512 * scale])
    sa_modules = nn.ModuleList([sa1_module, sa2_module, sa3_module])
    fc_layer = nn.Sequential(
        nn.Linear(int(512 * scale), int(1024 * scale)), nn.BatchNorm1d(int(1024 * scale)), nn.ReLU(True),
        nn.Linear(int(1024 * scale), int(512 * scale)), nn.BatchNorm1d(int(512 * scale)), nn.ReLU(True))
    return nn.ModuleList([sa_modules, fc_layer])


def weights_init_(m):
    if isinstance(m, nn.Linear):
        torch.nn.init.xavier_uniform_(m.weight, gain=1)
        torch.nn.init.constant_(m.bias, 0)


class GoalFeature(nn.Module):
    """Built by make_nets_opts_schedulers but never evaluated on the update path; forward() works all the
    same (tests/golden/offpath_forms.npz: the reference's class on the same weights)."""

    def __init__(self, input_dim=3, pointnet_radius=0.02, pointnet_nclusters=128, model_scale=1,
                 action_concat=False):
        super(GoalFeature, self).__init__()
        self.num_grasp_samples = 1
        self.encoder = base_network(pointnet_radius, pointnet_nclusters, model_scale, 3)
        self.q = nn.Linear(model_scale * 512, 4)
        self.t = nn.Linear(model_scale * 512, 3)
        self.confidence = nn.Linear(model_scale * 512, 1)

    def encode(self, xyz, xyz_features):
        """the three set-abstraction modules through libgaddpg (pointnet2_ops facade: differentiable, BatchNorm mode follows
        .training), then the FC head"""
        for sa in self.encoder[0]:
            xyz, xyz_features = sa(xyz, xyz_features)
        return self.encoder[1](xyz_features.squeeze(-1))

    def forward(self, pc, grasp=None, goal_head=False):
        """pc (B, N, 3) -> (unit quaternion | translation (B, 7), confidence (B,)); reference core/networks.py:171-178.
        Not on the update-step path (policy_goal / critic_goal are off in every shipped config): the SA stack runs through the
        HIP kernels, the 512 -> 1024 -> 512 head and the three small output layers through torch's library GEMMs on the GPU."""
        pc = pc.cuda()
        z = self.encode(pc[..., :3].contiguous(), pc.transpose(1, -1).contiguous())
        qt = torch.cat((nn.functional.normalize(self.q(z), p=2, dim=-1), self.t(z)), -1)
        return qt, torch.sigmoid(self.confidence(z)).squeeze()

    def grasp_pred(self, *args):
        return self(*args, goal_head=True)[0]


class PointNetFeature(nn.Module):
    """Two PointNet++ encoders: `encoder` for the policy (xyz + hand flag), `value_encoder` for the
    critic (+ 6 action channels broadcast over the points when action_concat)."""

    def __init__(self, input_dim=3, pointnet_nclusters=32, pointnet_radius=0.02, model_scale=1, extra_latent=0,
                 split_feature=False, policy_extra_latent=-1, critic_extra_latent=-1, action_concat=False):
        super(PointNetFeature, self).__init__()
        self.input_dim = 3 + extra_latent
        self.split_feature = False
        self.pointnet_nclusters, self.pointnet_radius, self.model_scale = pointnet_nclusters, pointnet_radius, model_scale
        input_dim = 3 + policy_extra_latent if policy_extra_latent > 0 else self.input_dim
        self.policy_input_dim = input_dim
        self.encoder = self.create_encoder(model_scale, pointnet_radius, pointnet_nclusters, self.policy_input_dim)
        input_dim = 3 + critic_extra_latent if critic_extra_latent > 0 else input_dim
        self.critic_input_dim = input_dim
        if action_concat:
            self.critic_input_dim = 10
        self.value_encoder = self.create_encoder(model_scale, pointnet_radius, pointnet_nclusters, self.critic_input_dim)
        self._runtime = None

    def create_encoder(self, model_scale, pointnet_radius, pointnet_nclusters, input_dim=0):
        return base_network(pointnet_radius, pointnet_nclusters, model_scale, input_dim)

    def forward(self, pc, grasp=None, concat_option="channel_wise", rotz=True, feature_2=False, train=True):
        """pc (B, C, 1030|1024) CUDA float32 -> (z (B,512), pc); BatchNorm mode follows self.training."""
        from ..runtime import feature_forward
        z = feature_forward(self, pc, value=feature_2)
        return z, pc


class QNetwork(nn.Module):
    """Twin Q + optional 7-D aux (grasp pose) head."""

    def __init__(self, num_inputs, num_actions, hidden_dim, pixel_supervision=False, extra_pred_dim=0):
        super(QNetwork, self).__init__()
        self.linear1 = nn.Linear(num_inputs + num_actions, hidden_dim)
        self.linear2 = nn.Linear(hidden_dim, hidden_dim)
        self.linear3 = nn.Linear(hidden_dim, 1)
        self.extra_pred_dim = extra_pred_dim
        self.linear4 = nn.Linear(num_inputs + num_actions, hidden_dim)
        self.linear5 = nn.Linear(hidden_dim, hidden_dim)
        self.linear6 = nn.Linear(hidden_dim, 1)
        if self.extra_pred_dim > 0:
            self.linear7 = nn.Linear(num_inputs, hidden_dim)
            self.linear8 = nn.Linear(hidden_dim, hidden_dim)
            self.extra_pred = nn.Linear(hidden_dim, self.extra_pred_dim)
        self.apply(weights_init_)
        self.num_actions = num_actions

    def forward(self, state, action=None):
        """value_model form Q(feature) (sa_channel_concat: the action rides in the point cloud's channels): the fused head kernels.
        The classic Q(state, action) form (reference core/networks.py:280-300, no shipped config builds it) is evaluated layer by
        layer with torch's library GEMMs on the module's device: twin trunks on [state | action], the aux trunk on the state
        alone, unit quaternion for a 7-D aux head."""
        if action is None and self.num_actions == 0:
            from ..runtime import critic_forward
            return critic_forward(self, state)
        F = nn.functional
        xu = state if action is None else torch.cat([state, action], 1)
        q1 = self.linear3(F.relu(self.linear2(F.relu(self.linear1(xu)))))
        q2 = self.linear6(F.relu(self.linear5(F.relu(self.linear4(xu)))))
        aux = None
        if self.extra_pred_dim:
            aux = self.extra_pred(F.relu(self.linear8(F.relu(self.linear7(state)))))
            if self.extra_pred_dim == 7:
                aux = torch.cat((F.normalize(aux[:, :4], p=2, dim=-1), aux[:, 4:]), dim=-1)
        return q1, q2, aux


class GaussianPolicy(nn.Module):
    def __init__(self, num_inputs, num_actions, hidden_dim, action_space=None, extra_pred_dim=0, uncertainty=False):
        super(GaussianPolicy, self).__init__()
        self.linear1 = nn.Linear(num_inputs, hidden_dim)
        self.linear2 = nn.Linear(hidden_dim, hidden_dim)
        self.uncertainty = uncertainty
        self.extra_pred_dim = extra_pred_dim
        self.mean = nn.Linear(hidden_dim, num_actions)
        self.extra_pred = nn.Linear(hidden_dim, self.extra_pred_dim)
        self.log_std_linear = nn.Linear(hidden_dim, num_actions)
        self.apply(weights_init_)
        self.action_space = action_space
        if action_space is None:
            self.action_scale = torch.ones(num_actions)
            self.action_bias = torch.zeros(num_actions)
        else:
            self.action_scale = torch.FloatTensor((action_space.high - action_space.low) / 2.0)
            # (asymmetric bounds: pi = tanh(mean) * scale + bias in every pass of the update step, gad_policy_outputs' action_bias;
            # PandaTaskSpace6D is symmetric, tests/golden/ddpg_steps_asym_bounds_B32.npz pins the other case)
            self.action_bias = torch.FloatTensor((action_space.high + action_space.low) / 2.0)

    def sample(self, state, eps=None):
        """-> (squashed mean, log_prob (B,1), action, extra_pred) as reference core/networks.py:353-371:
        x = mean + exp(log_std) * eps with eps ~ N(0,1) (`eps` (B,6) injects the draw of Normal.rsample),
        action = tanh(x) * scale + bias, log_prob with the tanh correction.  Inference-side evaluation through the
        HIP head kernels (no autograd graph: the update step differentiates the heads inside the fused path)."""
        from ..runtime import policy_sample
        r = policy_sample(self, state, eps=eps)
        return r["mean_sq"], r["log_prob"], r["action"], r["extra"]

    def forward(self, state):
        """-> (mean (raw), log_std clamped to [LOG_SIG_MIN, LOG_SIG_MAX], extra_pred with a unit quaternion when
        extra_pred_dim == 7) as reference core/networks.py:339-351."""
        from ..runtime import policy_sample
        r = policy_sample(self, state, draw=False)
        return r["mean"], r["log_std"], r["extra"]

    def to(self, device):
        self.action_scale = self.action_scale.to(device)
        self.action_bias = self.action_bias.to(device)
        return super(GaussianPolicy, self).to(device)
