"""CPU oracle of the GA-DDPG update step (pure PyTorch, float32).  TEST INFRASTRUCTURE ONLY.

A device-agnostic restatement of the reference's hot path, used (a) as the checker for the HIP
path on arbitrary seeded inputs and (b) as the timed ``cpu_baseline`` ("port") in bench.py.
Pinned against tests/golden/{ddpg_steps,bc_steps,encoder,heads,losses}*.npz, which were produced
by the reference's own Python (oracle/make_golden.py).  Each block cites the reference lines it
follows (paths relative to /root/reference).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.optim import Adam
from torch.optim.lr_scheduler import MultiStepLR

from oracle.pointnet2_ops import pointnet2_modules as pn2

LOSS_KEYS = ("bc_loss", "policy_grasp_aux_loss", "critic_grasp_aux_loss", "critic_loss",
             "actor_critic_loss", "reward_mask_num", "expert_mask_num", "policy_param",
             "critic_grad", "critic_param", "train_batch_size")          # core/utils.py:1008-1020

ACTION_HIGH = np.array([0.06, 0.06, 0.06, np.pi / 6, np.pi / 6, np.pi / 6])   # core/utils.py:505-510


# ----------------------------------------------------------------------------- networks
def make_encoder(in_features, radius=0.02, nclusters=32, scale=1):
    """core/networks.py:65-92 (base_network): 3 SA modules + Linear/BN1d/ReLU x2."""
    sa = nn.ModuleList([
        pn2.PointnetSAModule(npoint=nclusters, radius=radius, nsample=64,
                             mlp=[in_features, 64 * scale, 64 * scale, 128 * scale]),
        pn2.PointnetSAModule(npoint=32, radius=0.04, nsample=128,
                             mlp=[128 * scale, 128 * scale, 128 * scale, 256 * scale]),
        pn2.PointnetSAModule(mlp=[256 * scale, 256 * scale, 256 * scale, 512 * scale]),
    ])
    fc = nn.Sequential(nn.Linear(512 * scale, 1024 * scale), nn.BatchNorm1d(1024 * scale), nn.ReLU(True),
                       nn.Linear(1024 * scale, 512 * scale), nn.BatchNorm1d(512 * scale), nn.ReLU(True))
    return nn.ModuleList([sa, fc])


class PointFeature(nn.Module):
    """core/networks.py:182-250 (PointNetFeature): policy `encoder` (C=4) + critic `value_encoder`."""

    def __init__(self, extra_latent=1, action_concat=True):
        super().__init__()
        self.policy_input_dim = 3 + extra_latent
        self.critic_input_dim = 10 if action_concat else self.policy_input_dim
        self.encoder = make_encoder(self.policy_input_dim)
        self.value_encoder = make_encoder(self.critic_input_dim)

    def forward(self, pc, value=False):
        x = pc[..., 6:] if pc.shape[-1] != 1024 else pc               # drop the 6 gripper points (:234-235)
        c = self.critic_input_dim if value else self.policy_input_dim
        feats = x[:, :c].contiguous()
        xyz = feats.transpose(1, -1)[..., :3].contiguous()
        enc = self.value_encoder if value else self.encoder
        for sa in enc[0]:
            xyz, feats = sa(xyz, feats)
        return enc[1](feats.squeeze(-1))


class _DataParallelShell(nn.Module):
    """Only reproduces nn.DataParallel's `module.` state-dict prefix (core/utils.py:202)."""

    def __init__(self, net):
        super().__init__()
        self.module = net

    def forward(self, *a, **k):
        return self.module(*a, **k)


def _xavier(m):
    if isinstance(m, nn.Linear):                                      # core/networks.py:100-103
        nn.init.xavier_uniform_(m.weight, gain=1)
        nn.init.constant_(m.bias, 0)


def _unit_quat_head(x):
    return torch.cat((F.normalize(x[:, :4], p=2, dim=-1), x[:, 4:]), dim=-1)


class QNet(nn.Module):
    """core/networks.py:253-300 (QNetwork, value_model: no action input)."""

    def __init__(self, num_inputs=513, hidden=256, extra_pred_dim=7):
        super().__init__()
        self.linear1, self.linear2, self.linear3 = nn.Linear(num_inputs, hidden), nn.Linear(hidden, hidden), nn.Linear(hidden, 1)
        self.linear4, self.linear5, self.linear6 = nn.Linear(num_inputs, hidden), nn.Linear(hidden, hidden), nn.Linear(hidden, 1)
        self.extra_pred_dim = extra_pred_dim
        if extra_pred_dim > 0:
            self.linear7, self.linear8 = nn.Linear(num_inputs, hidden), nn.Linear(hidden, hidden)
            self.extra_pred = nn.Linear(hidden, extra_pred_dim)
        self.apply(_xavier)

    def forward(self, s):
        q1 = self.linear3(F.relu(self.linear2(F.relu(self.linear1(s)))))
        q2 = self.linear6(F.relu(self.linear5(F.relu(self.linear4(s)))))
        aux = None
        if self.extra_pred_dim:
            aux = self.extra_pred(F.relu(self.linear8(F.relu(self.linear7(s)))))
            if self.extra_pred_dim == 7:
                aux = _unit_quat_head(aux)
        return q1, q2, aux


class PolicyNet(nn.Module):
    """core/networks.py:303-377 (GaussianPolicy).  Only the squashed mean and the aux head reach a
    loss; the rsample/log_prob branch (:355-368) feeds nothing on the update path."""

    def __init__(self, num_inputs=513, num_actions=6, hidden=256, extra_pred_dim=7):
        super().__init__()
        self.linear1, self.linear2 = nn.Linear(num_inputs, hidden), nn.Linear(hidden, hidden)
        self.mean = nn.Linear(hidden, num_actions)
        self.extra_pred = nn.Linear(hidden, extra_pred_dim)
        self.log_std_linear = nn.Linear(hidden, num_actions)
        self.extra_pred_dim = extra_pred_dim
        self.apply(_xavier)
        self.register_buffer("action_scale", torch.tensor(ACTION_HIGH, dtype=torch.float32), persistent=False)
        self.register_buffer("action_bias", torch.zeros(num_actions, dtype=torch.float32), persistent=False)
        # (buffers, so .double() converts them together with the weights)

    def set_action_space(self, space):
        """core/networks.py:329-337: scale = (high - low) / 2, bias = (high + low) / 2"""
        self.action_scale.copy_(torch.as_tensor((space.high - space.low) / 2.0, dtype=self.action_scale.dtype))
        self.action_bias.copy_(torch.as_tensor((space.high + space.low) / 2.0, dtype=self.action_bias.dtype))

    def forward(self, s):
        h = F.relu(self.linear2(F.relu(self.linear1(s))))
        aux = self.extra_pred(h)
        if self.extra_pred_dim == 7:
            aux = _unit_quat_head(aux)
        pi = torch.tanh(self.mean(h)) * self.action_scale + self.action_bias
        return pi, aux

    def sample(self, s, eps):
        """core/networks.py:339-371 (forward + sample) with the N(0,1) draw of Normal.rsample injected as `eps`:
        -> (squashed mean, log_prob (B,1), action, extra_pred, raw mean, clamped log_std)"""
        h = F.relu(self.linear2(F.relu(self.linear1(s))))
        mean = self.mean(h)
        extra = self.extra_pred(h)
        if self.extra_pred_dim == 7:
            extra = _unit_quat_head(extra)
        log_std = torch.clamp(self.log_std_linear(h), min=-10, max=2)          # LOG_SIG_MIN / LOG_SIG_MAX (:20-21)
        std = log_std.exp()
        x_t = mean + std * eps
        y_t = torch.tanh(x_t)
        action = y_t * self.action_scale + self.action_bias
        log_prob = -((x_t - mean) ** 2) / (2 * std ** 2) - log_std - np.log(np.sqrt(2 * np.pi))
        log_prob = log_prob - torch.log(self.action_scale * (1 - y_t.pow(2)) + 1e-6)
        return (torch.tanh(mean) * self.action_scale + self.action_bias, log_prob.sum(1, keepdim=True), action, extra, mean, log_std)


# ----------------------------------------------------------------------------- losses / pose math
_CP = np.array([[0, 0, 0], [0, 0, 0], [0.053, -0., 0.075], [-0.053, 0., 0.075],
                [0.053, -0., 0.105], [-0.053, 0., 0.105]], dtype=np.float32)   # core/utils.py:819-824


def control_points(rotz, device, dtype=torch.float32):
    cp = _CP
    if rotz:                                                        # core/utils.py:826-827, rotZ(pi/2)
        c, s = np.cos(np.pi / 2), np.sin(np.pi / 2)
        cp = cp @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    return torch.tensor(cp, dtype=torch.float32, device=device).to(dtype)


def quat_rotate(q, v):
    """core/utils.py:940-958 (qrot): v + 2*(w*(u x v) + u x (u x v))."""
    u = q[..., 1:]
    uv = torch.cross(u, v, dim=-1)
    uuv = torch.cross(u, uv, dim=-1)
    return v + 2 * (q[..., :1] * uv + uuv)


def goal_pred_loss(pred, gt):
    """core/loss.py:17-23: mean over (rows, 6 pts) of sum_xyz |P(pred) - P(gt)|, P = q-rotate + t."""
    cp = control_points(True, pred.device, pred.dtype)[None]          # (1,6,3)

    def pts(g):
        return quat_rotate(g[:, None, :4].expand(-1, 6, -1), cp.expand(g.shape[0], -1, -1)) + g[:, None, 4:]
    return torch.abs(pts(pred) - pts(gt)).sum(-1).mean()


def euler_matrix(az, el, th):
    """core/utils.py:890-910 (tc_rotation_matrix batched): Rz(th) @ Ry(el) @ Rx(az)."""
    cx, cy, cz, sx, sy, sz = az.cos(), el.cos(), th.cos(), az.sin(), el.sin(), th.sin()
    o, z = torch.ones_like(cx), torch.zeros_like(cx)
    rx = torch.stack([o, z, z, z, cx, -sx, z, sx, cx], -1).view(-1, 3, 3)
    ry = torch.stack([cy, z, sy, z, o, z, -sy, z, cy], -1).view(-1, 3, 3)
    rz = torch.stack([cz, -sz, z, sz, cz, z, z, z, o], -1).view(-1, 3, 3)
    return rz @ (ry @ rx)


def pose_bc_loss(pi, act):
    """core/loss.py:25-31: control points moved by euler rotation pi[3:] and translation pi[:3]."""
    cp = control_points(False, pi.device, pi.dtype)[None]

    def pts(a):
        R = euler_matrix(a[:, 3], a[:, 4], a[:, 5])
        return cp.expand(a.shape[0], -1, -1) @ R.transpose(1, 2) + a[:, None, :3]
    return torch.abs(pts(pi) - pts(act)).sum(-1).mean()


def target_noise(u, level):
    """core/utils.py:568-576 'uniform' branch on tensors: (u*3 - 6)*level, rotation part x5."""
    d = (u * 3 - 6) * level
    d[:, 3:] *= 5
    return d


# ----------------------------------------------------------------------------- agent
def _max_abs(tensors):
    vals = [float(t.detach().abs().max()) for t in tensors if t is not None]
    return max(vals) if vals else 0.0


class OracleAgent(object):
    """DDPG / BC update step of the reference (core/agent.py, core/ddpg.py, core/bc.py)."""

    def __init__(self, train_cfg, spec=None, kind=None, device="cpu", dtype=torch.float32):
        """dtype=torch.float64 gives a higher-precision yardstick (tests/test_gpu_step.py uses it to
        judge float32 gradient error; call .double() on the nets after filling them)."""
        c = self.c = train_cfg
        self.dtype = dtype
        self.kind = kind or ("DDPG" if c.RL else "BC")
        self.has_critic = self.kind != "BC"
        self.device = torch.device(device)
        self.update_step = 1                                         # core/agent.py:28
        pol_aux = 7 if c.policy_aux else 1
        self.policy = PolicyNet(513, 6, c.hidden_size, pol_aux).to(device)
        self.policy_target = PolicyNet(513, 6, c.hidden_size, pol_aux).to(device)
        self.policy_optim = Adam(self.policy.parameters(), lr=c.lr, eps=1e-5, weight_decay=1e-5)
        self.policy_scheduler = MultiStepLR(self.policy_optim, milestones=list(c.policy_milestones), gamma=c.lr_gamma)
        if self.has_critic:                                          # core/utils.py:984-1006
            cr_aux = 7 if c.critic_aux else 0
            self.critic = QNet(513, c.hidden_size, cr_aux).to(device)
            self.critic_target = QNet(513, c.hidden_size, cr_aux).to(device)
            self.critic_optim = Adam(self.critic.parameters(), lr=c.value_lr, eps=1e-5, weight_decay=1e-5)
            self.critic_scheduler = MultiStepLR(self.critic_optim, milestones=list(c.value_milestones), gamma=c.value_lr_gamma)
        spec = spec or {"opt_kwargs": {"lr": 1e-3}, "scheduler_kwargs":
                        {"milestones": [8000, 16000, 30000, 50000, 70000, 90000], "gamma": 0.3}}
        self.state_feature_extractor = _DataParallelShell(
            PointFeature(extra_latent=1, action_concat=bool(c.sa_channel_concat))).to(device)
        net = self.state_feature_extractor.module                    # core/utils.py:183-237
        self.encoder_optim = Adam(net.encoder.parameters(), **spec["opt_kwargs"])
        self.val_encoder_optim = Adam(net.value_encoder.parameters(), **spec["opt_kwargs"])
        self.encoder_scheduler = MultiStepLR(self.encoder_optim, **spec["scheduler_kwargs"])
        # the value-encoder scheduler exists but is never stepped (core/agent.py:179-190)

    # -- features ------------------------------------------------------------------
    def features(self, pc, time, action=None):
        """core/ddpg.py:36-59 / core/bc.py:58-69 + core/utils.py:291-297."""
        if action is not None:
            pc = torch.cat((pc, action.unsqueeze(2).expand(-1, -1, pc.shape[2])), 1)
        z = self.state_feature_extractor(pc, value=action is not None)
        return torch.cat((z, time[:, None]), dim=1)

    def mix_policy_ratio(self):
        c = self.c                                                    # core/ddpg.py:108-117
        idx = int((self.update_step > np.array(c.mix_milestones)).sum())
        r = c.mix_policy_ratio_list[min(len(c.mix_policy_ratio_list) - 1, idx)]
        return min(r, c.ddpg_coefficients[4])

    def _load(self, batch):
        t = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32, device=self.device).to(self.dtype)
             for k, v in batch.items() if k not in ("grasp_sample_batch",)}
        m = {}
        m["reward"] = (t["return_batch"] > 0).view(-1)               # core/agent.py:224-229
        m["expert"] = (t["expert_flag_batch"] >= 1).view(-1)
        m["expert_reward"] = m["reward"] & m["expert"]
        m["keep"] = t["perturb_flag_batch"] < 1
        m["goal_reward"] = m["reward"]
        return t, m

    # -- DDPG ------------------------------------------------------------------------
    def update_ddpg(self, batch, noise_u=None, test=False):
        """test=True (core/agent.py:261-280 set_mode): the same update with the online networks in eval mode -- BatchNorm on its
        running statistics, forward and backward"""
        c = self.c
        ratio = self.mix_policy_ratio()
        for n in (self.state_feature_extractor, self.policy, self.critic):
            n.train(not test)
        t, m = self._load(batch)
        out = OrderedDict((k, 0.0) for k in LOSS_KEYS)
        pc, nxt, time = t["point_state_batch"], t["next_point_state_batch"], t["time_batch"]

        # critic phase (core/ddpg.py:154-161, 119-143)
        self.critic_optim.zero_grad()
        self.val_encoder_optim.zero_grad()
        value_feat = self.features(pc, time, t["action_batch"])
        with torch.no_grad():                                         # target_value, core/ddpg.py:61-88
            ns = self.features(nxt, time - 1)
            a_next, _ = self.policy_target(ns)
            idx = int((self.update_step > np.array(c.mix_milestones)).sum())
            level = c.action_noise * c.noise_ratio_list[min(len(c.noise_ratio_list) - 1, idx)]
            if noise_u is None:
                noise_u = torch.rand_like(a_next)
            d = target_noise(torch.as_tensor(noise_u, dtype=torch.float32, device=self.device).to(self.dtype).clone(), level)
            d[:, :3] = torch.clamp(d[:, :3], -0.01, 0.01)
            a_next = a_next + d
            nt = self.features(nxt, time - 1, a_next)
            q1t, q2t, _ = self.critic_target(nt)
            y = t["reward_batch"] + (1 - t["mask_batch"]) * c.gamma * torch.min(q1t, q2t).squeeze()
        q1, q2, aux = self.critic(value_feat)
        q1, q2 = q1.squeeze(), q2.squeeze()
        critic_loss = F.smooth_l1_loss(q1[m["keep"]], y[m["keep"]]) + F.smooth_l1_loss(q2[m["keep"]], y[m["keep"]])
        critic_aux_loss = torch.zeros((), device=self.device)
        if c.critic_aux:
            critic_aux_loss = goal_pred_loss(aux[m["goal_reward"], :7], t["goal_batch"][m["goal_reward"]])
        (critic_aux_loss + critic_loss).backward()
        torch.nn.utils.clip_grad_norm_(self.critic.parameters(), c.clip_grad)
        self.val_encoder_optim.step()
        self.critic_optim.step()
        self.dbg = dict(value_feat=value_feat.detach(), next_state=ns, next_target=nt, q1=q1.detach(),
                        q2=q2.detach(), y=y, critic_aux=aux.detach() if aux is not None else None)

        # actor phase (core/ddpg.py:164-180, core/agent.py:127-139, 192-209)
        policy_feat = self.features(pc, time)
        pi, aux_pred = self.policy(policy_feat)
        actor_critic_loss = torch.zeros((), device=self.device)
        if self.update_step % c.policy_update_gap == 0:
            vf = self.features(pc, time, pi)
            q1p, q2p, _ = self.critic(vf)
            keep = ~m["expert_reward"]
            actor_critic_loss = -ratio * torch.min(q1p.squeeze()[keep], q2p.squeeze()[keep]).mean()
            self.dbg.update(value_pi=vf.detach(), q1_pi=q1p.squeeze()[keep].detach(), q2_pi=q2p.squeeze()[keep].detach())
        pol_aux_loss = torch.zeros((), device=self.device)
        if c.policy_aux:
            pol_aux_loss = goal_pred_loss(aux_pred[m["goal_reward"], :7], t["goal_batch"][m["goal_reward"], :7])
        bc = pose_bc_loss(pi[m["expert"]], t["expert_action_batch"][m["expert"]]) * (1 - ratio)
        self.encoder_optim.zero_grad()
        self.policy_optim.zero_grad()
        (pol_aux_loss + bc + actor_critic_loss).backward()
        self.policy_optim.step()
        if c.train_feature:
            self.encoder_optim.step()
        self._target_updates()
        self.dbg.update(policy_feat=policy_feat.detach(), pi=pi.detach(), aux_pred=aux_pred.detach())
        self.update_step += 1

        out.update(bc_loss=float(bc.detach()), policy_grasp_aux_loss=float(pol_aux_loss.detach()),
                   critic_grasp_aux_loss=float(critic_aux_loss), critic_loss=float(critic_loss),
                   actor_critic_loss=float(actor_critic_loss), reward_mask_num=float(m["reward"].sum()),
                   policy_param=_max_abs(self.policy.parameters()),
                   critic_grad=_max_abs(p.grad for p in self.critic.parameters()),
                   critic_param=_max_abs(self.critic.parameters()))
        return out

    def _target_updates(self):
        tau = self.c.tau                                              # core/utils.py:750-774
        with torch.no_grad():
            for tp, p in zip(self.policy_target.parameters(), self.policy.parameters()):
                tp.copy_(tp * (1.0 - tau) + p * tau)
            if self.has_critic:
                src = dict(self.critic.named_parameters())
                for name, tp in self.critic_target.named_parameters():
                    if name[:7] in ("linear1", "linear2", "linear3"):
                        tp.copy_(tp * (1.0 - tau) + src[name] * tau)
                    elif name[:7] in ("linear4", "linear5", "linear6") and \
                            self.update_step % self.c.target_update_interval == 0:
                        tp.copy_(src[name])

    # -- BC ------------------------------------------------------------------------------
    def update_bc(self, batch):
        c = self.c                                                    # core/bc.py:71-87
        self.state_feature_extractor.train()
        self.policy.train()
        t, m = self._load(batch)
        out = OrderedDict((k, 0.0) for k in LOSS_KEYS)
        feat = self.features(t["point_state_batch"], t["time_batch"])
        pi, aux_pred = self.policy(feat)
        pol_aux_loss = torch.zeros((), device=self.device)
        if c.policy_aux:
            pol_aux_loss = goal_pred_loss(aux_pred[m["goal_reward"], :7], t["goal_batch"][m["goal_reward"], :7])
        bc = pose_bc_loss(pi[m["expert"]], t["expert_action_batch"][m["expert"]])
        self.encoder_optim.zero_grad()
        self.policy_optim.zero_grad()
        (pol_aux_loss + bc).backward()
        self.policy_optim.step()
        if c.train_feature:
            self.encoder_optim.step()
        self._target_updates()
        self.dbg = dict(policy_feat=feat.detach(), pi=pi.detach(), aux_pred=aux_pred.detach())
        self.update_step += 1
        out.update(bc_loss=float(bc.detach()), policy_grasp_aux_loss=float(pol_aux_loss.detach()),
                   reward_mask_num=float(m["reward"].float().sum()),
                   policy_param=_max_abs(self.policy.parameters()))
        return out

    def update_parameters(self, batch, noise_u=None, test=False):
        if test and not self.has_critic:
            raise NotImplementedError("oracle: test=True is restated for the DDPG step only")
        return self.update_ddpg(batch, noise_u, test) if self.has_critic else self.update_bc(batch)

    def step_scheduler(self):
        if self.has_critic:                                           # core/agent.py:179-190
            self.critic_scheduler.step()
        self.policy_scheduler.step()
        self.encoder_scheduler.step()

    def to_dtype(self, dtype):
        """convert every network (after a deterministic float32 fill) to `dtype`"""
        self.dtype = dtype
        for n in self.nets().values():
            n.to(dtype)
        return self

    def nets(self):
        d = {"policy": self.policy, "policy_target": self.policy_target,
             "state_feature_extractor": self.state_feature_extractor}
        if self.has_critic:
            d.update(critic=self.critic, critic_target=self.critic_target)
        return d
