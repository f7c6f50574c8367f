"""Agent base class with the reference's surface (reference core/agent.py): constructor
(:21-48), setup_feature_extractor (:149-164), get_lr / step_scheduler (:166-190), prepare_data
masks (:211-240), log_stat (:242-259), set_mode (:261-280), save_model / load_model (:282-431).
The arithmetic of update_parameters runs in the fused HIP runtime (ga_ddpg_amd.runtime)."""
import os

import numpy as np
import torch

from .utils import get_loss_info_dict, get_policy_class, hard_update


class PendingLog(dict):
    """Result dict of an update that was enqueued without waiting for it: same keys and floats as the reference's
    (core/utils.py:1008-1020); the first read of any entry waits for the step to finish on the GPU."""

    def __init__(self, resolve):
        dict.__init__(self)
        self._resolve = resolve

    def _fill(self):
        if self._resolve is not None:
            r, self._resolve = self._resolve, None
            dict.update(self, r())
        return self

    def done(self):
        return self._resolve is None


def _lazy(name):
    def f(self, *a, **k):
        return getattr(dict, name)(self._fill(), *a, **k)
    f.__name__ = name
    return f


for _n in ("__getitem__", "__iter__", "__len__", "__contains__", "__repr__", "__eq__", "get", "items", "keys", "values", "copy"):
    setattr(PendingLog, _n, _lazy(_n))


class Agent(object):
    def __init__(self, num_inputs, action_space, args, name):
        for key, val in args.items():
            setattr(self, key, val)
        self.name = name
        self.device = "cuda"
        self.update_step = 1
        self.init_step = 1
        self.extra_pred_dim = 1
        self.critic_extra_pred_dim = 0
        self.action_dim = action_space.shape[0]
        self.has_critic = self.name != "BC"
        if self.policy_aux:
            self.extra_pred_dim = 7
        if self.critic_aux:
            self.critic_extra_pred_dim = 7
        if self.use_time:
            num_inputs += 1
        action_dim = 0 if self.value_model else action_space.shape[0]
        action_space = action_space if self.use_action_limit else None
        self.action_space = action_space
        self.num_inputs = num_inputs
        self.policy, self.policy_optim, self.policy_scheduler, self.policy_target = \
            get_policy_class("GaussianPolicy", self)
        self.action_dim = action_dim
        self.loss_info = list(get_loss_info_dict().keys())
        self.mix_policy_ratio = 0.0
        self._rt = None
        self._dp = None

    # ------------------------------------------------------------------ wiring
    def setup_feature_extractor(self, net_dict, eval=False):
        s, g = net_dict["state_feature_extractor"], net_dict["goal_feature_extractor"]
        self.goal_feature_extractor = g["net"]
        self.goal_feature_extractor_opt = g["opt"]
        self.goal_feature_extractor_sch = g["scheduler"]
        self.state_feature_extractor = s["net"]
        self.state_feature_extractor_optim = s["opt"]
        self.state_feature_extractor_scheduler = s["scheduler"]
        self.state_feat_encoder_optim = s["encoder_opt"]
        self.state_feat_encoder_scheduler = s["encoder_scheduler"]
        self.state_feat_val_encoder_optim = s["val_encoder_opt"]
        self.state_feat_val_encoder_scheduler = s["val_encoder_scheduler"]

    def runtime(self, batch_size, n_cols):
        """the fused runtime for this (batch size, cloud width); built on first use"""
        if self._rt is None or self._rt.B != batch_size or self._rt.NP != n_cols:
            from ..runtime import FusedRuntime
            if self._rt is not None:
                raise RuntimeError("the fused runtime is specialised to one batch shape (B=%d, NP=%d)"
                                   % (self._rt.B, self._rt.NP))
            self._rt = FusedRuntime(self, batch_size, n_cols)
            # a checkpoint loaded BEFORE the first update left its Adam moments / step counts in the torch optimizers only
            # (load_model on a fresh process): the flat optimiser state starts from them, not from zero
            self._optim_states_in()
            if self._dp is not None:
                self._dp.attach(self._rt)
        return self._rt

    def prefetch(self, batch_data):
        """Stage the NEXT update's minibatch while the current one runs (its upload and the furthest-point-sampling / ball-query
        geometry of both cloud sets, on the prefetch lanes): the following update_parameters(batch_data) -- the same object -- starts
        from there.  A hint: results never depend on it; device-resident minibatches only.  -> True if something was staged."""
        if self._rt is None or batch_data is None:
            return False
        return bool(self._rt.prefetch_inputs(batch_data))

    def get_lr(self):
        return {"policy_lr": self.policy_optim.param_groups[0]["lr"],
                "feature_lr": self.state_feature_extractor_optim.param_groups[0]["lr"],
                "value_lr": self.critic_optim.param_groups[0]["lr"] if hasattr(self, "critic") else 0}

    def step_scheduler(self, step=None):
        if hasattr(self, "critic"):
            self.critic_scheduler.step()
        if hasattr(self, "policy"):
            self.policy_scheduler.step()
        if self.train_feature or self.train_value_feature:
            self.state_feature_extractor_scheduler.step()
            self.state_feat_encoder_scheduler.step()
        # NB: the value-encoder scheduler is never stepped in the reference either (agent.py:179-190)

    def set_mode(self, test):
        self.test_mode = test
        nets = [self.state_feature_extractor, self.policy] + ([self.critic] if hasattr(self, "critic") else [])
        for n in nets:
            if n.training == bool(test):             # nn.Module.train() walks the whole module tree: only on a real change
                n.train(not test)

    def update_parameters(self, batch_data, updates, k):
        return {}

    @torch.no_grad()
    def select_action(self, state, actions=None, goal_state=None, vis=False, remain_timestep=0, grasp_set=None,
                      gt_goal_rollout=False, repeat=False, eps=None):
        """Rollout-side inference (reference core/agent.py:82-125): eval-mode BatchNorm (running statistics), batch of
        one, policy.sample on the feature.  Returns the reference's tuple (action = squashed mean (6,), log-prob of the
        sample (scalar), action_sample = tanh(mean + std * eps) * scale (6,), aux pose (7,) or the goal state).
        `eps` (6,) injects the N(0,1) draw of the reparameterised sample (default: drawn on the device)."""
        from ..runtime import feature_forward
        self.state_feature_extractor.eval()
        self.policy.eval()
        pc = torch.as_tensor(np.asarray(state[0][0], dtype=np.float32)[None]).cuda()
        z = feature_forward(self.state_feature_extractor.module, pc, value=False)
        feat = torch.cat((z, torch.full((1, 1), float(remain_timestep), device=z.device)), dim=1)
        if eps is not None:
            eps = np.asarray(eps, dtype=np.float32).reshape(1, 6)
        mean_sq, log_prob, sample, aux = self.policy.sample(feat, eps=eps)
        action = mean_sq[0].cpu().numpy()
        extra_pred = float(log_prob[0, 0])
        action_sample = sample[0].cpu().numpy()
        if self.policy_aux:
            aux_pred = aux[0].cpu().numpy()
        else:
            aux_pred = np.asarray(goal_state, dtype=np.float32).reshape(-1) if goal_state is not None else None
        return action, extra_pred, action_sample, aux_pred

    def _pending_result(self, pend, has_critic):
        """update_parameters(sync=False): the reference's result dict, filled in on first access (which waits for the step)"""
        if not hasattr(pend, "wait"):                 # the runtime synchronised anyway (HIP-graph replay)
            return self._result(pend, has_critic)
        return PendingLog(lambda: self._result(pend.wait(), has_critic))

    def flush(self):
        """wait for every update enqueued with sync=False"""
        if self._rt is not None:
            self._rt.flush()

    def _result(self, s, has_critic):
        """map the runtime's scalar block to the reference's 11-key dict (core/utils.py:1008-1020)"""
        out = {k: 0.0 for k in self.loss_info}
        out["bc_loss"] = float(s[4])
        out["policy_grasp_aux_loss"] = float(s[5])
        out["policy_param"] = float(s[10])
        if has_critic:
            out["critic_loss"] = float(s[0])
            out["critic_grasp_aux_loss"] = float(s[1])
            out["actor_critic_loss"] = float(s[8])
            out["reward_mask_num"] = float(s[3])
            out["critic_grad"] = float(s[11])
            out["critic_param"] = float(s[12])
        else:
            out["reward_mask_num"] = float(s[7])
        for k, v in out.items():
            setattr(self, k, v)
        return out

    # log_stat attributes of the reference that are not part of the returned dict (agent.py:246-250)
    @property
    def policy_grad(self):
        return float(self._rt.pol.flat.grad.abs().max())

    @property
    def feat_grad(self):
        return float(self._rt.enc.flat.grad.abs().max())

    @property
    def feat_param(self):
        return float(self._rt.enc.flat.master.abs().max())

    @property
    def val_feat_grad(self):
        return float(self._rt.venc.flat.grad.abs().max())

    @property
    def val_feat_param(self):
        return float(self._rt.venc.flat.master.abs().max())

    # ------------------------------------------------------------------ checkpoints
    def _paths(self, output_dir, surfix):
        fmt = "{}/{}_{}_{}_{}"
        return {k: fmt.format(output_dir, self.name, k, self.env_name, surfix)
                for k in ("actor", "critic", "goal_feat", "state_feat")}

    def _optim_states_out(self):
        """mirror the flat Adam moments into the torch optimizers so their state_dict() is complete"""
        rt = self._rt
        if rt is None:
            return
        pairs = [(rt.pol.flat, self.policy_optim), (rt.enc.flat, self.state_feat_encoder_optim),
                 (rt.venc.flat, self.state_feat_val_encoder_optim)]
        if self.has_critic:
            pairs.append((rt.cr.flat, self.critic_optim))
        for flat, opt in pairs:
            act = flat.active.cpu().numpy()
            for p, o in zip(flat.params, flat.offsets[:-1]):
                o = int(o)
                if not act[o] or flat.step_count == 0:
                    continue
                opt.state[p] = {"step": torch.tensor(float(flat.step_count)),
                                "exp_avg": flat.exp_avg[o:o + p.numel()].view(p.shape),
                                "exp_avg_sq": flat.exp_avg_sq[o:o + p.numel()].view(p.shape)}

    def _optim_states_in(self):
        rt = self._rt
        if rt is None:
            return
        pairs = [(rt.pol.flat, self.policy_optim), (rt.enc.flat, self.state_feat_encoder_optim),
                 (rt.venc.flat, self.state_feat_val_encoder_optim)]
        if self.has_critic:
            pairs.append((rt.cr.flat, self.critic_optim))
        for flat, opt in pairs:
            for p, o in zip(flat.params, flat.offsets[:-1]):
                st = opt.state.get(p)
                if st:
                    o = int(o)
                    flat.exp_avg[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                    flat.exp_avg_sq[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                    flat.step_count = int(st["step"])
            flat.sync_packed()

    def save_model(self, step, output_dir="", surfix="latest", actor_path=None, critic_path=None,
                   goal_feat_path=None, state_feat_path=None):
        self.flush()                        # run-ahead updates (sync=False) still in flight finish before anything is read
        os.makedirs(output_dir, exist_ok=True)
        paths = self._paths(output_dir, surfix)
        actor_path = actor_path or paths["actor"]
        critic_path = critic_path or paths["critic"]
        state_feat_path = state_feat_path or paths["state_feat"]
        self._optim_states_out()
        torch.save({"net": self.policy.state_dict(), "opt": self.policy_optim.state_dict(),
                    "sch": self.policy_scheduler.state_dict()}, actor_path)
        if hasattr(self, "critic"):
            torch.save({"net": self.critic.state_dict(), "opt": self.critic_optim.state_dict(),
                        "sch": self.critic_scheduler.state_dict()}, critic_path)
        if self.use_point_state:
            torch.save({"net": self.state_feature_extractor.state_dict(),
                        "opt": self.state_feature_extractor_optim.state_dict(),
                        "encoder_opt": self.state_feat_encoder_optim.state_dict(),
                        "sch": self.state_feature_extractor_scheduler.state_dict(),
                        "encoder_sch": self.state_feat_encoder_scheduler.state_dict(),
                        "val_encoder_opt": self.state_feat_val_encoder_optim.state_dict(),
                        "val_encoder_sch": self.state_feat_val_encoder_scheduler.state_dict(),
                        "step": step}, state_feat_path)

    def _reinit_lr(self, optim, milestones):
        """fine-tuning restart of an optimiser's learning-rate schedule (reference core/agent.py:375-383): lr back to
        `reinit_lr`, a fresh MultiStepLR over the configured milestones"""
        for g in optim.param_groups:
            g["lr"] = self.reinit_lr
        sch = torch.optim.lr_scheduler.MultiStepLR(optim, milestones=list(milestones), gamma=0.5)
        sch.initial_lr = self.reinit_lr
        sch.base_lrs[0] = self.reinit_lr
        return sch

    def load_model(self, output_dir, surfix="latest", set_init_step=False, reinit_value_feat=False):
        """reference core/agent.py:348-431 (`reinit_value_feat` is accepted and unused there as well)"""
        paths = self._paths(output_dir, surfix)
        if os.path.exists(paths["actor"]):
            d = torch.load(paths["actor"], weights_only=False)
            self.policy.load_state_dict(d["net"])
            self.policy_optim.load_state_dict(d["opt"])
            self.policy_scheduler.load_state_dict(d["sch"])
            if self.reinit_optim and set_init_step:                  # reference core/agent.py:375-383
                self.policy_scheduler = self._reinit_lr(self.policy_optim, self.policy_milestones)
            hard_update(self.policy_target, self.policy, self.tau)
        if hasattr(self, "critic") and os.path.exists(paths["critic"]):
            d = torch.load(paths["critic"], weights_only=False)
            self.critic.load_state_dict(d["net"])
            self.critic_optim.load_state_dict(d["opt"])
            self.critic_scheduler.load_state_dict(d["sch"])
            if self.reinit_optim and set_init_step:                  # reference core/agent.py:394-401
                self.critic_scheduler = self._reinit_lr(self.critic_optim, self.value_milestones)
            hard_update(self.critic_target, self.critic, self.tau)
        step = 0
        if os.path.exists(paths["state_feat"]):
            d = torch.load(paths["state_feat"], weights_only=False)
            self.state_feature_extractor.load_state_dict(d["net"])
            try:
                self.state_feature_extractor_optim.load_state_dict(d["opt"])
                self.state_feature_extractor_scheduler.load_state_dict(d["sch"])
                self.state_feat_encoder_optim.load_state_dict(d["encoder_opt"])
                self.state_feat_encoder_scheduler.load_state_dict(d["encoder_sch"])
                self.state_feat_val_encoder_optim.load_state_dict(d["val_encoder_opt"])
                self.state_feat_val_encoder_scheduler.load_state_dict(d["val_encoder_sch"])
            except Exception:
                print("loading feature optim has mismatches")
            self.update_step = d["step"]
            if set_init_step:
                self.init_step = self.update_step
            step = self.update_step
        self._optim_states_in()
        return step
