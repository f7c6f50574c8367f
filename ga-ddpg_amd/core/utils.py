"""Factories, target-update helpers and small utilities of the update-step path with the
reference's names (reference core/utils.py: make_nets_opts_schedulers :183-237, get_valid_index
:240-241, concat_state_action_channelwise :291-297, PandaTaskSpace6D :505-510, get_noise_delta
:568-584, soft/half-soft/half-hard/hard updates :750-774, get_policy_class :960-981, get_critic
:984-1006, get_loss_info_dict :1008-1020, module_max_param/gradient :92-108).
Everything env/vision related in the reference's utils.py is out of scope."""
from collections import deque

import numpy as np
import torch
import yaml
from torch import optim
from torch.optim import Adam

from ..synth_data import HAND_FINGER_POINT as hand_finger_point  # noqa: F401
from ..synth_data import PandaTaskSpace6D  # noqa: F401


class DataParallel(torch.nn.Module):
    """Stands in for torch.nn.DataParallel (reference utils.py:202): same `.module` attribute and
    `module.` state-dict prefix.  Multi-GPU is one process per GPU + RCCL all-reduce
    (ga_ddpg_amd.parallel), not single-process replicas."""

    def __init__(self, module):
        super(DataParallel, self).__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def make_nets_opts_schedulers(model_spec, config, cuda_device="cuda"):
    from . import networks
    specs = yaml.load(open(model_spec).read(), Loader=yaml.SafeLoader)
    ret = {}
    for net_name, spec in specs.items():
        net_args = dict(spec.get("net_kwargs", {}))
        net_args["input_dim"] = config.channel_num
        if net_name == "state_feature_extractor":
            if hasattr(config, "policy_extra_latent"):
                net_args["policy_extra_latent"] = config.policy_extra_latent
                net_args["critic_extra_latent"] = config.critic_extra_latent
            if config.sa_channel_concat:
                net_args["action_concat"] = True
        net = DataParallel(getattr(networks, spec["class"])(**net_args))
        if cuda_device is not None and torch.cuda.is_available():       # reference :202-204 (before the optimisers exist)
            net = net.to(cuda_device)
        d = {"net": net}
        if "opt" in spec:
            d["opt"] = getattr(optim, spec["opt"])(net.parameters(), **spec["opt_kwargs"])
            sk = dict(spec["scheduler_kwargs"])
            if len(config.overwrite_feat_milestone) > 0:
                sk["milestones"] = config.overwrite_feat_milestone
            d["scheduler"] = getattr(optim.lr_scheduler, spec["scheduler"])(d["opt"], **sk)
            if hasattr(net.module, "encoder"):
                d["encoder_opt"] = getattr(optim, spec["opt"])(net.module.encoder.parameters(), **spec["opt_kwargs"])
                d["encoder_scheduler"] = getattr(optim.lr_scheduler, spec["scheduler"])(d["encoder_opt"], **sk)
            if hasattr(net.module, "value_encoder"):
                d["val_encoder_opt"] = getattr(optim, spec["opt"])(net.module.value_encoder.parameters(),
                                                                   **spec["opt_kwargs"])
                d["val_encoder_scheduler"] = getattr(optim.lr_scheduler, spec["scheduler"])(d["val_encoder_opt"], **sk)
        ret[net_name] = d
    return ret


def get_valid_index(arr, index):
    return arr[min(len(arr) - 1, index)]


def get_policy_class(policy_net_name, args):
    from . import networks
    cls = getattr(networks, policy_net_name)
    policy = cls(args.num_inputs, args.action_dim, args.hidden_size, args.action_space,
                 extra_pred_dim=args.extra_pred_dim).to("cuda")          # reference :961-968: on the device from the start
    policy_optim = Adam(policy.parameters(), lr=args.lr, eps=1e-5, weight_decay=1e-5)
    policy_scheduler = torch.optim.lr_scheduler.MultiStepLR(policy_optim, milestones=list(args.policy_milestones),
                                                            gamma=args.lr_gamma)
    policy_target = cls(args.num_inputs, args.action_dim, args.hidden_size, args.action_space,
                        extra_pred_dim=args.extra_pred_dim).to("cuda")
    return policy, policy_optim, policy_scheduler, policy_target


def get_critic(args):
    from . import networks
    model = networks.QNetwork
    critic = model(args.critic_num_input, args.critic_value_dim, args.hidden_size,
                   extra_pred_dim=args.critic_extra_pred_dim).to("cuda")
    critic_optim = Adam(critic.parameters(), lr=args.value_lr, eps=1e-5, weight_decay=1e-5)
    critic_scheduler = torch.optim.lr_scheduler.MultiStepLR(critic_optim, milestones=list(args.value_milestones),
                                                            gamma=args.value_lr_gamma)
    critic_target = model(args.critic_num_input, args.critic_value_dim, args.hidden_size,
                          extra_pred_dim=args.critic_extra_pred_dim).to("cuda")
    return critic, critic_optim, critic_scheduler, critic_target


def get_loss_info_dict():
    return {"bc_loss": deque([0], maxlen=50), "policy_grasp_aux_loss": deque([0], maxlen=50),
            "critic_grasp_aux_loss": deque([0], maxlen=100), "critic_loss": deque([0], maxlen=100),
            "actor_critic_loss": deque([0], maxlen=50), "reward_mask_num": deque([0], maxlen=5),
            "expert_mask_num": deque([0], maxlen=5), "policy_param": deque([0], maxlen=5),
            "critic_grad": deque([0], maxlen=5), "critic_param": deque([0], maxlen=5),
            "train_batch_size": deque([0], maxlen=5)}


def _polyak(target, source, tau, select):
    """target/source nn.Modules whose parameters live on the GPU; runs gad_polyak per tensor.
    (The fused step uses one flat launch instead: runtime.FusedRuntime._target_updates.)"""
    from .. import hip
    for (tn, tp), (_, sp) in zip(target.named_parameters(), source.named_parameters()):
        k = select(tn)
        if k:
            hip.require_cuda(tp.data, sp.data)
            hip.call("gad_polyak", tp.data, sp.data, None, None, None, tp.numel(), float(tau) if k == 1 else 1.0, 0)
    from ..runtime import sync_module
    sync_module(target)


def soft_update(target, source, tau):
    _polyak(target, source, tau, lambda n: 1)


def half_soft_update(target, source, tau):
    _polyak(target, source, tau, lambda n: 1 if n[:7] in ("linear1", "linear2", "linear3") else 0)


def half_hard_update(target, source, tau):
    _polyak(target, source, tau, lambda n: 2 if n[:7] in ("linear4", "linear5", "linear6") else 0)


def hard_update(target, source, tau=None):
    _polyak(target, source, tau, lambda n: 2)


def module_max_param(module):
    vals = [float(p.data.abs().max()) for _, p in module.named_parameters()]
    return max(vals) if vals else 0.0


def module_max_gradient(module):
    vals = [float(p.grad.abs().max()) for _, p in module.named_parameters() if p.grad is not None]
    return max(vals) if vals else 0.0


def get_noise_delta(action, noise_level, noise_type="uniform"):
    """numpy branch of the reference helper (rollout-side exploration noise)."""
    if noise_type != "uniform":
        d = np.random.normal(size=(6,)) * noise_level / 2.0
    else:
        d = np.random.uniform(-3, 3, size=(6,)) * noise_level
    d[3:] *= 5
    return d


def migrate_model(in_model, out_model, surfix="latest", grasp_model=None):
    """Copy a pretrained checkpoint set into a new run directory under the DDPG file names (reference
    core/utils.py:319-334): the BC_* files of `in_model` when they exist, else its DDPG_* files; missing files are
    skipped.  Returns the list of (source, destination) pairs copied."""
    import os
    import shutil
    in_policy_name, out_policy_name = "BC", "DDPG"
    copied = []
    for name in ("actor", "state_feat", "goal_feat", "critic"):
        fname = "{}_PandaYCBEnv_{}".format(name, surfix)
        if not os.path.exists("{}/{}_{}".format(in_model, in_policy_name, fname)):
            in_policy_name = "DDPG"              # sticky, as in the reference: later files are looked up as DDPG_*
        src = "{}/{}_{}".format(in_model, in_policy_name, fname)
        dst = "{}/{}_{}".format(out_model, out_policy_name, fname)
        if os.path.exists(src):
            os.makedirs(out_model, exist_ok=True)
            shutil.copyfile(src, dst)
            copied.append((src, dst))
    return copied


# ---------------------------------------------------------------------------------------------------------------------
# pose helpers of the replay buffer's hindsight goal relabelling (reference core/utils.py:299-306, 446-452, 672-676)
# ---------------------------------------------------------------------------------------------------------------------
def se3_inverse(RT):
    """inverse of a rigid 4x4 transform (reference core/utils.py:446-452)"""
    import numpy as np
    R, T = RT[:3, :3], RT[:3, 3].reshape((3, 1))
    out = np.eye(4, dtype=np.float32)
    out[:3, :3] = R.transpose()
    out[:3, 3] = -1 * np.dot(R.transpose(), T).reshape(3)
    return out


def mat2quat(M):
    """rotation matrix -> quaternion (w, x, y, z), w >= 0.  The reference imports this from transforms3d (third-party, not
    vendored under /root/reference and absent from this image: PARITY UNPINNED); restated from the library's published
    algorithm (Bar-Itzhack 2000: the eigenvector of the symmetric 4x4 K matrix with the largest eigenvalue)."""
    import numpy as np
    Qxx, Qyx, Qzx, Qxy, Qyy, Qzy, Qxz, Qyz, Qzz = np.asarray(M, dtype=np.float64).flat
    K = np.array([[Qxx - Qyy - Qzz, 0, 0, 0],
                  [Qyx + Qxy, Qyy - Qxx - Qzz, 0, 0],
                  [Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, 0],
                  [Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)                       # (uses the lower triangle)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


def safemat2quat(mat):
    """reference core/utils.py:299-306: identity quaternion when the conversion fails, NaN entries zeroed"""
    import numpy as np
    quat = np.array([1.0, 0.0, 0.0, 0.0])
    try:
        quat = mat2quat(mat)
    except Exception:
        pass
    quat[np.isnan(quat)] = 0
    return quat


def pack_pose_rot_first(pose):
    """4x4 pose -> [quaternion (w, x, y, z) | translation] (reference core/utils.py:672-676)"""
    import numpy as np
    packed = np.zeros(7)
    packed[4:] = pose[:3, 3]
    packed[:4] = safemat2quat(pose[:3, :3])
    return packed

