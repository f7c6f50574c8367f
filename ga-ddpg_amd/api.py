"""Convenience constructors that assemble an agent exactly the way the reference's offline driver
does (reference core/train_test_offline.py:60-104 setup(), :347-349)."""
import torch

from .core.bc import BC
from .core.ddpg import DDPG
from .core.utils import PandaTaskSpace6D, make_nets_opts_schedulers
from .experiments.config import load_cfg


def make_agent(cfg_name_or_cfg="ddpg_td3_aux.yaml", kind=None, action_space=None):
    """-> (agent, cfg).  kind defaults to 'DDPG' when cfg.RL_TRAIN.RL else 'BC' (the reference's
    train_test_offline.py:329 ignores its --policy flag the same way)."""
    if not torch.cuda.is_available():
        raise RuntimeError("ga_ddpg_amd needs an MI355X (HIP) device: the update step has no CPU fallback")
    cfg = load_cfg(cfg_name_or_cfg) if isinstance(cfg_name_or_cfg, str) else cfg_name_or_cfg
    train = cfg.RL_TRAIN
    kind = kind or ("DDPG" if train.RL else "BC")
    net_dict = make_nets_opts_schedulers(cfg.RL_MODEL_SPEC, train)
    agent = (DDPG if kind == "DDPG" else BC)(train.feature_input_dim, action_space or PandaTaskSpace6D(), train)
    agent.setup_feature_extractor(net_dict, False)
    return agent, cfg
