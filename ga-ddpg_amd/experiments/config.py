"""Config defaults + yaml merge for the update-step path.

Mirrors the surface of the reference's experiments/config.py (``cfg``, ``cfg_from_file``,
``process_cfg``, ``_merge_a_into_b``; reference config.py:31-177, 180-259, 275-315) for the keys
the path consumes.  Differences by design: no import-time directory creation (reference
config.py:22-29), ``yaml.safe_load`` instead of the Loader-less ``yaml.load`` (config.py:305),
and the env / OMG planner dictionaries (config.py:207-259) are out of scope.
"""
import copy
import os

import yaml

_CFG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")


class AttrDict(dict):
    """dict with attribute access (stands in for easydict.EasyDict, which is not a dependency)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _load_yaml(path):
    with open(path, "r") as f:
        return yaml.safe_load(f) or {}


def default_cfg():
    c = AttrDict(_load_yaml(os.path.join(_CFG_DIR, "defaults.yaml")))
    c.RL_MODEL_SPEC = os.path.join(_CFG_DIR, "rl_pointnet_model_spec.yaml")
    return c


cfg = default_cfg()


def _merge_a_into_b(a, b):
    """Same rule as the reference (config.py:275-298): a key of ``a`` is taken only if it already
    exists in ``b`` with exactly the same Python type; dicts merge recursively."""
    if not isinstance(a, dict):
        return
    for k, v in a.items():
        if k not in b:
            continue
        if isinstance(v, dict) and isinstance(b[k], dict):
            _merge_a_into_b(v, b[k])
        elif type(b[k]) is type(v):
            b[k] = v


def process_cfg(c=None, reset_model_spec=True):
    """Coupling rules of reference config.py:180-205 that touch the path."""
    c = cfg if c is None else c
    t = c.RL_TRAIN
    if t.onpolicy and t.RL:
        t.explore_cap = 1.0
    if t.self_supervision and t.RL:
        t.expert_initial_state = False
        t.explore_ratio = 1.0
        t.action_noise = 0.0
    if t.use_image:
        t.domain_randomization = True
    if reset_model_spec and not t.use_image:
        c.RL_MODEL_SPEC = os.path.join(_CFG_DIR, "rl_pointnet_model_spec.yaml")
    if t.sa_channel_concat:
        t.value_model = True
    if t.policy_goal:
        t.train_goal_feature = True
    return c


def cfg_from_file(filename=None, dict=None, reset_model_spec=True):
    """Merge a yaml file into the global ``cfg`` (or ``dict``) and apply process_cfg."""
    target = cfg if dict is None else dict
    _merge_a_into_b(_load_yaml(filename), target)
    return process_cfg(target, reset_model_spec=reset_model_spec)


def load_cfg(name_or_path):
    """Fresh (non-global) config: defaults merged with a yaml from configs/ or an explicit path."""
    c = default_cfg()
    path = name_or_path if os.path.exists(name_or_path) else os.path.join(_CFG_DIR, name_or_path)
    return cfg_from_file(path, dict=c)


def save_cfg_to_file(filename, c):
    def plain(d):
        return {k: plain(v) if isinstance(v, dict) else v for k, v in d.items()}
    with open(filename, "w+") as f:
        yaml.dump(plain(c), f, default_flow_style=False)
