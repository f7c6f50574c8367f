"""Deterministic, name-keyed parameter fill shared by the golden generator, the oracle and the
tests, so that every side starts from bit-identical weights without shipping 14 MB state dicts.
Oracle / test infrastructure only."""
import zlib

import numpy as np
import torch


def _rng(tag, seed):
    return np.random.default_rng([int(seed), zlib.crc32(tag.encode())])


def fill_value(tag, shape, seed):
    """float32 array for the parameter called ``tag`` (e.g. 'critic/linear1.weight')."""
    r = _rng(tag, seed)
    shape = tuple(int(s) for s in shape)
    leaf = tag.rsplit(".", 1)[-1]
    if len(shape) >= 2:                      # conv / linear weight
        fan_in = int(np.prod(shape[1:]))
        a = 1.4 * np.sqrt(3.0 / fan_in)
        v = r.uniform(-a, a, size=shape)
    elif leaf == "weight":                   # BatchNorm gamma: mostly positive, ~10 % negative
        v = r.uniform(0.6, 1.4, size=shape) * np.where(r.random(shape) < 0.1, -1.0, 1.0)
    else:                                    # biases / BatchNorm beta
        v = r.uniform(-0.2, 0.2, size=shape)
    return v.astype(np.float32)


def fill_module_(module, prefix, seed):
    """Overwrite every parameter of ``module`` in place (buffers keep their defaults)."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.copy_(torch.from_numpy(fill_value(prefix + "/" + name, p.shape, seed)).to(p.device))
    return module


def fill_running_stats_(module, prefix, seed):
    """Deterministic BatchNorm running statistics (mean in [-0.2, 0.2], variance in [0.5, 1.5]) for the fixtures that evaluate the
    networks in eval mode (update_parameters(test=True)): with the default buffers (0 / 1) eval-mode BatchNorm is almost an
    identity and would pin nothing."""
    with torch.no_grad():
        for name, b in module.named_buffers():
            leaf = name.rsplit(".", 1)[-1]
            r = _rng(prefix + "/" + name, seed)
            if leaf == "running_mean":
                b.copy_(torch.from_numpy(r.uniform(-0.2, 0.2, size=tuple(b.shape)).astype(np.float32)).to(b.device))
            elif leaf == "running_var":
                b.copy_(torch.from_numpy(r.uniform(0.5, 1.5, size=tuple(b.shape)).astype(np.float32)).to(b.device))
    return module


class AsymTaskSpace6D(object):
    """an action space with ASYMMETRIC bounds (the reference's PandaTaskSpace6D is symmetric, core/utils.py:505-510): the fixtures
    of GaussianPolicy's action_bias = (high + low) / 2 path (core/networks.py:329-337)"""

    def __init__(self):
        self.high = np.array([0.08, 0.05, 0.07, np.pi / 5, np.pi / 7, np.pi / 6])
        self.low = np.array([-0.04, -0.06, -0.03, -np.pi / 8, -np.pi / 6, -np.pi / 9])
        self.shape = [6]
        self.bounds = np.vstack([self.low, self.high])


SAMPLES = 24


def summarize(t):
    """Compact fingerprint of a tensor: [sum, abs-sum, l2, abs-max] + SAMPLES strided entries
    (the full tensor when it has <= 4096 entries)."""
    a = np.asarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t, dtype=np.float64).ravel()
    stats = np.array([a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum()), np.abs(a).max() if a.size else 0.0])
    if a.size <= 4096:
        return stats, a.astype(np.float32)
    pos = np.linspace(0, a.size - 1, SAMPLES).astype(np.int64)
    return stats, a[pos].astype(np.float32)


def summarize_named(named, out, key_prefix):
    for name, t in named:
        if t is None:
            continue
        s, v = summarize(t)
        out[key_prefix + name + "#stats"] = s
        out[key_prefix + name + "#vals"] = v
