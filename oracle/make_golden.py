"""Generate tests/golden/* by importing the REFERENCE's own Python (oracle tooling; runs only in the
build container where /root/reference exists -- never on the GPU box, never from the product).

Recipe: copy reference core/ + experiments/ to a scratch dir (its config.py creates
directories at import time), stub the modules this image lacks, alias `pointnet2_ops` to the
oracle restatement (the only non-reference arithmetic involved: parity unpinned at that boundary),
redirect every hard-coded 'cuda' to the CPU, then drive the reference's DDPG / BC / BaseMemory /
loss code on seeded inputs and record inputs + outputs as small .npz/.json fixtures.

    python -m oracle.make_golden            # writes tests/golden/
"""
import json
import os
import shutil
import sys
import types

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SCRATCH = "/tmp/gaddpg_ref_scratch"
OUT = os.path.join(ROOT, "tests", "golden")
SEED = 1234


# ----------------------------------------------------------------------------- shims
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(EasyDict(x) if isinstance(x, dict) else x for x in v)
        super().__setattr__(k, v)
        super().__setitem__(k, v)

    __setitem__ = __setattr__


def _cpu(dev):
    if isinstance(dev, str) and dev.startswith("cuda"):
        return "cpu"
    if isinstance(dev, torch.device) and dev.type == "cuda":
        return torch.device("cpu")
    return dev


def install_shims():
    sys.dont_write_bytecode = True
    if os.path.exists(SCRATCH):
        shutil.rmtree(SCRATCH)
    os.makedirs(SCRATCH)
    for d in ("core", "experiments"):
        shutil.copytree(os.path.join(REF, d), os.path.join(SCRATCH, d),
                        ignore=shutil.ignore_patterns("__pycache__"))
    os.chdir(SCRATCH)
    sys.path.insert(0, SCRATCH)
    sys.path.insert(0, ROOT)

    for n in ("IPython", "cv2", "matplotlib", "matplotlib.pyplot", "transforms3d",
              "transforms3d.quaternions", "transforms3d.euler", "transforms3d.axangles",
              "torchvision", "torchvision.models"):
        _stub(n)
    _stub("GPUtil", getGPUs=lambda: [])
    _stub("easydict", EasyDict=EasyDict)
    _stub("torchvision.models.resnet", BasicBlock=object, ResNet=torch.nn.Module)

    import oracle.pointnet2_ops as p2
    import oracle.pointnet2_ops.pointnet2_modules as p2m
    import oracle.pointnet2_ops.pointnet2_utils as p2u
    sys.modules["pointnet2_ops"] = p2
    sys.modules["pointnet2_ops.pointnet2_modules"] = p2m
    sys.modules["pointnet2_ops.pointnet2_utils"] = p2u

    np.int = int
    _yaml_load = yaml.load
    yaml.load = lambda s, Loader=yaml.SafeLoader: _yaml_load(s, Loader=Loader)

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor

    _t_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: _t_to(self, *[_cpu(x) for x in a],
                                                  **{kk: _cpu(v) for kk, v in k.items()})
    _m_to = torch.nn.Module.to
    torch.nn.Module.to = lambda self, *a, **k: _m_to(self, *[_cpu(x) for x in a],
                                                     **{kk: _cpu(v) for kk, v in k.items()})
    for fn in ("zeros", "ones", "tensor"):
        orig = getattr(torch, fn)
        setattr(torch, fn, (lambda o: lambda *a, **k: o(*a, **{kk: _cpu(v) for kk, v in k.items()}))(orig))


# ----------------------------------------------------------------------------- helpers
def _np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def _nets_of(agent):
    nets = {"policy": agent.policy, "policy_target": agent.policy_target,
            "state_feature_extractor": agent.state_feature_extractor}
    if hasattr(agent, "critic"):
        nets["critic"] = agent.critic
        nets["critic_target"] = agent.critic_target
    return nets


def _fill_agent(agent, seed):
    from oracle.detfill import fill_module_
    for name, net in _nets_of(agent).items():
        fill_module_(net, name, seed)


def _record_state(agent, out, prefix):
    from oracle.detfill import summarize_named
    for name, net in _nets_of(agent).items():
        summarize_named(net.state_dict().items(), out, "%sparam/%s/" % (prefix, name))


def _record_grads(agent, out, prefix, which):
    from oracle.detfill import summarize_named
    nets = _nets_of(agent)
    for name in which:
        summarize_named(((n, p.grad) for n, p in nets[name].named_parameters()), out,
                        "%sgrad/%s/" % (prefix, name))


def make_batch(cfg_name, B, n_trans, seed, n_pts=1024):
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    c = load_cfg(cfg_name)
    c.RL_TRAIN.uniform_num_pts = n_pts
    mem = BaseMemory(n_trans, c, point_dtype=np.float32)
    fill_synthetic_buffer(mem, n_trans, seed=seed)
    return sample_valid_batch(mem, B, np.random.default_rng(seed + 1))


# ----------------------------------------------------------------------------- fixtures
def _fresh_ref_cfg(ref_yaml):
    """Reference defaults + one yaml, from a freshly (re)loaded experiments/config.py."""
    import importlib
    from experiments import config as rc
    rc = importlib.reload(rc)
    rc.cfg_from_file(os.path.join("experiments/cfgs", ref_yaml))
    return rc.cfg


def gen_config():
    res = {}
    for ref_yaml, mine in (("td3_critic_aux_policy_aux.yaml", "ddpg_td3_aux.yaml"),
                           ("bc_aux_dagger.yaml", "bc_dagger_aux.yaml"),
                           ("bc_save_data.yaml", "bc_save_data.yaml")):
        c = _fresh_ref_cfg(ref_yaml)
        top = {k: c[k] for k in ("RL_MAX_STEP", "RL_SAVE_DATA_NAME", "RL_MEMORY_SIZE",
                                 "OFFLINE_RL_MEMORY_SIZE", "OFFLINE_BATCH_SIZE", "ONPOLICY_MEMORY_SIZE")}
        res[mine] = {"top": top, "RL_TRAIN": dict(c["RL_TRAIN"].items())}
    with open(os.path.join(OUT, "config_rl_train.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True, default=lambda o: list(o))


def _make_agent(kind, ref_yaml, space=None):
    """Build the reference agent exactly as train_test_offline.py does (setup() :60-104, :347-349)."""
    import importlib
    cfg = _fresh_ref_cfg(ref_yaml)
    from core.utils import make_nets_opts_schedulers, PandaTaskSpace6D
    net_dict = make_nets_opts_schedulers(cfg.RL_MODEL_SPEC, cfg.RL_TRAIN)
    mod = importlib.import_module("core.ddpg" if kind == "DDPG" else "core.bc")
    agent = getattr(mod, kind)(cfg.RL_TRAIN.feature_input_dim, space or PandaTaskSpace6D(), cfg.RL_TRAIN)
    agent.setup_feature_extractor(net_dict, False)
    return agent, cfg


def _hooked(agent, feats, rands, crit_snap):
    orig_extract = agent.extract_feature
    def extract(*a, **k):
        f = orig_extract(*a, **k)
        feats.append(_np(f).copy())
        return f
    agent.extract_feature = extract
    if hasattr(agent, "critic_optimize"):
        orig_co = agent.critic_optimize
        def critic_optimize():
            orig_co()
            crit_snap.clear()
            _record_grads(agent, crit_snap, "", ["critic", "state_feature_extractor"])
        agent.critic_optimize = critic_optimize


# Minibatch seeds of the golden DDPG runs.  They are CHOSEN (find_ddpg_seeds below) so that no pre-activation of the two FC
# BatchNorm1d layers -- 32 rows per channel -- lies within 2e-5 of the ReLU kink in the reference's float64 evaluation: one
# such tie resolved differently by two float32 evaluations removes / adds a whole row's contribution (1/32) to every
# gradient upstream (measured: one flip at fc[1] moved every value-encoder tensor by 2 - 4e-3 of its scale), so a fixture
# sitting on a tie cannot pin anybody's arithmetic to better than that.  With 5e4 such pre-activations per pass a random
# minibatch carries a tie with probability ~ 0.4; the layers with thousands of rows (SA stages) keep their ties -- there
# a flip is one row in 1e3 - 3e4.
DDPG_BATCH_SEED = {"a": SEED + 8000, "b": SEED + 100 + 3000}     # min |fc pre-activation| 1.8e-5 / 1.8e-5 (float64)


def _fc_kink_distance(agent, fn):
    """min |BatchNorm1d output| over every pass `fn()` runs through the agent's feature extractor"""
    dist = [np.inf]
    hooks = []
    for m in agent.state_feature_extractor.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            hooks.append(m.register_forward_hook(lambda mod, i, o: dist.__setitem__(0, min(dist[0], float(o.detach().abs().min())))))
    try:
        out = fn()
    finally:
        for h in hooks:
            h.remove()
    return dist[0], out


def find_ddpg_seeds(B=32, tries=12, margin=2e-5):
    """python -m oracle.make_golden seeds: candidate minibatch seeds whose float64 reference step stays `margin` away from
    every FC-level ReLU kink (prints the distances; the chosen ones go into DDPG_BATCH_SEED)"""
    orig_rand_like, orig_ft, orig_float = torch.rand_like, torch.cuda.FloatTensor, torch.Tensor.float
    torch.cuda.FloatTensor = torch.DoubleTensor
    torch.Tensor.float = lambda self, *a, **k: self.double()
    found = {}
    try:
        for run, start, base in (("a", 1, SEED), ("b", 2, SEED + 100)):
            for t in range(tries):
                seed = base + 1000 * t
                agent, cfg = _make_agent("DDPG", "td3_critic_aux_policy_aux.yaml")
                _fill_agent(agent, SEED)
                for net in _nets_of(agent).values():
                    net.double()
                for pol in (agent.policy, agent.policy_target):
                    pol.action_scale, pol.action_bias = pol.action_scale.double(), pol.action_bias.double()
                agent.update_step = start
                batch = make_batch("ddpg_td3_aux.yaml", B, 1200, seed)
                torch.manual_seed(SEED)
                d, _ = _fc_kink_distance(agent, lambda: agent.update_parameters(batch, agent.update_step, 0))
                print("run %s seed %d: min |fc pre-activation| = %.3e %s" % (run, seed, d, "OK" if d > margin else ""), flush=True)
                if d > margin:
                    found[run] = seed
                    break
    finally:
        torch.rand_like, torch.cuda.FloatTensor, torch.Tensor.float = orig_rand_like, orig_ft, orig_float
    return found


def gen_ddpg(B=32):
    """Runs: 'a' starts at update_step 1 (no actor-critic term), 'b' at update_step 2 (policy step) and
    continues for a second step.  a0 and b0 start from identical det-filled parameters, so they pin the
    arithmetic of one step tightly; b1 additionally exercises optimiser state / schedulers."""
    out = {}
    orig_rand_like = torch.rand_like
    rands = []
    def rand_like(x, *a, **k):
        r = orig_rand_like(x, *a, **k)
        rands.append(_np(r).copy())
        return r
    torch.rand_like = rand_like
    ret = None
    for run, start, nsteps in (("a", 1, 1), ("b", 2, 2)):
        agent, cfg = _make_agent("DDPG", "td3_critic_aux_policy_aux.yaml")
        _fill_agent(agent, SEED)
        agent.update_step = start
        feats, crit_snap = [], {}
        _hooked(agent, feats, rands, crit_snap)
        torch.manual_seed(SEED)
        for s in range(nsteps):
            batch = make_batch("ddpg_td3_aux.yaml", B, 1200, DDPG_BATCH_SEED[run] + 10 * s)
            feats.clear(); rands.clear()
            p = "%s%d/" % (run, s)
            for k, v in batch.items():
                if k not in ("state_pose_batch", "grasp_sample_batch", "image_state_batch", "next_image_state_batch"):
                    out[p + "batch/" + k] = np.asarray(v)
            out[p + "update_step"] = np.int64(agent.update_step)
            ret = agent.update_parameters(batch, agent.update_step, s)
            agent.step_scheduler(agent.update_step)
            assert len(rands) == 1
            out[p + "noise_u"] = rands[0]
            for i, f in enumerate(feats):
                out[p + "feat%d" % i] = f      # value_feat, next_state, next_target, policy_feat[, value_pi]
            for k, v in ret.items():
                out[p + "ret/" + k] = np.float64(v)
            for k in ("qf1", "qf2", "next_q_value", "critic_grasp_aux", "pi", "aux_pred"):
                out[p + "t/" + k] = _np(getattr(agent, k))
            if agent.update_step % 2 == 1:         # update_step was incremented: an even (policy) step just ran
                out[p + "t/qf1_pi"] = _np(agent.qf1_pi)
                out[p + "t/qf2_pi"] = _np(agent.qf2_pi)
            for k, v in crit_snap.items():
                out[p + "critic_phase/" + k] = v
            _record_grads(agent, out, p + "end/", ["policy", "critic", "state_feature_extractor"])
            _record_state(agent, out, p + "end/")
            out[p + "lr"] = np.array([agent.get_lr()[k] for k in ("policy_lr", "feature_lr", "value_lr")])
    torch.rand_like = orig_rand_like
    np.savez_compressed(os.path.join(OUT, "ddpg_steps_B%d.npz" % B), **out)
    return ret


def gen_ddpg_test_mode(B=32):
    """update_parameters(..., test=True) (core/ddpg.py:146-150, core/agent.py:261-280): the SAME update with every online network in
    eval mode -- BatchNorm normalises with its running statistics in all five encoder passes, its backward has no batch-statistics
    terms, running statistics and num_batches_tracked stay as they are.  No driver of the reference calls it that way; the fixture
    pins the form all the same.  Runs: 'e' at update_step 1 (no actor-critic term), 'f' at update_step 2 (policy step), one step each,
    from det-filled parameters AND det-filled running statistics."""
    from oracle.detfill import fill_running_stats_
    out = {}
    orig_rand_like = torch.rand_like
    rands = []
    def rand_like(x, *a, **k):
        r = orig_rand_like(x, *a, **k)
        rands.append(_np(r).copy())
        return r
    torch.rand_like = rand_like
    ret = None
    for run, start, like in (("e", 1, "a"), ("f", 2, "b")):
        agent, cfg = _make_agent("DDPG", "td3_critic_aux_policy_aux.yaml")
        _fill_agent(agent, SEED)
        for name, net in _nets_of(agent).items():
            fill_running_stats_(net, name, SEED)
        agent.update_step = start
        feats, crit_snap = [], {}
        _hooked(agent, feats, rands, crit_snap)
        torch.manual_seed(SEED)
        batch = make_batch("ddpg_td3_aux.yaml", B, 1200, DDPG_BATCH_SEED[like])
        rands.clear()
        p = "%s0/" % run
        for k, v in batch.items():
            if k not in ("state_pose_batch", "grasp_sample_batch", "image_state_batch", "next_image_state_batch"):
                out[p + "batch/" + k] = np.asarray(v)
        out[p + "update_step"] = np.int64(agent.update_step)
        ret = agent.update_parameters(batch, agent.update_step, 0, test=True)
        agent.step_scheduler(agent.update_step)
        assert len(rands) == 1 and not agent.state_feature_extractor.training
        out[p + "noise_u"] = rands[0]
        for i, f in enumerate(feats):
            out[p + "feat%d" % i] = f
        for k, v in ret.items():
            out[p + "ret/" + k] = np.float64(v)
        for k in ("qf1", "qf2", "next_q_value", "critic_grasp_aux", "pi", "aux_pred"):
            out[p + "t/" + k] = _np(getattr(agent, k))
        if agent.update_step % 2 == 1:
            out[p + "t/qf1_pi"] = _np(agent.qf1_pi)
            out[p + "t/qf2_pi"] = _np(agent.qf2_pi)
        for k, v in crit_snap.items():
            out[p + "critic_phase/" + k] = v
        _record_grads(agent, out, p + "end/", ["policy", "critic", "state_feature_extractor"])
        _record_state(agent, out, p + "end/")
        out[p + "lr"] = np.array([agent.get_lr()[k] for k in ("policy_lr", "feature_lr", "value_lr")])
    torch.rand_like = orig_rand_like
    np.savez_compressed(os.path.join(OUT, "ddpg_steps_test_mode_B%d.npz" % B), **out)
    return ret


def gen_ddpg_asym_bounds(B=32):
    """One policy step (update_step 2: target action, Q(s, pi(s)) and the BC loss all see pi = tanh(mean) * scale + bias) of the
    reference's DDPG with an action space whose bounds are NOT symmetric (oracle.detfill.AsymTaskSpace6D; core/networks.py:329-337:
    action_bias = (high + low) / 2 != 0) -- the reference's own PandaTaskSpace6D never exercises the bias."""
    from oracle.detfill import AsymTaskSpace6D
    out = {}
    orig_rand_like = torch.rand_like
    rands = []
    def rand_like(x, *a, **k):
        r = orig_rand_like(x, *a, **k)
        rands.append(_np(r).copy())
        return r
    torch.rand_like = rand_like
    agent, cfg = _make_agent("DDPG", "td3_critic_aux_policy_aux.yaml", space=AsymTaskSpace6D())
    assert float(agent.policy.action_bias.abs().max()) > 0.01
    _fill_agent(agent, SEED)
    agent.update_step = 2
    feats, crit_snap = [], {}
    _hooked(agent, feats, rands, crit_snap)
    torch.manual_seed(SEED)
    batch = make_batch("ddpg_td3_aux.yaml", B, 1200, DDPG_BATCH_SEED["b"])
    p = "g0/"
    for k, v in batch.items():
        if k not in ("state_pose_batch", "grasp_sample_batch", "image_state_batch", "next_image_state_batch"):
            out[p + "batch/" + k] = np.asarray(v)
    out[p + "update_step"] = np.int64(agent.update_step)
    ret = agent.update_parameters(batch, agent.update_step, 0)
    agent.step_scheduler(agent.update_step)
    out[p + "noise_u"] = rands[0]
    out[p + "action_bias"] = _np(agent.policy.action_bias)
    for i, f in enumerate(feats):
        out[p + "feat%d" % i] = f
    for k, v in ret.items():
        out[p + "ret/" + k] = np.float64(v)
    for k in ("qf1", "qf2", "next_q_value", "critic_grasp_aux", "pi", "aux_pred", "qf1_pi", "qf2_pi"):
        out[p + "t/" + k] = _np(getattr(agent, k))
    for k, v in crit_snap.items():
        out[p + "critic_phase/" + k] = v
    _record_grads(agent, out, p + "end/", ["policy", "critic", "state_feature_extractor"])
    _record_state(agent, out, p + "end/")
    out[p + "lr"] = np.array([agent.get_lr()[k] for k in ("policy_lr", "feature_lr", "value_lr")])
    torch.rand_like = orig_rand_like
    np.savez_compressed(os.path.join(OUT, "ddpg_steps_asym_bounds_B%d.npz" % B), **out)
    return ret


def gen_ddpg_f64(B=32):
    """The reference's OWN update step evaluated in float64 on the a0 / b0 inputs of gen_ddpg (same det-filled weights,
    same batches, the recorded noise draw): the yardstick for gradient accuracy -- a float32 implementation is judged by
    its distance to this, relative to the distance of the reference's float32 run (ddpg_steps_B32.npz) to it."""
    g32 = np.load(os.path.join(OUT, "ddpg_steps_B%d.npz" % B))
    out = {}
    orig_rand_like, orig_ft, orig_float = torch.rand_like, torch.cuda.FloatTensor, torch.Tensor.float
    torch.cuda.FloatTensor = torch.DoubleTensor
    torch.Tensor.float = lambda self, *a, **k: self.double()        # the reference's hard-coded .float() casts (control points)
    try:
        for run, start in (("a", 1), ("b", 2)):
            agent, cfg = _make_agent("DDPG", "td3_critic_aux_policy_aux.yaml")
            _fill_agent(agent, SEED)
            for net in _nets_of(agent).values():
                net.double()
            agent.policy.action_scale = agent.policy.action_scale.double()
            agent.policy.action_bias = agent.policy.action_bias.double()
            agent.policy_target.action_scale = agent.policy_target.action_scale.double()
            agent.policy_target.action_bias = agent.policy_target.action_bias.double()
            agent.update_step = start
            feats, crit_snap = [], {}
            _hooked(agent, feats, [], crit_snap)
            p = "%s0/" % run
            batch = make_batch("ddpg_td3_aux.yaml", B, 1200, DDPG_BATCH_SEED[run])                  # gen_ddpg's s = 0 batches
            assert np.array_equal(batch["point_state_batch"], g32[p + "batch/point_state_batch"])
            u = g32[p + "noise_u"]
            torch.rand_like = lambda x, *a, **k: torch.tensor(u, dtype=x.dtype)
            ret = agent.update_parameters(batch, agent.update_step, 0)
            for k, v in ret.items():
                out[p + "ret/" + k] = np.float64(v)
            for k in ("qf1", "qf2", "next_q_value", "pi"):
                out[p + "t/" + k] = _np(getattr(agent, k))
            for k, v in crit_snap.items():
                out[p + "critic_phase/" + k] = v
            _record_grads(agent, out, p + "end/", ["policy", "critic", "state_feature_extractor"])
    finally:
        torch.rand_like, torch.cuda.FloatTensor, torch.Tensor.float = orig_rand_like, orig_ft, orig_float
    np.savez_compressed(os.path.join(OUT, "ddpg_steps_B%d_f64.npz" % B), **out)
    return {k: float(v) for k, v in out.items() if "/ret/" in k and "loss" in k}


def gen_bc(B=32, steps=2):
    agent, cfg = _make_agent("BC", "bc_aux_dagger.yaml")
    _fill_agent(agent, SEED + 1)
    out = {}
    feats = []
    _hooked(agent, feats, [], {})
    torch.manual_seed(SEED)
    for s in range(steps):
        batch = make_batch("bc_dagger_aux.yaml", B, 1200, SEED + 100 + 10 * s)
        feats.clear()
        p = "a%d/" % s
        for k, v in batch.items():
            if k not in ("state_pose_batch", "grasp_sample_batch", "image_state_batch", "next_image_state_batch",
                         "next_point_state_batch"):
                out[p + "batch/" + k] = np.asarray(v)
        ret = agent.update_parameters(batch, agent.update_step, s)
        agent.step_scheduler(agent.update_step)
        out[p + "feat0"] = feats[0]
        for k, v in ret.items():
            out[p + "ret/" + k] = np.float64(v)
        for k in ("pi", "aux_pred"):
            out[p + "t/" + k] = _np(getattr(agent, k))
        _record_grads(agent, out, p + "end/", ["policy", "state_feature_extractor"])
        _record_state(agent, out, p + "end/")
    np.savez_compressed(os.path.join(OUT, "bc_steps_B%d.npz" % B), **out)
    return ret


def gen_losses():
    from core import loss as rl
    from core.utils import get_noise_delta
    rng = np.random.default_rng(SEED)
    out = {}
    B = 16
    q = rng.normal(size=(B, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    pred = torch.tensor(np.concatenate([q, rng.uniform(-0.2, 0.2, (B, 3))], 1), dtype=torch.float32, requires_grad=True)
    q2 = rng.normal(size=(B, 4)); q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
    goal = torch.tensor(np.concatenate([q2, rng.uniform(-0.2, 0.2, (B, 3))], 1), dtype=torch.float32)
    l = rl.goal_pred_loss(pred, goal)
    l.backward()
    out.update(goal_pred=_np(pred), goal_gt=_np(goal), goal_loss=_np(l), goal_grad=_np(pred.grad))
    hi = np.array([0.06] * 3 + [np.pi / 6] * 3)
    pi = torch.tensor(rng.uniform(-hi, hi, (B, 6)), dtype=torch.float32, requires_grad=True)
    act = torch.tensor(rng.uniform(-hi, hi, (B, 6)), dtype=torch.float32)
    l2 = rl.pose_bc_loss(pi, act)
    l2.backward()
    out.update(bc_pi=_np(pi), bc_act=_np(act), bc_loss=_np(l2), bc_grad=_np(pi.grad))
    # target-noise quirk (utils.py:568-584): uniform noise is (u*3-6)*level, rot part x5
    u = torch.tensor(rng.random((B, 6)), dtype=torch.float32)
    orig = torch.rand_like
    torch.rand_like = lambda x, *a, **k: u.clone()
    nd = get_noise_delta(torch.zeros(B, 6), 0.03, "uniform")
    torch.rand_like = orig
    out.update(noise_u=_np(u), noise_level=np.float64(0.03), noise_delta=_np(nd))
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)


def gen_heads():
    from core import networks as rn
    from core.utils import PandaTaskSpace6D
    from oracle.detfill import fill_module_
    rng = np.random.default_rng(SEED + 5)
    B = 12
    x = torch.tensor(rng.normal(size=(B, 513)), dtype=torch.float32)
    out = {"x": _np(x)}
    q = fill_module_(rn.QNetwork(513, 0, 256, extra_pred_dim=7), "critic", SEED)
    q1, q2, aux = q(x, None)
    out.update(q1=_np(q1), q2=_np(q2), aux=_np(aux))
    p = fill_module_(rn.GaussianPolicy(513, 6, 256, PandaTaskSpace6D(), extra_pred_dim=7), "policy", SEED)
    # Normal.rsample draws eps through torch.distributions.normal._standard_normal: inject a recorded draw so that the
    # reparameterised action and its log-prob (networks.py:355-368) become fixtures too
    import torch.distributions.normal as tdn
    eps = torch.tensor(rng.normal(size=(B, 6)), dtype=torch.float32)
    orig = tdn._standard_normal
    tdn._standard_normal = lambda shape, dtype, device: eps.clone().to(dtype)
    mean, logp, act, extra = p.sample(x)
    tdn._standard_normal = orig
    m0, logstd, _ = p.forward(x)
    out.update(pi_mean=_np(mean), pi_extra=_np(extra), pi_raw_mean=_np(m0), pi_log_std=_np(logstd), pi_eps=_np(eps),
               pi_log_prob=_np(logp), pi_action=_np(act))
    # the head without policy_aux (extra_pred_dim = 1, core/agent.py:31-36): raw extra column, no quaternion
    p1 = fill_module_(rn.GaussianPolicy(513, 6, 256, PandaTaskSpace6D(), extra_pred_dim=1), "policy", SEED)
    tdn._standard_normal = lambda shape, dtype, device: eps.clone().to(dtype)
    mean1, logp1, act1, extra1 = p1.sample(x)
    tdn._standard_normal = orig
    out.update(p1_mean=_np(mean1), p1_extra=_np(extra1), p1_log_prob=_np(logp1), p1_action=_np(act1))
    np.savez_compressed(os.path.join(OUT, "heads.npz"), **out)


def gen_offpath():
    """Two forms the update step never reaches (VERDICT r04 missing 4), through the reference's own classes:
    GoalFeature.forward (core/networks.py:150-178; eval and train mode) and the non-value_model QNetwork(state, action)
    (core/networks.py:280-300)."""
    from core import networks as rn
    from oracle.detfill import fill_module_
    rng = np.random.default_rng(SEED + 11)
    out = {}
    B, N = 4, 1024
    pc = torch.tensor(rng.uniform(-0.15, 0.15, size=(B, N, 3)), dtype=torch.float32)
    g = fill_module_(rn.GoalFeature(), "goal_feature", SEED)
    for mode in ("train", "eval"):
        g.train(mode == "train")
        with torch.no_grad():
            qt, conf = g(pc)
        out["goal_qt_" + mode], out["goal_conf_" + mode] = _np(qt), _np(conf)
    out["goal_pc"] = _np(pc)
    state = torch.tensor(rng.normal(size=(12, 512)), dtype=torch.float32)
    action = torch.tensor(rng.uniform(-1, 1, size=(12, 6)), dtype=torch.float32)
    q = fill_module_(rn.QNetwork(512, 6, 256, extra_pred_dim=7), "critic_sa", SEED)
    q1, q2, aux = q(state, action)
    out.update(q_state=_np(state), q_action=_np(action), q1=_np(q1), q2=_np(q2), q_aux=_np(aux))
    q0 = fill_module_(rn.QNetwork(512, 6, 256, extra_pred_dim=0), "critic_sa0", SEED)
    r1, r2, r3 = q0(state, action)
    assert r3 is None
    out.update(q0_q1=_np(r1), q0_q2=_np(r2))
    np.savez_compressed(os.path.join(OUT, "offpath_forms.npz"), **out)


def gen_encoder(B=16):
    """PointNetFeature forward (both encoders) + grads of a scalar probe, via the reference's class."""
    from core import networks as rn
    from oracle.detfill import fill_module_, summarize_named
    net = rn.PointNetFeature(input_dim=5, extra_latent=1, action_concat=True)
    fill_module_(torch.nn.DataParallel(net), "state_feature_extractor", SEED)
    net.train()
    batch = make_batch("ddpg_td3_aux.yaml", B, 300, SEED + 7)
    pc = torch.tensor(batch["point_state_batch"], dtype=torch.float32)
    act = torch.tensor(batch["action_batch"], dtype=torch.float32, requires_grad=True)
    out = {"point_state": _np(pc), "action": _np(act)}
    taps = {}
    for tag, enc in (("policy", net.encoder), ("value", net.value_encoder)):
        for i, sa in enumerate(enc[0]):
            sa.register_forward_hook(lambda m, a, o, k="%s_sa%d" % (tag, i + 1): taps.__setitem__(k, _np(o[1]).copy()))
    z_pol, _ = net(pc, feature_2=False)
    pc10 = torch.cat((pc, act.unsqueeze(2).expand(-1, -1, pc.shape[2])), 1)
    z_val, _ = net(pc10, feature_2=True)
    probe = torch.tensor(np.random.default_rng(SEED).normal(size=(B, 512)), dtype=torch.float32)
    ((z_pol * probe).sum() + (z_val * probe.flip(1)).sum()).backward()
    out.update(z_policy=_np(z_pol), z_value=_np(z_val), probe=_np(probe), action_grad=_np(act.grad))
    out.update(taps)                       # pooled SA outputs (B,C,npoint): well-conditioned check points
    summarize_named(((n, p.grad) for n, p in net.named_parameters()), out, "grad/")
    summarize_named(net.state_dict().items(), out, "state/")
    np.savez_compressed(os.path.join(OUT, "encoder_B%d.npz" % B), **out)


def gen_replay():
    """BaseMemory.sample on a seeded synthetic buffer through the reference's own class."""
    from core.replay_memory import BaseMemory as RefMemory
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer
    cfg = _fresh_ref_cfg("td3_critic_aux_policy_aux.yaml")
    cfg.RL_TRAIN.uniform_num_pts = 64         # small clouds keep the fixture tiny
    mem = RefMemory(400, cfg)
    fill_synthetic_buffer(mem, 400, seed=SEED + 3)
    np.random.seed(SEED)
    data = mem.sample(16)
    out = {"buffer/" + k: getattr(mem, k) for k in
           ("action", "expert_action", "point_state", "reward", "terminal", "timestep", "returns",
            "goal", "episode_map", "expert_flags", "perturb_flags")}
    out["buffer/cur_idx"] = np.int64(mem.cur_idx)
    for k, v in data.items():
        out["batch/" + k] = np.asarray(v)
    mem.recompute_return_with_gamma()
    out["recomputed_returns"] = mem.returns
    np.savez_compressed(os.path.join(OUT, "replay_sample.npz"), **out)


def _episodes(n_pts, seed):
    """seeded rollouts in the reference's transition-dict format (add_episode / push input)"""
    rng = np.random.default_rng(seed)
    eps = []
    for e in range(7):
        L = int(rng.integers(3, 9))
        success = bool(e % 3 != 1)
        ep = []
        for t in range(L):
            cloud = rng.normal(size=(4, n_pts + 6)) * 0.1
            if e == 4 and t == 2:
                cloud[:] = 0.0                                # all-zero frame: push() drops it
            ep.append({"point_state": cloud, "action": rng.normal(size=6).astype(np.float32),
                       "expert_action": rng.normal(size=6).astype(np.float32), "goal": rng.normal(size=7).astype(np.float32),
                       "reward": np.float32(1.0 if (success and t == L - 1) else 0.0), "terminal": np.float32(t == L - 1),
                       "timestep": np.float32(t), "expert_flags": np.float32(e % 2), "perturb_flags": np.float32(t == 1),
                       "state_pose": np.eye(4, dtype=np.float32), "target_idx": np.float32(e), "target_name": "obj%d" % (e % 3)})
        eps.append(ep)
    return eps


def gen_replay_io():
    """writer side + on-disk format through the reference's own BaseMemory: add_episode -> save -> (fresh) load -> sample.
    The saved .npz is committed as a fixture (a data file in the reference's format); the episodes are regenerated
    from the seed by the test."""
    import tempfile
    from core.replay_memory import BaseMemory as RefMemory
    cfg = _fresh_ref_cfg("td3_critic_aux_policy_aux.yaml")
    cfg.RL_TRAIN.uniform_num_pts = 128
    cfg.RL_SAVE_DATA_NAME = "replay_io_saved.npz"
    mem = RefMemory(48, cfg)
    for ep in _episodes(128, SEED + 9):
        mem.add_episode(ep)
    out = {"after_add/" + k: np.asarray(getattr(mem, k)) for k in
           ("action", "reward", "returns", "terminal", "timestep", "episode_map", "expert_flags", "perturb_flags", "goal")}
    out["after_add/cur_idx"] = np.int64(mem.cur_idx)
    out["after_add/total_env_step"] = np.int64(mem.total_env_step)
    out["after_add/is_full"] = np.bool_(mem.is_full)
    mem.save(OUT)                                             # -> tests/golden/replay_io_saved.npz
    mem2 = RefMemory(48, cfg)
    mem2.load(OUT)
    out["after_load/cur_idx"] = np.int64(mem2.cur_idx)
    out["after_load/returns"] = np.asarray(mem2.returns)
    out["after_load/point_state_sum"] = np.float64(mem2.point_state.sum())
    np.random.seed(SEED + 1)
    mem2.episode_max_len = 2
    data = mem2.sample(8)
    for k, v in data.items():
        out["batch/" + k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "replay_io.npz"), **out)


def _manifest(obj):
    """key / shape / dtype structure of a checkpoint object (tensors -> [shape, dtype]; scalars -> type name)"""
    if torch.is_tensor(obj):
        return {"tensor": list(obj.shape), "dtype": str(obj.dtype).replace("torch.", "")}
    if isinstance(obj, dict):
        return {"dict": {str(k): _manifest(v) for k, v in obj.items()}, "int_keys": all(isinstance(k, int) for k in obj) and len(obj) > 0}
    if isinstance(obj, (list, tuple)):
        return {"list": [_manifest(v) for v in obj]}
    return {"scalar": type(obj).__name__}


def gen_checkpoint(B=8):
    """A checkpoint set WRITTEN BY THE REFERENCE's own Agent.save_model (core/agent.py:282-346) after one DDPG update (so the
    Adam states exist): file names, dict keys, state-dict keys, tensor shapes / dtypes, optimiser-state structure.  Only the
    manifest is kept (the files themselves are 40 MB of det-filled weights): tests/test_oracle_golden.py checks that our
    save_model writes exactly this structure and that load_model accepts a file set built from it."""
    import tempfile
    agent, cfg = _make_agent("DDPG", "td3_critic_aux_policy_aux.yaml")
    _fill_agent(agent, SEED)
    agent.update_step = 2
    torch.manual_seed(SEED)
    batch = make_batch("ddpg_td3_aux.yaml", B, 600, 77)
    agent.update_parameters(batch, agent.update_step, 0)
    agent.step_scheduler(agent.update_step)
    out = {"files": {}, "step_saved": 7, "agent_name": agent.name, "env_name": agent.env_name}
    with tempfile.TemporaryDirectory() as d:
        agent.save_model(7, output_dir=d, surfix="latest")
        for f in sorted(os.listdir(d)):
            out["files"][f] = _manifest(torch.load(os.path.join(d, f), weights_only=False))
    with open(os.path.join(OUT, "checkpoint_manifest.json"), "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    return sorted(out["files"])


def main():
    """python -m oracle.make_golden [name ...]: regenerate all fixtures, or only the named generators
    (config losses heads replay replay_io encoder bc ddpg ...)"""
    os.makedirs(OUT, exist_ok=True)
    install_shims()
    torch.set_num_threads(8)
    gens = [("config", gen_config), ("losses", gen_losses), ("heads", gen_heads), ("replay", gen_replay),
            ("replay_io", gen_replay_io), ("offpath", gen_offpath), ("encoder", gen_encoder), ("bc", gen_bc), ("ddpg", gen_ddpg),
            ("ddpg_f64", gen_ddpg_f64), ("ddpg_test_mode", gen_ddpg_test_mode), ("ddpg_asym_bounds", gen_ddpg_asym_bounds),
            ("checkpoint", gen_checkpoint)]
    if sys.argv[1:] == ["seeds"]:
        print(find_ddpg_seeds())
        return
    only = sys.argv[1:]
    for name, fn in gens:
        if not only or name in only:
            print(name, "->", fn())
    for f in sorted(os.listdir(OUT)):
        print("%-28s %8.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))


if __name__ == "__main__":
    main()
