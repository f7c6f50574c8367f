"""Module-level API (the reference's nn.Module surface evaluated through libgaddpg, forward only) against the
CPU oracle: PointnetSAModule, PointNetFeature (train + eval BatchNorm), QNetwork, GaussianPolicy, select_action."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu
SEED = 99


def _copy_params(dst, src):
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    dst.load_state_dict(sd)


@pytest.mark.parametrize("group_all", [False, True])
def test_sa_module_forward_matches_oracle(group_all):
    from ga_ddpg_amd.pointnet2_ops import pointnet2_modules as pm
    from oracle.pointnet2_ops import pointnet2_modules as opm
    from oracle.detfill import fill_module_
    rng = np.random.default_rng(0)
    B, N, C = 3, 512, 8
    kw = dict(mlp=[C, 32, 32, 64]) if group_all else dict(npoint=16, radius=0.08, nsample=32, mlp=[C, 32, 32, 64])
    ref = fill_module_(opm.PointnetSAModule(**kw), "sa", SEED).train()
    mine = fill_module_(pm.PointnetSAModule(**kw), "sa", SEED).train()
    xyz = torch.tensor(rng.random((B, N, 3)) * 0.3 + 0.2, dtype=torch.float32)
    feats = torch.tensor(rng.normal(size=(B, C, N)), dtype=torch.float32)
    with torch.no_grad():
        w_xyz, w_out = ref(xyz, feats)
        g_xyz, g_out = mine(xyz.cuda(), feats.cuda())
    if group_all:
        assert g_xyz is None and w_xyz is None
    else:
        np.testing.assert_array_equal(g_xyz.cpu().numpy(), w_xyz.numpy())
    assert_close(g_out.cpu().numpy(), w_out.numpy(), 1e-4, 2e-5, "SA output")
    for (n, a), (_, b) in zip(mine.state_dict().items(), ref.state_dict().items()):
        if "running" in n:
            assert_close(a.cpu().numpy(), b.numpy(), 1e-4, 1e-6, n)
    # eval mode uses the running statistics
    ref.eval(); mine.eval()
    with torch.no_grad():
        _, w2 = ref(xyz, feats)
        _, g2 = mine(xyz.cuda(), feats.cuda())
    assert_close(g2.cpu().numpy(), w2.numpy(), 1e-4, 2e-5, "SA output (eval)")


def test_feature_heads_and_select_action():
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from oracle import ref_step
    from oracle.detfill import fill_module_
    agent, cfg = make_agent("ddpg_td3_aux.yaml")
    oracle = ref_step.OracleAgent(load_cfg("ddpg_td3_aux.yaml").RL_TRAIN)
    nets = {"policy": agent.policy, "policy_target": agent.policy_target, "critic": agent.critic,
            "critic_target": agent.critic_target, "state_feature_extractor": agent.state_feature_extractor}
    for name, net in nets.items():
        fill_module_(net, name, SEED)
    for name, net in oracle.nets().items():
        fill_module_(net, name, SEED)
    mem = BaseMemory(600, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 600, seed=4)
    batch = sample_valid_batch(mem, 16, np.random.default_rng(2))
    pc = torch.tensor(batch["point_state_batch"])
    fe, ofe = agent.state_feature_extractor, oracle.state_feature_extractor
    # train-mode feature extraction (policy encoder), then eval-mode with the updated running statistics
    fe.train(); ofe.train()
    with torch.no_grad():
        z, _ = fe(pc.cuda(), feature_2=False)
        wz = ofe(pc, value=False)
    assert_close(z.cpu().numpy(), wz.numpy(), 1e-4, 2e-5, "train-mode feature")
    fe.eval(); ofe.eval()
    with torch.no_grad():
        z2, _ = fe(pc.cuda(), feature_2=False)
        wz2 = ofe(pc, value=False)
    assert_close(z2.cpu().numpy(), wz2.numpy(), 1e-4, 2e-5, "eval-mode feature")
    # heads on a plain (B,513) tensor
    state = torch.cat([wz2, torch.tensor(batch["time_batch"])[:, None]], 1)
    with torch.no_grad():
        q1, q2, aux = agent.critic(state.cuda())
        wq1, wq2, waux = oracle.critic(state)
        pi, _, _, paux = agent.policy.sample(state.cuda())
        wpi, wpaux = oracle.policy(state)
    assert_close(q1.cpu().numpy(), wq1.numpy(), 1e-4, 2e-5, "q1")
    assert_close(q2.cpu().numpy(), wq2.numpy(), 1e-4, 2e-5, "q2")
    assert_close(aux.cpu().numpy(), waux.numpy(), 1e-4, 2e-5, "critic aux")
    assert_close(pi.cpu().numpy(), wpi.numpy(), 1e-4, 2e-6, "pi")
    assert_close(paux.cpu().numpy(), wpaux.numpy(), 1e-4, 2e-5, "policy aux")
    # select_action: one cloud, eval mode
    one = batch["point_state_batch"][3]
    action, _, sample, aux_pred = agent.select_action([[one, None]], remain_timestep=7)
    with torch.no_grad():
        wz1 = ofe(torch.tensor(one[None]), value=False)
        wa, wx = oracle.policy(torch.cat([wz1, torch.tensor([[7.0]])], 1))
    assert_close(action, wa[0].numpy(), 1e-4, 2e-6, "select_action")
    assert_close(aux_pred, wx[0].numpy(), 1e-4, 2e-5, "select_action aux")
    # the agent still trains after these inference calls (parameters were re-homed once)
    out = agent.update_parameters(batch, agent.update_step, 0)
    assert np.isfinite(list(out.values())).all()


def test_checkpoint_roundtrip(tmp_path):
    """save_model / load_model (reference core/agent.py:282-431 file layout): a reloaded agent reproduces the
    next update step of the original bit-for-bit (same kernels, same state)."""
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    torch.manual_seed(5)
    a1, cfg = make_agent("ddpg_td3_aux.yaml")
    mem = BaseMemory(600, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 600, seed=4)
    rng = np.random.default_rng(2)
    b0, b1 = sample_valid_batch(mem, 16, rng), sample_valid_batch(mem, 16, rng)
    u = rng.random((16, 6)).astype(np.float32)
    a1.update_parameters(b0, a1.update_step, 0, noise_u=u)
    a1.step_scheduler()
    a1.save_model(a1.update_step, output_dir=str(tmp_path))
    import os
    assert sorted(os.listdir(tmp_path)) == ["DDPG_actor_PandaYCBEnv_latest", "DDPG_critic_PandaYCBEnv_latest",
                                            "DDPG_state_feat_PandaYCBEnv_latest"]
    sd = torch.load(os.path.join(tmp_path, "DDPG_state_feat_PandaYCBEnv_latest"), weights_only=False)
    assert set(sd.keys()) == {"net", "opt", "encoder_opt", "sch", "encoder_sch", "val_encoder_opt", "val_encoder_sch", "step"}
    assert "module.encoder.0.0.mlps.0.0.weight" in sd["net"] and "module.value_encoder.1.4.running_var" in sd["net"]
    torch.manual_seed(6)
    a2, _ = make_agent("ddpg_td3_aux.yaml")
    a2.update_parameters(b0, a2.update_step, 0, noise_u=u)        # builds its runtime with different weights
    assert a2.load_model(str(tmp_path)) == a1.update_step
    # targets are hard-copied from the loaded nets by load_model (reference :386,404): do the same on a1
    from ga_ddpg_amd.core.utils import hard_update
    hard_update(a1.policy_target, a1.policy)
    hard_update(a1.critic_target, a1.critic)
    r1 = a1.update_parameters(b1, a1.update_step, 1, noise_u=u)
    r2 = a2.update_parameters(b1, a2.update_step, 1, noise_u=u)
    # not bit-for-bit: the BatchNorm / weight-gradient sums are built with f64 atomics whose order varies run to run;
    # analytically-zero gradient entries (bias in front of a BatchNorm, weights of dead inputs) are then pure noise
    # and Adam's sign-like first steps move them by +-lr in either run (tools/diag_determinism.py: ~95 of 254 tensors
    # differ between two identical runs, in every kernel configuration).  The losses barely see those weights.
    for k in r1:
        assert_close(r2[k], r1[k], 5e-3, 1e-6, "after reload: " + k)


def test_device_replay_matches_host_sampling():
    """the GPU-resident replay mirror returns, for the same indices, exactly the minibatch of
    BaseMemory.sample (float32 view of it), and an update step fed from it equals one fed from the host batch."""
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.device_replay import DeviceReplay
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.parallel import mask_counts
    from ga_ddpg_amd.runtime import BATCH_KEYS
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer
    from oracle.detfill import fill_module_
    agents = []
    for _ in range(2):
        a, cfg = make_agent("ddpg_td3_aux.yaml")
        for name in ("policy", "policy_target", "critic", "critic_target", "state_feature_extractor"):
            fill_module_(getattr(a, name), name, 7)
        agents.append(a)
    mem = BaseMemory(900, cfg)                                   # float64 clouds, as the reference stores them
    fill_synthetic_buffer(mem, 900, seed=3)
    dmem = DeviceReplay(mem)
    rng_a, rng_b = np.random.default_rng(11), np.random.default_rng(11)
    for it in range(3):
        host = mem.sample(24, rng_a)
        dev = dmem.sample(24, rng_b)
        np.testing.assert_array_equal(dev["batch_idx"], host["batch_idx"])
        for k in BATCH_KEYS:
            np.testing.assert_array_equal(dev[k].cpu().numpy(), np.asarray(host[k], dtype=np.float32).reshape(dev[k].shape), err_msg=k)
        np.testing.assert_array_equal(dev["mask_counts"], mask_counts(host))
    if float(host["return_batch"].max()) > 0 and float(host["expert_flag_batch"].max()) >= 1:
        u = np.random.default_rng(5).random((24, 6)).astype(np.float32)
        r_host = agents[0].update_parameters(host, agents[0].update_step, 0, noise_u=u)
        r_dev = agents[1].update_parameters(dev, agents[1].update_step, 0, noise_u=u)
        for k in r_host:
            assert_close(r_dev[k], r_host[k], 2e-3, 1e-6, k)       # same inputs; run-to-run atomics noise only
    # the lazy form: one gather launch straight into the runtime's input buffers
    lazy = dmem.sample_lazy(24, rng=np.random.default_rng(4))
    ref = mem.sample(24, rng=np.random.default_rng(4))
    rt = agents[1].runtime(24, ref["point_state_batch"].shape[2])
    rt.upload(lazy)
    torch.cuda.synchronize()
    for k in BATCH_KEYS:
        np.testing.assert_array_equal(rt.dbuf[k].cpu().numpy(), np.asarray(ref[k], dtype=np.float32).reshape(rt.dbuf[k].shape), err_msg="lazy " + k)
    np.testing.assert_array_equal(rt.dbuf["time_m1"].cpu().numpy(), np.asarray(ref["time_batch"], dtype=np.float32) - 1.0)
    # online insertion: overwrite a slice on the host, refresh it, sample it back
    mem.action[100:110] = 0.25
    dmem.refresh(100, 110)
    got = dmem.sample(10, batch_idx=np.arange(100, 110))
    assert float((got["action_batch"] - 0.25).abs().max()) == 0.0


def test_device_replay_relabels_onpolicy_goals_like_the_host_path():
    """with self_supervision on a non-expert buffer BaseMemory.sample relabels the goals of the on-policy rows
    (reference core/replay_memory.py:233-249,271-272); the GPU-resident path must hand the update the same goals -- eager
    sample(), the lazy gather, and a prefetched lazy handle alike."""
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.device_replay import DeviceReplay
    from ga_ddpg_amd.core.prefetch import PrefetchSampler
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.runtime import BATCH_KEYS
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer
    agent, cfg = make_agent("ddpg_td3_aux.yaml")
    mem = BaseMemory(700, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 700, seed=5)
    mem.name, mem.self_supervision = "online", True
    rng = np.random.default_rng(2)
    for i in range(700):                                           # proper rigid poses, so the relabelled goals are well defined
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        w, x, y, z = q
        mem.state_pose[i] = np.eye(4)
        mem.state_pose[i][:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
        mem.state_pose[i][:3, 3] = rng.normal(size=3)
    dmem = DeviceReplay(mem)
    idx = mem.draw_indices(24, np.random.default_rng(1))
    host = mem.sample(24, batch_idx=idx)
    assert (mem.expert_flags[idx] == 0).any() and (mem.expert_flags[idx] != 0).any()     # both kinds of rows are in the batch
    assert np.abs(host["goal_batch"] - mem.goal[idx]).max() > 1e-3                       # ... and the relabelling changed some goals
    dev = dmem.sample(24, batch_idx=idx)
    for k in BATCH_KEYS:
        np.testing.assert_array_equal(dev[k].cpu().numpy(), np.asarray(host[k], dtype=np.float32).reshape(dev[k].shape), err_msg=k)
    rt = agent.runtime(24, host["point_state_batch"].shape[2])
    rt.upload(dmem.sample_lazy(24, batch_idx=idx))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rt.dbuf["goal_batch"].cpu().numpy(), np.asarray(host["goal_batch"], dtype=np.float32))
    # handles drawn ahead by a prefetcher and never consumed give their staging sets back (DeviceReplay.release)
    for _ in range(40):
        with PrefetchSampler(dmem, 24, depth=3, rng=np.random.default_rng(0)) as s:
            dmem.release(s.next())                                 # (a batch the caller drops, e.g. after a validity check)
    sets = dmem._stage[("sets", 24)]["items"]
    assert sum(1 for it in sets if it is not None and it["pending"]) <= 2, "leaked staging sets"


def _manifest(obj):
    """structure of a checkpoint object, as oracle/make_golden.py records it for the reference's files"""
    if torch.is_tensor(obj):
        return {"tensor": list(obj.shape), "dtype": str(obj.dtype).replace("torch.", "")}
    if isinstance(obj, dict):
        return {"dict": {str(k): _manifest(v) for k, v in obj.items()}, "int_keys": all(isinstance(k, int) for k in obj) and len(obj) > 0}
    if isinstance(obj, (list, tuple)):
        return {"list": [_manifest(v) for v in obj]}
    return {"scalar": type(obj).__name__}


def _from_manifest(m, rng):
    """a checkpoint object with the manifest's structure and random contents"""
    if "tensor" in m:
        dt = getattr(torch, m["dtype"])
        if dt.is_floating_point:
            return (torch.as_tensor(np.asarray(rng.standard_normal(m["tensor"]), dtype=np.float32)).reshape(m["tensor"]) * 0.05).to(dt)
        return torch.full(m["tensor"], 3, dtype=dt)
    if "dict" in m:
        return {(int(k) if m.get("int_keys") else k): _from_manifest(v, rng) for k, v in m["dict"].items()}
    if "list" in m:
        return [_from_manifest(v, rng) for v in m["list"]]
    return {"int": 3, "float": 0.5, "bool": False, "str": "x", "NoneType": None}[m["scalar"]]


def _diff(a, b, path=""):
    """structural differences between two manifests (scalar types int / float are interchangeable: torch versions differ)"""
    if set(a) - {"int_keys"} != set(b) - {"int_keys"}:
        return ["%s: %s vs %s" % (path, sorted(a), sorted(b))]
    if "tensor" in a:
        return [] if (a["tensor"] == b["tensor"] and a["dtype"] == b["dtype"]) else ["%s: %s %s vs %s %s" % (path, a["tensor"], a["dtype"], b["tensor"], b["dtype"])]
    if "dict" in a:
        out = []
        if set(a["dict"]) != set(b["dict"]):
            out.append("%s: keys differ: only ours %s, only reference %s" % (path, sorted(set(a["dict"]) - set(b["dict"]))[:6], sorted(set(b["dict"]) - set(a["dict"]))[:6]))
        for k in set(a["dict"]) & set(b["dict"]):
            out += _diff(a["dict"][k], b["dict"][k], path + "/" + k)
        return out
    if "list" in a:
        if len(a["list"]) != len(b["list"]):
            return ["%s: list length %d vs %d" % (path, len(a["list"]), len(b["list"]))]
        return [d for i, (x, y) in enumerate(zip(a["list"], b["list"])) for d in _diff(x, y, path + "[%d]" % i)]
    return []


def test_checkpoints_have_the_reference_writers_structure(tmp_path, golden_dir):
    """tests/golden/checkpoint_manifest.json = file names, dict keys, state-dict keys, tensor shapes / dtypes and optimiser
    state structure of a checkpoint set WRITTEN BY THE REFERENCE's Agent.save_model (core/agent.py:282-346) after one update
    (oracle/make_golden.py gen_checkpoint).  (1) our save_model writes exactly that structure; (2) a file set built from the
    manifest -- what a reference run leaves on disk -- is accepted by our load_model: every tensor arrives, the schedulers
    and Adam states are restored, update_step comes from the file."""
    import json
    import os
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    man = json.load(open(os.path.join(golden_dir, "checkpoint_manifest.json")))
    torch.manual_seed(5)
    a1, cfg = make_agent("ddpg_td3_aux.yaml")
    assert (a1.name, a1.env_name) == (man["agent_name"], man["env_name"])
    mem = BaseMemory(600, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 600, seed=4)
    rng = np.random.default_rng(2)
    a1.update_step = 2
    a1.update_parameters(sample_valid_batch(mem, 8, rng), a1.update_step, 0)
    a1.step_scheduler()
    ours_dir = os.path.join(str(tmp_path), "ours")
    a1.save_model(7, output_dir=ours_dir)
    assert sorted(os.listdir(ours_dir)) == sorted(man["files"])
    bad = []
    for f in man["files"]:
        bad += _diff(_manifest(torch.load(os.path.join(ours_dir, f), weights_only=False)), man["files"][f], f)
    assert not bad, "\n".join(bad[:20])
    # (2) a file set as the reference leaves it -> load_model
    ref_dir = os.path.join(str(tmp_path), "ref")
    os.makedirs(ref_dir)
    objs = {}
    for f, m in man["files"].items():
        objs[f] = _from_manifest(m, rng)
        for part in ("opt", "encoder_opt", "val_encoder_opt"):           # valid optimiser hyper-parameters, positive second moments
            if part in objs[f]:
                for st in objs[f][part]["state"].values():
                    st["exp_avg_sq"] = st["exp_avg_sq"].abs()
                first = 0
                for g in objs[f][part]["param_groups"]:
                    g.update(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                             params=list(range(first, first + len(g["params"]))))
                    first += len(g["params"])
        for k, v in objs[f]["net"].items():
            if k.endswith("running_var"):
                objs[f]["net"][k] = v.abs() + 0.5
        for part in ("sch", "encoder_sch", "val_encoder_sch"):
            if part in objs[f]:
                objs[f][part] = a1.policy_scheduler.state_dict()          # a well-formed MultiStepLR state
        torch.save(objs[f], os.path.join(ref_dir, f))
    a2, _ = make_agent("ddpg_td3_aux.yaml")
    assert a2.load_model(ref_dir) == 3                                     # `step` of the state_feat file (the manifest's int -> 3)
    for f, attr in (("DDPG_actor_PandaYCBEnv_latest", "policy"), ("DDPG_critic_PandaYCBEnv_latest", "critic"),
                    ("DDPG_state_feat_PandaYCBEnv_latest", "state_feature_extractor")):
        sd = getattr(a2, attr).state_dict()
        for k, v in objs[f]["net"].items():
            assert torch.equal(sd[k].cpu().to(v.dtype), v), (f, k)
    st = a2.policy_optim.state_dict()["state"]
    ref_st = objs["DDPG_actor_PandaYCBEnv_latest"]["opt"]["state"]
    assert set(st) == set(ref_st) and all(torch.equal(st[i]["exp_avg"].cpu(), ref_st[i]["exp_avg"]) for i in st)


def test_device_mirror_is_built_once_and_follows_writes():
    """train_off_policy's default feeding path keeps ONE HBM mirror per memory object (no re-allocation per call) and
    re-uploads it when the buffer was written in between, so a caller alternating training and data collection never trains on a
    stale snapshot; the rng hook draws the same indices as BaseMemory.sample with that generator."""
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core import train_test_offline as tto
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer
    _, cfg = make_agent("ddpg_td3_aux.yaml")
    mem = BaseMemory(300, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 200, seed=9)
    d1 = tto.device_mirror(mem)
    ptr = d1.point_state.data_ptr()
    assert tto.device_mirror(mem) is d1 and d1.point_state.data_ptr() == ptr
    lo = mem.upper_idx()
    fill_synthetic_buffer(mem, 240, seed=10)                  # the collector wrote: 240 (different) transitions now
    assert mem.upper_idx() > lo
    d2 = tto.device_mirror(mem)
    assert d2 is d1 and d2.point_state.data_ptr() == ptr      # same allocation, refreshed
    hi = mem.upper_idx()
    np.testing.assert_array_equal(d2.point_state[:hi].cpu().numpy(), np.asarray(mem.point_state[:hi], dtype=np.float32))
    np.testing.assert_array_equal(d2.timestep[:hi].cpu().numpy(), np.asarray(mem.timestep[:hi], dtype=np.float32))
    a = d2.sample_lazy(16, rng=np.random.default_rng(4))
    b = mem.draw_indices(16, np.random.default_rng(4))
    a["ready_event"].synchronize()                            # the index upload rides a stream of its own
    np.testing.assert_array_equal(a["idx"].cpu().numpy(), b)
    d2.release(a)
