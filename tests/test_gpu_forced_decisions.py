"""Gradients of a whole update step against the float64 oracle with the ReLU / max-pool decisions of the HIP pass imposed
on the oracle (tests/kink_forcing.py): the tight, tie-proof version of the gradient parity check."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu
SKIP = (".1.0.bias", ".1.3.bias")          # biases in front of a train-mode BatchNorm: analytically zero gradient


@pytest.mark.parametrize("B,seed,policy_step", [(32, 1, False), (64, 2, False), (32, 3, True), (256, 4, False)])
def test_step_gradients_with_forced_decisions(B, seed, policy_step):
    """B = 256 is the bench size: there the row counts route the layers to other kernels than at B <= 64
    (wide tiles instead of 64 x 64 tiles for SA2 / SA3, the fused streaming backward for SA1) -- every gradient tensor is held
    to the same criterion against the float64 oracle (two ~30 - 60 s CPU steps: float64 and float32 with the HIP decisions)."""
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from oracle import ref_step
    from oracle.detfill import fill_module_
    from tests.kink_forcing import decisions_from_slot, forced_forward
    from tests.test_gpu_step import _filled_agent
    c = load_cfg("ddpg_td3_aux.yaml")
    mem = BaseMemory(2000, c, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 2000, seed=5 + seed)
    rng = np.random.default_rng(seed)
    batch = sample_valid_batch(mem, B, rng)
    u = rng.random((B, 6)).astype(np.float32)
    agent, nets = _filled_agent("ddpg_td3_aux.yaml", 3)
    agent.update_step = 1                                  # no actor-critic term: slot_v still holds the critic-phase value pass
    got = agent.update_parameters(batch, agent.update_step, 0, noise_u=u)
    torch.cuda.synchronize()
    rt = agent._rt
    # value encoder call 0 = the critic phase's value pass; encoder call 1 = the actor phase's policy pass (call 0: TD target)
    dec = {("value", 0): decisions_from_slot(rt.venc, rt.slot_v), ("policy", 1): decisions_from_slot(rt.enc, rt.slot_p)}
    calls = {"value": 2, "policy": 2}
    if policy_step:
        # the critic phase is the same computation on either kind of step: its decisions come from the run above; a twin
        # agent then takes the policy step, whose slots end up holding the policy pass and the Q(s, pi(s)) value pass
        agent, nets = _filled_agent("ddpg_td3_aux.yaml", 3)
        agent.update_step = 2
        got = agent.update_parameters(batch, agent.update_step, 0, noise_u=u)
        torch.cuda.synchronize()
        rt = agent._rt
        dec[("policy", 1)] = decisions_from_slot(rt.enc, rt.slot_p)
        dec[("value", 2)] = decisions_from_slot(rt.venc, rt.slot_v)
        calls = {"value": 3, "policy": 2}
    step = 2 if policy_step else 1

    def oracle(dtype):
        o = ref_step.OracleAgent(c.RL_TRAIN)
        for n, net in o.nets().items():
            fill_module_(net, n, 3)
        o.to_dtype(dtype)
        o.update_step = step
        with forced_forward(o.state_feature_extractor.module, dec) as ff:
            out = o.update_ddpg(batch, noise_u=u)
        assert ff.calls == calls
        return out, {nn + "/" + n: p.grad.double() for nn, net in o.nets().items()
                     for n, p in net.named_parameters() if p.grad is not None}, o
    out64, g64, o64 = oracle(torch.float64)
    out32, g32, o32 = oracle(torch.float32)
    for k in ("critic_loss", "critic_grasp_aux_loss", "bc_loss", "policy_grasp_aux_loss"):
        assert_close(got[k], out64[k], 2e-5, 1e-7, k)
    if policy_step:      # a mean of cancelling Q values through the critic updated in this very step
        assert_close(got["actor_critic_loss"], out64["actor_critic_loss"], 2e-4, 1e-6, "actor_critic_loss")
    d, d32 = o64.dbg, o32.dbg
    for mine, key, what in ((agent.qf1, "q1", "qf1"), (agent.qf2, "q2", "qf2"), (agent.next_q_value, "y", "td target"),
                            (agent.pi, "pi", "pi"), (agent.aux_pred, "aux_pred", "aux_pred")):
        ref = d[key].numpy()
        scale = np.abs(ref).max()
        e32 = np.abs(d32[key].double().numpy() - ref).max() / scale      # torch float32's own distance from float64
        eh = np.abs(mine.cpu().numpy() - ref).max() / scale
        print("%-10s max err / max|ref|: hip %.2e   oracle-f32 %.2e" % (what, eh, e32))
        assert eh <= max(3 * e32, 2e-5), (what, eh, e32)
    lines = ["%-64s %10s %10s %10s %10s %10s" % ("tensor (B=%d%s, forced decisions)" % (B, ", policy step" if policy_step else ""), "max|f64|", "hip med", "hip max", "f32 med", "f32 max")]
    bad = []
    for key, ref in sorted(g64.items()):
        nn, n = key.split("/", 1)
        if any(x in n for x in SKIP) or (policy_step and (nn == "critic" or "value_encoder" in n)):
            continue           # policy step: the reference leaves a gradient it discards on the value side; we skip that work
        mine = dict(nets[nn].named_parameters())[n].grad.double().cpu()
        scale = float(ref.abs().max()) + 1e-300
        eh, e3 = (mine - ref).abs() / scale, (g32[key] - ref).abs() / scale
        row = (float(eh.median()), float(eh.max()), float(e3.median()), float(e3.max()))
        lines.append("%-64s %10.3e %10.2e %10.2e %10.2e %10.2e" % ((key, scale) + row))
        # as accurate as torch's own float32 evaluation of the same smooth function (x3), floors at float32 resolution of a
        # sum of ~1e5 cancelling terms
        if row[0] > max(3 * row[2], 2e-6) or row[1] > max(3 * row[3], 1e-4):
            bad.append(lines[-1])
    lines.append("violations of  hip med <= max(3 f32 med, 2e-6)  and  hip max <= max(3 f32 max, 1e-4): %d of %d" % (len(bad), len(lines) - 1))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        from ga_ddpg_amd import hip as _hip
        mode = "_split" if _hip.get_option("mfma_split") else "_f32mfma"          # (both arithmetic modes: tests/conftest.py BOTH_MODES)
        open(os.path.join(out_dir, "grad_accuracy_forced_B%d%s%s.txt" % (B, "_policy_step" if policy_step else "", mode)), "w").write("\n".join(lines) + "\n")
    assert len(lines) > (35 if policy_step else 90)
    assert not bad, "\n".join([lines[0]] + bad)


def test_bc_step_gradients_with_forced_decisions():
    """the behaviour-cloning step of BASELINE configs[0] (B = 64): one policy pass forward + backward, same tight criterion"""
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from oracle import ref_step
    from oracle.detfill import fill_module_
    from tests.kink_forcing import decisions_from_slot, forced_forward
    from tests.test_gpu_step import _filled_agent
    B = 64
    c = load_cfg("bc_dagger_aux.yaml")
    mem = BaseMemory(2000, c, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 2000, seed=11)
    batch = sample_valid_batch(mem, B, np.random.default_rng(4))
    agent, nets = _filled_agent("bc_dagger_aux.yaml", 3)
    got = agent.update_parameters(batch, agent.update_step, 0)
    torch.cuda.synchronize()
    rt = agent._rt
    dec = {("policy", 0): decisions_from_slot(rt.enc, rt.slot_p)}

    def oracle(dtype):
        o = ref_step.OracleAgent(c.RL_TRAIN)
        for n, net in o.nets().items():
            fill_module_(net, n, 3)
        o.to_dtype(dtype)
        with forced_forward(o.state_feature_extractor.module, dec) as ff:
            out = o.update_bc(batch)
        assert ff.calls == {"value": 0, "policy": 1}
        return out, {nn + "/" + n: p.grad.double() for nn, net in o.nets().items()
                     for n, p in net.named_parameters() if p.grad is not None}, o
    out64, g64, o64 = oracle(torch.float64)
    out32, g32, o32 = oracle(torch.float32)
    for k in ("bc_loss", "policy_grasp_aux_loss"):
        assert_close(got[k], out64[k], 2e-5, 1e-7, k)
    for mine, key in ((agent.pi, "pi"), (agent.aux_pred, "aux_pred")):
        ref = o64.dbg[key].numpy()
        scale = np.abs(ref).max()
        eh = np.abs(mine.cpu().numpy() - ref).max() / scale
        e32 = np.abs(o32.dbg[key].double().numpy() - ref).max() / scale
        assert eh <= max(3 * e32, 2e-5), (key, eh, e32)
    lines = ["%-64s %10s %10s %10s %10s %10s" % ("tensor (BC step, B=64, forced decisions)", "max|f64|", "hip med", "hip max", "f32 med", "f32 max")]
    bad = []
    for key, ref in sorted(g64.items()):
        nn, n = key.split("/", 1)
        if any(x in n for x in SKIP):
            continue
        mine = dict(nets[nn].named_parameters())[n].grad.double().cpu()
        scale = float(ref.abs().max()) + 1e-300
        eh, e3 = (mine - ref).abs() / scale, (g32[key] - ref).abs() / scale
        row = (float(eh.median()), float(eh.max()), float(e3.median()), float(e3.max()))
        lines.append("%-64s %10.3e %10.2e %10.2e %10.2e %10.2e" % ((key, scale) + row))
        if row[0] > max(3 * row[2], 2e-6) or row[1] > max(3 * row[3], 1e-4):
            bad.append(lines[-1])
    lines.append("violations of  hip med <= max(3 f32 med, 2e-6)  and  hip max <= max(3 f32 max, 1e-4): %d of %d" % (len(bad), len(lines) - 1))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        from ga_ddpg_amd import hip as _hip
        mode = "_split" if _hip.get_option("mfma_split") else "_f32mfma"
        open(os.path.join(out_dir, "grad_accuracy_forced_BC_B64%s.txt" % mode), "w").write("\n".join(lines) + "\n")
    assert len(lines) > 30
    assert not bad, "\n".join([lines[0]] + bad)


@pytest.mark.parametrize("B,seed", [(32, 1), (64, 2)])
def test_hip_decisions_agree_with_the_free_running_oracle(B, seed):
    """The forced-decision gate above takes its ReLU masks and max-pool winners FROM the HIP pass, so a wrong mask or a
    non-maximal winner would be followed, not caught.  Here the float64 oracle runs FREE on the same step and takes its own
    decisions: the two may differ only where the pre-activation (winner's margin) sits within rounding of the kink --
    fraction of differing decisions <= 1e-5 (ReLU) / 1e-4 (pool), every differing pre-activation within 1e-5 of zero on its
    channel's scale, every differing winner within 1e-5 (relative) of the oracle's maximum."""
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from oracle import ref_step
    from oracle.detfill import fill_module_
    from tests.kink_forcing import decision_differences, decisions_from_slot, recording_forward
    from tests.test_gpu_step import _filled_agent
    c = load_cfg("ddpg_td3_aux.yaml")
    mem = BaseMemory(2000, c, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 2000, seed=5 + seed)
    rng = np.random.default_rng(seed)
    batch = sample_valid_batch(mem, B, rng)
    u = rng.random((B, 6)).astype(np.float32)
    agent, nets = _filled_agent("ddpg_td3_aux.yaml", 3)
    agent.update_step = 1
    agent.update_parameters(batch, agent.update_step, 0, noise_u=u)
    torch.cuda.synchronize()
    rt = agent._rt
    dec = {("value", 0): decisions_from_slot(rt.venc, rt.slot_v), ("policy", 1): decisions_from_slot(rt.enc, rt.slot_p)}
    o = ref_step.OracleAgent(c.RL_TRAIN)
    for n, net in o.nets().items():
        fill_module_(net, n, 3)
    o.to_dtype(torch.float64)
    o.update_step = 1
    with recording_forward(o.state_feature_extractor.module) as rf:
        o.update_ddpg(batch, noise_u=u)
    lines = []
    for key in dec:
        r = decision_differences(dec[key], rf.records[key])
        lines.append("%s pass %d: ReLU %d of %d decisions differ (largest |pre-activation| / scale %.2e); pool %d of %d winners "
                     "differ (largest relative gap %.2e) [%s]" % (key[0], key[1], r["relu"]["n_diff"], r["relu"]["n"], r["relu"]["worst"],
                                                              r["pool"]["n_diff"], r["pool"]["n"], r["pool"]["worst"], r["pool"]["detail"]))
        assert r["relu"]["n_diff"] <= 1e-5 * r["relu"]["n"] and r["relu"]["worst"] <= 1e-5, lines[-1]
        assert r["pool"]["n_diff"] <= 1e-4 * r["pool"]["n"] and r["pool"]["worst"] <= 1e-5, lines[-1]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "decision_differences_B%d.txt" % B), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
