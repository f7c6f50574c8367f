"""Full update steps through the HIP path against (a) golden vectors produced by the reference's own
DDPG / BC code and (b) the CPU oracle on fresh seeded batches.  Tolerances are the north-star 1e-4
relative on losses / Q-values / actions; gradients and post-step parameters get 5e-4 because
train-mode BatchNorm1d over B=8 rows amplifies float rounding."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import assert_close, assert_close_rows, check_summaries, golden_batch

pytestmark = pytest.mark.gpu
SEED = 1234
SKIP = (".1.0.bias", ".1.3.bias")     # biases in front of a train-mode BN: zero gradient analytically


def _filled_agent(cfg_name, seed):
    from ga_ddpg_amd.api import make_agent
    from oracle.detfill import fill_module_
    agent, cfg = make_agent(cfg_name)
    nets = {"policy": agent.policy, "policy_target": agent.policy_target,
            "state_feature_extractor": agent.state_feature_extractor}
    if hasattr(agent, "critic"):
        nets.update(critic=agent.critic, critic_target=agent.critic_target)
    for name, net in nets.items():
        fill_module_(net, name, seed)
    return agent, nets


def _check_params_after_adam(g, prefix, named, lr_bound, nsteps=1):
    """Post-step parameters.  Adam's first steps are sign-like (lr*g/(|g|+eps)), so coordinates whose
    gradient is float noise move by up to +-lr in either implementation; everything else must agree
    tightly.  Bound: every sampled entry within 2.2*lr, l2 norm within 1e-3, median error tiny."""
    from oracle.detfill import summarize
    n = 0
    for name, t in named:
        ks = prefix + name + "#stats"
        if ks not in g.files:
            continue
        stats, vals = summarize(t)
        gv = g[prefix + name + "#vals"]
        err = np.abs(vals.astype(np.float64) - gv)
        assert err.max() <= 2.2 * lr_bound * nsteps + 1e-6, (name, err.max())
        # (a handful of sign-like entries moves the norm of a SMALL tensor -- mean.bias has six entries -- by up to ~lr: absolute slack
        # that vanishes with the tensor's size)
        small = 2.2 * lr_bound * nsteps * min(1.0, 8.0 / np.sqrt(max(t.numel(), 1)))
        assert abs(stats[2] - g[ks][2]) <= 2e-3 * g[ks][2] + 1e-6 + small, name
        n += 1
    assert n > 0


def _check_step(agent, nets, g, p, kind, s, tight, test=False):
    batch = golden_batch(g, p)
    if kind == "ddpg":
        ret = agent.update_parameters(batch, agent.update_step, s, test=test, noise_u=g[p + "noise_u"])
    else:
        ret = agent.update_parameters(batch, agent.update_step, s)
    agent.step_scheduler(agent.update_step)
    torch.cuda.synchronize()
    # follow-up steps (after an Adam step of the policy, whose policy-step gradients are themselves chaotic, see
    # below): sanity bound only -- same magnitude, finite; what they verify is the step bookkeeping (update gap,
    # learning-rate schedule, running-stat momentum, result keys)
    rt, at = (1e-4, 4e-6) if tight else (1.0, 1e-3)
    policy_step = p + "t/qf1_pi" in g.files
    assert set(ret.keys()) == set(k[len(p + "ret/"):] for k in g.files if k.startswith(p + "ret/"))
    def nclose(a, b, what, extra=1.0):
        # tight steps: entry-wise rt + at; follow-up steps (after an Adam step the float32 trajectories of ANY two
        # implementations separate: torch-f32 vs torch-f64 differ by 1.25e-2 in pi at BC step 1): norm-wise
        if tight:
            assert_close_rows(a, b, rt, extra * at, what)
        else:
            assert_close(a, b, 0.0, rt * np.abs(b).max() + at, what)
    nclose(agent.pi.cpu().numpy(), g[p + "t/pi"], p + "pi")
    nclose(agent.aux_pred.cpu().numpy(), g[p + "t/aux_pred"], p + "aux_pred", 10)
    if kind == "ddpg":
        nclose(agent.qf1.cpu().numpy(), g[p + "t/qf1"], p + "qf1", 10)
        nclose(agent.qf2.cpu().numpy(), g[p + "t/qf2"], p + "qf2", 10)
        yref = g[p + "t/next_q_value"]
        # follow-up steps: the TD target chains encoder -> target policy -> value encoder -> target critic and is the
        # most chaotic tensor of the step: the CPU reference run with 1 / 3 / 8 threads (same code, same machine)
        # moves y by 0.11-0.15 at b1 while pi moves 6e-3 and qf1 3e-4 (measured) -> sanity bound only
        assert_close(agent.next_q_value.cpu().numpy(), yref, 0.0, (1e-4 if tight else 1.0) * np.abs(yref).max() + 2e-5, p + "y")
        nclose(agent.critic_grasp_aux.cpu().numpy(), g[p + "t/critic_grasp_aux"], p + "caux", 10)
    for k, v in ret.items():
        tol = rt if "loss" in k else 5 * rt
        if k in ("actor_critic_loss", "critic_grad") and policy_step:
            # Q(s, pi(s)) is evaluated AFTER the critic / value-encoder Adam step of the same update: Adam's first
            # steps are sign-like, float noise in near-zero gradients moves those weights by +-lr (torch-float32 vs
            # torch-float64 of the reference arithmetic: 2 % on actor_critic_loss); critic_grad on a policy
            # step is max|grad| of the actor term through that updated critic
            tol = 3e-2 if k == "actor_critic_loss" else 1e-1
        assert_close(v, g[p + "ret/" + k], tol if tight else 0.3, 1e-6, p + k)
    which = ["policy", "state_feature_extractor"] + (["critic"] if kind == "ddpg" else [])
    for name in which:
        if not tight:
            break      # follow-up steps: gradients of separated trajectories are not comparable
        named = []
        for n, q in nets[name].named_parameters():
            if p + "end/grad/" + name + "/" + n + "#stats" not in g.files:
                # the reference never produced a gradient there (grad is None): ours must be exactly zero
                assert float(q.grad.abs().max()) == 0.0, (name, n)
                continue
            if policy_step and "value_encoder" in n:
                continue   # reference accumulates a discarded dW there on policy steps; we skip that work
            named.append((n, q.grad))
        loose = policy_step or not tight
        # tight steps: median entry error <= 5e-3 of the tensor scale.  The float32 golden is itself up to 1.1e-3
        # (norm-wise) from the float64 truth on the SA1 tensors and one ReLU flip in the FC layers (B rows) moves
        # every SA1 weight-gradient entry by ~1/B; closeness to the TRUTH is what test_gradient_accuracy_vs_float64 bounds
        check_summaries(g, p + "end/grad/" + name + "/", named, (3e-1 if name == "critic" else 1e-1) if loose else 5e-3, 2e-6, skip=SKIP, normwise=True,
                        l2_rtol=3e-1 if loose else 2e-2, max_rtol=5e-1 if loose else 5e-2)   # policy step: dQ/da through the just-updated critic (Adam chaos): median only
    for name, net in nets.items():
        sd = [(n, t) for n, t in net.state_dict().items() if "num_batches" not in n and not any(x in n for x in SKIP)]
        _check_params_after_adam(g, p + "end/param/" + name + "/", [(n, t) for n, t in sd if "running" not in n], 1e-3, s + 1)
        running = [(n, t) for n, t in sd if "running" in n]
        if running:     # policy steps re-run the value encoder after its Adam step -> looser
            if tight:
                check_summaries(g, p + "end/param/" + name + "/", running, 2e-2 if policy_step else rt, 1e-4)
            else:   # separated trajectories: the batch statistics folded in differ -> norm-wise per tensor
                check_summaries(g, p + "end/param/" + name + "/", running, 2e-2, 1e-4, normwise=True, l2_rtol=5e-2)
    if kind == "ddpg":
        lr = agent.get_lr()
        assert_close([lr["policy_lr"], lr["feature_lr"], lr["value_lr"]], g[p + "lr"], 1e-7, 0, p + "lr")


def test_ddpg_steps_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "ddpg_steps_B32.npz"))
    for run, start, nsteps in (("a", 1, 1), ("b", 2, 2)):
        agent, nets = _filled_agent("ddpg_td3_aux.yaml", SEED)
        agent.update_step = start
        for s in range(nsteps):
            _check_step(agent, nets, g, "%s%d/" % (run, s), "ddpg", s, tight=(s == 0))


def test_ddpg_step_test_mode_vs_reference_golden(golden_dir):
    """update_parameters(..., test=True) (reference core/agent.py:261-280: the update with every online network in eval mode):
    eval-mode BatchNorm in all five encoder passes and its derivative in the three backward passes, against the reference's own run
    (tests/golden/ddpg_steps_test_mode_B32.npz, oracle/make_golden.py gen_ddpg_test_mode) from det-filled parameters and running
    statistics; a policy and a non-policy step; running statistics and batch counters must stay untouched."""
    from oracle.detfill import fill_running_stats_
    g = np.load(os.path.join(golden_dir, "ddpg_steps_test_mode_B32.npz"))
    for run, start in (("e", 1), ("f", 2)):
        agent, nets = _filled_agent("ddpg_td3_aux.yaml", SEED)
        for name, net in nets.items():
            fill_running_stats_(net, name, SEED)
        before = {k: v.clone() for k, v in agent.state_feature_extractor.state_dict().items() if "running" in k or "num_batches" in k}
        assert len(before) >= 60
        agent.update_step = start
        _check_step(agent, nets, g, "%s0/" % run, "ddpg", 0, tight=True, test=True)
        assert not agent.state_feature_extractor.training and agent.test_mode
        for k, v in agent.state_feature_extractor.state_dict().items():
            if k in before:
                assert torch.equal(v.cpu(), before[k].cpu()), "eval-mode step touched " + k
        # ... and the next ordinary step trains again (mode switch, train-mode plans of the same runtime)
        batch = golden_batch(g, "%s0/" % run)
        ret = agent.update_parameters(batch, agent.update_step, 1, noise_u=g["%s0/noise_u" % run])
        assert agent.state_feature_extractor.training and all(np.isfinite(v) for v in ret.values())
        assert int(agent.state_feature_extractor.state_dict()["module.encoder.0.0.mlps.0.1.num_batches_tracked"]) == 2


def test_ddpg_step_asymmetric_action_bounds_vs_reference_golden(golden_dir):
    """GaussianPolicy over an action space with asymmetric bounds (reference core/networks.py:329-337: action_bias = (high + low) / 2):
    pi = tanh(mean) * scale + bias reaches the TD target (target policy), Q(s, pi(s)) and the BC loss of a policy step; against the
    reference's own run with the same space (oracle/make_golden.py gen_ddpg_asym_bounds)"""
    from ga_ddpg_amd.api import make_agent
    from oracle.detfill import AsymTaskSpace6D, fill_module_
    g = np.load(os.path.join(golden_dir, "ddpg_steps_asym_bounds_B32.npz"))
    agent, cfg = make_agent("ddpg_td3_aux.yaml", action_space=AsymTaskSpace6D())
    nets = {"policy": agent.policy, "policy_target": agent.policy_target, "state_feature_extractor": agent.state_feature_extractor,
            "critic": agent.critic, "critic_target": agent.critic_target}
    for name, net in nets.items():
        fill_module_(net, name, SEED)
    assert_close(agent.policy.action_bias.cpu().numpy(), g["g0/action_bias"], 1e-7, 0, "action_bias")
    agent.update_step = 2
    _check_step(agent, nets, g, "g0/", "ddpg", 0, tight=True)
    # the module-level forward (select_action's path) carries the bias too
    feat = torch.randn(8, 513, device="cuda")
    pi, _ = agent.policy.sample(feat)[0], None
    mean = agent.policy.forward(feat)[0]
    want = torch.tanh(mean) * agent.policy.action_scale.to(mean.device) + agent.policy.action_bias.to(mean.device)
    assert_close(pi.cpu().numpy(), want.cpu().numpy(), 1e-6, 1e-7, "policy.sample squashed mean")


@pytest.mark.parametrize("run,start", [("a", 1), ("b", 2)])
def test_gradients_vs_reference_float64(golden_dir, run, start):
    """Gradient accuracy with a reference-held yardstick: tests/golden/ddpg_steps_B32_f64.npz is the REFERENCE's own
    update step evaluated in float64 (oracle/make_golden.py gen_ddpg_f64) on the a0 / b0 inputs; ddpg_steps_B32.npz its
    float32 run.  This is the FREE-RUNNING comparison: each side takes its own ReLU / max-pool decisions, and one
    pre-activation within rounding of zero resolved differently (SA3.l2 of run a0: one of 1024 rows) moves every tensor
    upstream by ~1/rows -- for any two float32 evaluations, the reference's own included on other inputs.  So the floors
    here are those of a tie (median 1.5e-3, worst entry 3e-2 of the tensor's max; 5x the reference-float32 error on the
    policy step, where the reference's own float32 run is already 3e-3 from its float64 one); the arithmetic itself is held to
    float32-of-torch accuracy (~1e-6) by tests/test_gpu_forced_decisions.py, where the decisions are imposed.
    Skipped: biases in front of a train-mode BatchNorm (analytically zero: both sides hold rounding noise) and, on the
    policy step, the value encoder (the reference accumulates a gradient there that it discards; we skip that work)."""
    from tests.helpers import grad_accuracy_rows
    g32 = np.load(os.path.join(golden_dir, "ddpg_steps_B32.npz"))
    g64 = np.load(os.path.join(golden_dir, "ddpg_steps_B32_f64.npz"))
    agent, nets = _filled_agent("ddpg_td3_aux.yaml", SEED)
    agent.update_step = start
    p = "%s0/" % run
    ret = agent.update_parameters(golden_batch(g32, p), agent.update_step, 0, noise_u=g32[p + "noise_u"])
    torch.cuda.synchronize()
    policy_step = start % 2 == 0
    for k in ("critic_loss", "critic_grasp_aux_loss", "bc_loss", "policy_grasp_aux_loss"):
        assert_close(ret[k], float(g64[p + "ret/" + k]), 1e-4, 1e-6, k)           # losses against the float64 reference
    assert_close(agent.qf1.cpu().numpy(), g64[p + "t/qf1"], 0.0, 1e-4 * np.abs(g64[p + "t/qf1"]).max(), "qf1 vs float64")
    assert_close(agent.pi.cpu().numpy(), g64[p + "t/pi"], 0.0, 1e-4 * np.abs(g64[p + "t/pi"]).max(), "pi vs float64")
    rows, bad = [], []
    for name in ("policy", "state_feature_extractor", "critic"):
        named = [(n, q.grad) for n, q in nets[name].named_parameters()
                 if not (policy_step and ("value_encoder" in n or name == "critic"))]
        rows += [(name + "/" + r[0],) + r[1:] for r in
                 grad_accuracy_rows(g32, g64, p + "end/grad/" + name + "/", named, skip=SKIP)]
    assert len(rows) > (30 if policy_step else 80)
    lines = ["%-72s %10s %10s %10s %10s %10s" % ("tensor (run %s0, B=32)" % run, "max|ref64|", "hip med", "hip max", "ref32 med", "ref32 max")]
    for name, scale, hm, hx, rm, rx in sorted(rows, key=lambda r: -r[2] / max(3 * r[4], 1e-4)):
        lines.append("%-72s %10.3e %10.2e %10.2e %10.2e %10.2e" % (name, scale, hm, hx, rm, rx))
        # policy step: the median floor is the reference's own float32-vs-float64 level there (3e-3 on the SA1 tensors): which
        # ties fall which way changes with the kernels' summation order and, through the f64 atomics, from run to run (one FC
        # weight tensor sat at 1.62e-3 against a 1.59e-3 limit in one of two runs of the same build)
        med_floor = 3e-3 if policy_step else 1.5e-3
        if hm > max(5 * rm, med_floor) or hx > max(5 * rx, 1e-1 if policy_step else 3e-2):
            bad.append(lines[-1])
    lines.append("violations of  hip med <= max(5 ref32 med, %.1e)  and  hip max <= max(5 ref32 max, %.0e): %d of %d" %
                 (med_floor, 1e-1 if policy_step else 3e-2, len(bad), len(rows)))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        open(os.path.join(out_dir, "grad_accuracy_%s0.txt" % run), "w").write("\n".join(lines) + "\n")
    assert not bad, "\n".join([lines[0]] + bad)


def test_bc_steps_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "bc_steps_B32.npz"))
    agent, nets = _filled_agent("bc_dagger_aux.yaml", SEED + 1)
    for s in range(2):
        _check_step(agent, nets, g, "a%d/" % s, "bc", s, tight=(s == 0))


def test_steps_vs_oracle_fresh_batches():
    """B=32 DDPG steps on fresh synthetic batches: HIP agent and CPU oracle advance side by side."""
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from oracle import ref_step
    from oracle.detfill import fill_module_
    agent, nets = _filled_agent("ddpg_td3_aux.yaml", 77)
    c = load_cfg("ddpg_td3_aux.yaml")
    oracle = ref_step.OracleAgent(c.RL_TRAIN)
    for name, net in oracle.nets().items():
        fill_module_(net, name, 77)
    mem = BaseMemory(1500, c, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 1500, seed=5)
    rng = np.random.default_rng(9)
    for s in range(2):
        batch = sample_valid_batch(mem, 32, rng)
        u = rng.random((32, 6)).astype(np.float32)
        got = agent.update_parameters(batch, agent.update_step, s, noise_u=u)
        want = oracle.update_parameters(batch, noise_u=u)
        # after ONE Adam step float32 trajectories separate: torch-float32 vs torch-float64 of the same
        # code differ by 2 % in actor_critic_loss and 6.5e-3 in Q at step 1 (measured)
        rt = 1e-4 if s == 0 else 5e-2
        for k in want:
            tol = rt if "loss" in k else 5 * rt
            assert_close(got[k], want[k], tol, 1e-6, "step %d %s" % (s, k))
        q_ref, pi_ref = oracle.dbg["q1"].numpy(), oracle.dbg["pi"].numpy()
        assert_close(agent.qf1.cpu().numpy(), q_ref, 0.0, rt * np.abs(q_ref).max() + 2e-5, "q1")
        assert_close(agent.pi.cpu().numpy(), pi_ref, 0.0, rt * np.abs(pi_ref).max() + 2e-6, "pi")


@pytest.mark.parametrize("policy_step", [False, True])
def test_gradient_accuracy_vs_float64(policy_step):
    """The actor-critic step's gradients are ill-conditioned in float32 (torch's own float32 and
    float64 evaluations of the reference arithmetic differ by 1e-2 norm-wise on some tensors, see
    DESIGN.md 6), so a fixed 1e-4 bound against a float32 reference is not meaningful there.  Yardstick:
    the HIP gradient must be as close to the float64 truth as torch-float32 is (factor 3 + 1e-4)."""
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from oracle import ref_step
    from oracle.detfill import fill_module_
    B = 64
    c = load_cfg("ddpg_td3_aux.yaml")
    mem = BaseMemory(3000, c, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 3000, seed=5)
    rng = np.random.default_rng(int(os.environ.get("GAD_DIAG_SEED", "1")))      # env: A/B diagnostics over batches
    batch = sample_valid_batch(mem, B, rng)
    u = rng.random((B, 6)).astype(np.float32)

    def oracle_grads(dtype):
        o = ref_step.OracleAgent(c.RL_TRAIN)
        for n, net in o.nets().items():
            fill_module_(net, n, 3)
        o.to_dtype(dtype)
        o.update_step = 2 if policy_step else 1
        for opt in (o.val_encoder_optim, o.critic_optim):
            opt.param_groups[0]["lr"] = 0.0          # same critic in both phases: isolates the arithmetic
        out = o.update_ddpg(batch, noise_u=u)
        return out, {nn + "/" + n: p.grad.double() for nn, net in o.nets().items()
                     for n, p in net.named_parameters() if p.grad is not None}
    out32, g32 = oracle_grads(torch.float32)
    out64, g64 = oracle_grads(torch.float64)
    agent, nets = _filled_agent("ddpg_td3_aux.yaml", 3)
    agent.update_step = 2 if policy_step else 1
    for opt in (agent.state_feat_val_encoder_optim, agent.critic_optim):
        opt.param_groups[0]["lr"] = 0.0
    got = agent.update_parameters(batch, agent.update_step, 0, noise_u=u)
    torch.cuda.synchronize()
    for k in ("critic_loss", "critic_grasp_aux_loss", "bc_loss", "policy_grasp_aux_loss", "actor_critic_loss"):
        assert_close(got[k], out64[k], 1e-4, 1e-6, k)
    worst = 0.0
    for key, ref in g64.items():
        nn, n = key.split("/", 1)
        if any(x in n for x in SKIP) or (policy_step and "value_encoder" in n):
            continue
        mine = dict(nets[nn].named_parameters())[n].grad.double().cpu()
        scale = float(ref.abs().max()) + 1e-30
        # median entry error comparable to torch-float32's (x10 + 5e-3; policy step 3e-2, where torch-float32
        # itself is 4e-2 from float64 on policy/mean.bias); max capped at 5e-2 or 3x torch-float32's own max error.  A kink flip in a layer with
        # few rows (FC: B rows, SA3: 32B rows) shifts every upstream entry by ~1/rows, so tighter is not
        # attainable by ANY float32 implementation (helpers.check_summaries)
        e_hip = float((mine - ref).abs().median()) / scale
        e_t32 = float((g32[key] - ref).abs().median()) / scale
        allow = 10 * e_t32 + (3e-2 if policy_step else 5e-3)
        worst = max(worst, e_hip / allow)
        assert e_hip <= allow, (key, e_hip, e_t32)
        m_t32 = float((g32[key] - ref).abs().max()) / scale
        # min(Q1,Q2) / ReLU / max-pool flips of single samples move single entries by ~1/B of the tensor scale each
        # (B=64 here).  Over 4 batches x {tiled, streaming} forward kernels the worst entry seen was 1.1e-1 and the
        # outcome depends on the batch, not on the kernel (GAD_DIAG_SEED / GAD_OPT_fwd_stream A/B)
        assert float((mine - ref).abs().max()) / scale <= max(1.5e-1, 3.0 * m_t32), (key, m_t32)
    print("worst HIP-error / allowance ratio:", worst)


def test_bc_step_config0_batch64_vs_oracle():
    """BASELINE configs[0]: offline BC update (bc_aux_dagger), batch 64, 1024-point clouds: one step from identical
    parameters against the CPU oracle -- losses 1e-4, actions / aux poses 1e-4 of the tensor scale, policy and encoder
    gradients norm-wise (helpers.check... policy: same kink caveat as the DDPG steps)."""
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from oracle import ref_step
    from oracle.detfill import fill_module_
    agent, nets = _filled_agent("bc_dagger_aux.yaml", 21)
    c = load_cfg("bc_dagger_aux.yaml")
    oracle = ref_step.OracleAgent(c.RL_TRAIN)
    for name, net in oracle.nets().items():
        fill_module_(net, name, 21)
    mem = BaseMemory(1500, c, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 1500, seed=8)
    batch = sample_valid_batch(mem, 64, np.random.default_rng(2))
    got = agent.update_parameters(batch, agent.update_step, 0)
    want = oracle.update_parameters(batch)
    torch.cuda.synchronize()
    assert set(got.keys()) == set(want.keys())
    for k in ("bc_loss", "policy_grasp_aux_loss"):
        assert_close(got[k], want[k], 1e-4, 1e-6, k)
    for k in ("critic_loss", "critic_grasp_aux_loss", "actor_critic_loss"):
        assert got[k] == 0.0 and want[k] == 0.0, k                      # BC has no critic
    pi_ref, aux_ref = oracle.dbg["pi"].numpy(), oracle.dbg["aux_pred"].numpy()
    assert_close(agent.pi.cpu().numpy(), pi_ref, 1e-4, 1e-5 * np.abs(pi_ref).max(), "pi")
    assert_close(agent.aux_pred.cpu().numpy(), aux_ref, 1e-4, 1e-5 * np.abs(aux_ref).max(), "aux_pred")
    on = {n + "/" + k: p.grad for n, net in oracle.nets().items() for k, p in net.named_parameters() if p.grad is not None}
    worst = 0.0
    for name in ("policy", "state_feature_extractor"):
        for k, p in nets[name].named_parameters():
            ref = on.get(name + "/" + k)
            if ref is None or any(x in k for x in SKIP):
                continue
            scale = float(ref.abs().max()) + 1e-30
            err = (p.grad.cpu() - ref).abs()
            worst = max(worst, float(err.median()) / scale)
            assert float(err.median()) / scale <= 5e-3 and float(err.max()) / scale <= 1.5e-1, (name, k)
    print("worst median gradient error / scale:", worst)


def test_ddpg_step_config4_batch512_properties():
    """BASELINE configs[4] runs B=512 per GPU: the fused step at that size -- finite results, the 11 keys, row
    bookkeeping, and scale consistency with the same clouds at B=256 (losses are means: same order of magnitude)."""
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from oracle.detfill import fill_module_
    outs = {}
    for B in (512, 256):
        agent, cfg = make_agent("ddpg_td3_aux.yaml")
        for name in ("policy", "policy_target", "critic", "critic_target", "state_feature_extractor"):
            fill_module_(getattr(agent, name), name, 5)
        mem = BaseMemory(3000, cfg, point_dtype=np.float32)
        fill_synthetic_buffer(mem, 3000, seed=6)
        rng = np.random.default_rng(4)
        r = None
        for s in range(2):                                        # a policy step and a non-policy step
            batch = sample_valid_batch(mem, B, rng)
            r = agent.update_parameters(batch, agent.update_step, s, noise_u=rng.random((B, 6)).astype(np.float32))
            assert len(r) == 11 and all(np.isfinite(v) for v in r.values()), r
            assert r["train_batch_size"] == 0.0                    # a slot the reference zeroes and never fills (core/agent.py:213-216)
        outs[B] = r
        rt = agent._rt
        assert int(rt.geo.rows[0]["n"].item()) <= B * 32 * 64 and int(rt.geo.rows[2]["n"].item()) == B * 32
    for k in ("critic_loss", "bc_loss", "policy_grasp_aux_loss", "critic_grasp_aux_loss"):
        assert 0.3 < outs[512][k] / outs[256][k] < 3.0, (k, outs[512][k], outs[256][k])


@pytest.mark.parametrize("B,NP", [(10, 512), (37, 1024), (200, 256)])
def test_ragged_batch_and_cloud_sizes_vs_oracle(B, NP):
    """batch sizes that are no multiple of the 32-row MFMA tile and cloud sizes other than 1024: one DDPG step from
    identical parameters against the CPU oracle (exercises the clamped / masked edges of every kernel family: at B=10
    the SA1 layers take the tiled path, at B=37 and 200 the streaming one with a ragged last slab)."""
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from oracle import ref_step
    from oracle.detfill import fill_module_
    agent, cfg = make_agent("ddpg_td3_aux.yaml")
    cfg.RL_TRAIN.uniform_num_pts = NP
    c2 = load_cfg("ddpg_td3_aux.yaml")
    c2.RL_TRAIN.uniform_num_pts = NP
    oracle = ref_step.OracleAgent(c2.RL_TRAIN)
    for name in ("policy", "policy_target", "critic", "critic_target", "state_feature_extractor"):
        fill_module_(getattr(agent, name), name, 3)
    for name, net in oracle.nets().items():
        fill_module_(net, name, 3)
    mem = BaseMemory(600, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 600, seed=2)
    rng = np.random.default_rng(1)
    batch = sample_valid_batch(mem, B, rng)
    u = rng.random((B, 6)).astype(np.float32)
    got = agent.update_parameters(batch, agent.update_step, 0, noise_u=u)
    want = oracle.update_parameters(batch, noise_u=u)
    # BatchNorm1d over only B rows amplifies rounding at small B: 1e-4 from B = 32 up, 1e-3 below
    tol = 1e-4 if B >= 32 else 1e-3
    for k in ("critic_loss", "critic_grasp_aux_loss", "bc_loss", "policy_grasp_aux_loss"):
        assert_close(got[k], want[k], tol, 1e-6, k)
    q_ref = oracle.dbg["q1"].numpy()
    assert_close(agent.qf1.cpu().numpy(), q_ref, 0.0, tol * np.abs(q_ref).max() + 2e-6, "q1")


def test_ddpg_step_B256_vs_oracle():
    """The benchmark's own size (BASELINE configs[1]: B = 256, N = 1024): one DDPG update from identical parameters against the
    CPU oracle on the same minibatch and target noise, a step with the actor-critic term (all 4.5 encoder passes and both
    critic evaluations take part).  Losses to the north star's 1e-4 against the float32 oracle; Q1 / Q2, the TD target,
    actions and the aux pose against the oracle evaluated in FLOAT64, with the float32 oracle's own distance from it as the
    yardstick: ours <= max(1e-4, 2 x torch-float32) of the tensor's scale (the TD target runs through four chained
    train-mode BatchNorm networks; two float32 evaluations of it sit ~1e-4 apart at this size).
    (An oracle step takes ~20 s on the box's host cores; bench.py times the same call as its cpu_baseline.)"""
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from oracle import ref_step
    from oracle.detfill import fill_module_
    B = 256
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    c = load_cfg("ddpg_td3_aux.yaml")
    mem = BaseMemory(3000, c, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 3000, seed=6)
    rng = np.random.default_rng(12)
    batch = sample_valid_batch(mem, B, rng)
    u = rng.random((B, 6)).astype(np.float32)
    agent, nets = _filled_agent("ddpg_td3_aux.yaml", 29)
    agent.update_step = 2                                  # policy_update_gap 2: the actor-critic term is on
    got = agent.update_parameters(batch, agent.update_step, 0, noise_u=u)
    torch.cuda.synchronize()

    def run(dtype):
        o = ref_step.OracleAgent(c.RL_TRAIN)
        for name, net in o.nets().items():
            fill_module_(net, name, 29)
        o.to_dtype(dtype)
        o.update_step = 2
        return o.update_ddpg(batch, noise_u=u), {k: o.dbg[k].double().numpy() for k in ("q1", "q2", "y", "pi", "aux_pred")}
    want, t32 = run(torch.float32)
    _, t64 = run(torch.float64)
    for k in ("critic_loss", "critic_grasp_aux_loss", "bc_loss", "policy_grasp_aux_loss"):
        assert_close(got[k], want[k], 1e-4, 1e-6, k)
    # evaluated through the critic AFTER its Adam step of the same update (sign-like first step: DESIGN.md 6)
    assert_close(got["actor_critic_loss"], want["actor_critic_loss"], 5e-2, 1e-4, "actor_critic_loss")
    lines = []
    for name, ours, k in (("q1", agent.qf1, "q1"), ("q2", agent.qf2, "q2"), ("td target", agent.next_q_value, "y"),
                          ("pi", agent.pi, "pi"), ("aux_pred", agent.aux_pred, "aux_pred")):
        ref = t64[k]
        scale = np.abs(ref).max()
        e_hip = np.abs(ours.cpu().double().numpy().reshape(ref.shape) - ref).max() / scale
        e_f32 = np.abs(t32[k] - ref).max() / scale
        lines.append("%-10s max error / scale vs float64: hip %.2e   torch-float32 %.2e" % (name, e_hip, e_f32))
        assert e_hip <= max(1e-4, 2.0 * e_f32), lines[-1]
    print("\n".join(lines))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "step_B256_vs_oracle.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


def test_overlapped_schedule_equals_serial_schedule():
    """the step runs on five HIP streams (value pass, target chain, actor pass, two weight-gradient lanes) with a host
    enqueue order chosen for the critical path; with every fork folded onto one stream (engine.SERIAL) the same plans run
    strictly in program order.  One step from identical parameters, without (update_step 1) and with (update_step 2) the
    actor-critic term: same losses / Q / actions / critic gradient up to the summation-order noise of the f64 atomics,
    and the same running BatchNorm statistics -- their update ORDER between the passes that share a network (value pass,
    target chain, actor-critic pass) is what the schedule has to preserve."""
    from ga_ddpg_amd import engine
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    c = load_cfg("ddpg_td3_aux.yaml")
    mem = BaseMemory(1500, c, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 1500, seed=6)
    for start in (1, 2):
        out = {}
        for serial in (False, True):
            agent, nets = _filled_agent("ddpg_td3_aux.yaml", 91)
            agent.update_step = start
            rng = np.random.default_rng(10 + start)
            batch = sample_valid_batch(mem, 64, rng)
            u = rng.random((64, 6)).astype(np.float32)
            engine.SERIAL = serial
            try:
                res = agent.update_parameters(batch, agent.update_step, 0, noise_u=u)
                torch.cuda.synchronize()
            finally:
                engine.SERIAL = False
            bufs = {n: b.detach().cpu().numpy().copy() for n, b in agent.state_feature_extractor.named_buffers()
                    if "running" in n}
            out[serial] = (res, agent.qf1.cpu().numpy().copy(), agent.pi.cpu().numpy().copy(), bufs)
        (r0, q0, p0, b0), (r1, q1, p1, b1) = out[False], out[True]
        for k in r0:
            # actor_critic_loss is evaluated after the critic's Adam step of the same update (sign-like first step:
            # gradient noise moves parameters by +-lr): looser, like everywhere else in this file
            tol = 2e-2 if k == "actor_critic_loss" else 1e-5
            assert_close(r0[k], r1[k], tol, 1e-7, "update_step %d %s" % (start, k))
        assert_close(q0, q1, 0.0, 1e-5 * np.abs(q1).max(), "q1")
        assert_close(p0, p1, 0.0, 1e-5 * np.abs(p1).max(), "pi")
        assert len(b0) >= 20
        venc = [n for n in b0 if "value_encoder" in n]
        for n in b0:
            # value_encoder sees a third pass AFTER the critic's Adam step on steps with the actor-critic term
            tol = 2e-3 if (start == 2 and n in venc) else 1e-5
            assert_close(b0[n], b1[n], tol, tol * max(1.0, float(np.abs(b1[n]).max())), n)
