"""Reference-surface items around the update step, on the GPU: GaussianPolicy.forward / sample (log-prob,
reparameterised draw), the full select_action tuple, the offline caller loop (train_off_policy) with checkpoints,
resume on a fresh process and migrate_model."""
import os
import shutil

import numpy as np
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu
SEED = 1234


def test_gaussian_policy_forward_and_sample_vs_reference_golden(golden_dir):
    """tests/golden/heads.npz holds the reference's own GaussianPolicy.forward / .sample outputs (rsample draw injected):
    raw mean, clamped log_std, squashed mean, log_prob, action, extra -- for the aux head (extra_pred_dim 7) and the plain
    one (extra_pred_dim 1, policy_aux off: a 7-float head pitch through the same kernels)."""
    from ga_ddpg_amd.core.networks import GaussianPolicy
    from ga_ddpg_amd.core.utils import PandaTaskSpace6D
    from oracle.detfill import fill_module_
    g = np.load(os.path.join(golden_dir, "heads.npz"))
    x = torch.tensor(g["x"]).cuda()             # a plain normal draw: negative "features" too (no ReLU is implied)
    for tag, dim in (("pi", 7), ("p1", 1)):
        p = fill_module_(GaussianPolicy(513, 6, 256, PandaTaskSpace6D(), extra_pred_dim=dim), "policy", SEED)
        p.cuda()
        msq, logp, act, extra = p.sample(x, eps=g["pi_eps"])
        assert_close(msq.cpu().numpy(), g[tag + "_mean"], 1e-4, 2e-6, tag + " squashed mean")
        assert_close(logp.cpu().numpy(), g[tag + "_log_prob"], 1e-4, 2e-4, tag + " log_prob")
        assert_close(act.cpu().numpy(), g[tag + "_action"], 1e-4, 2e-6, tag + " action")
        assert_close(extra.cpu().numpy(), g[tag + "_extra"], 1e-4, 2e-5, tag + " extra")
    p7 = fill_module_(GaussianPolicy(513, 6, 256, PandaTaskSpace6D(), extra_pred_dim=7), "policy", SEED)
    p7.cuda()
    mean, log_std, extra = p7(x)
    assert_close(mean.cpu().numpy(), g["pi_raw_mean"], 1e-4, 2e-5, "raw mean")
    assert_close(log_std.cpu().numpy(), g["pi_log_std"], 1e-4, 2e-5, "log_std")
    assert_close(extra.cpu().numpy(), g["pi_extra"], 1e-4, 2e-5, "extra (forward)")
    # QNetwork on the same plain tensor
    from ga_ddpg_amd.core.networks import QNetwork
    q = fill_module_(QNetwork(513, 0, 256, extra_pred_dim=7), "critic", SEED)
    q.cuda()
    q1, q2, aux = q(x)
    assert_close(q1.cpu().numpy(), g["q1"], 1e-4, 2e-5, "q1")
    assert_close(q2.cpu().numpy(), g["q2"], 1e-4, 2e-5, "q2")
    assert_close(aux.cpu().numpy(), g["aux"], 1e-4, 2e-5, "critic aux")


def test_gaussian_policy_forward_and_sample_vs_oracle(golden_dir):
    """same quantities against the CPU oracle's restatement (itself pinned to the reference golden by
    tests/test_oracle_golden.py::test_heads) on non-negative features, as the heads see them after the encoder's ReLU"""
    from ga_ddpg_amd.core.networks import GaussianPolicy
    from ga_ddpg_amd.core.utils import PandaTaskSpace6D
    from oracle import ref_step
    from oracle.detfill import fill_module_
    rng = np.random.default_rng(5)
    B = 37
    x = rng.normal(size=(B, 513)).astype(np.float32)
    eps = rng.normal(size=(B, 6)).astype(np.float32)
    for dim in (7, 1):
        p = fill_module_(GaussianPolicy(513, 6, 256, PandaTaskSpace6D(), extra_pred_dim=dim), "policy", SEED)
        p.cuda()
        o = fill_module_(ref_step.PolicyNet(513, 6, 256, dim), "policy", SEED)
        with torch.no_grad():
            w_msq, w_logp, w_act, w_extra, w_mean, w_ls = [t.numpy() for t in o.sample(torch.tensor(x), torch.tensor(eps))]
        msq, logp, act, extra = p.sample(torch.tensor(x).cuda(), eps=eps)
        mean, log_std, extra2 = p.forward(torch.tensor(x).cuda())
        assert_close(msq.cpu().numpy(), w_msq, 1e-4, 2e-6, "squashed mean")
        assert_close(act.cpu().numpy(), w_act, 1e-4, 2e-6, "action")
        assert_close(logp.cpu().numpy(), w_logp, 1e-4, 2e-4, "log_prob")
        assert_close(extra.cpu().numpy(), w_extra, 1e-4, 2e-5, "extra")
        assert_close(extra2.cpu().numpy(), w_extra, 1e-4, 2e-5, "extra (forward)")
        assert_close(mean.cpu().numpy(), w_mean, 1e-4, 2e-5, "raw mean")
        assert_close(log_std.cpu().numpy(), w_ls, 1e-4, 2e-5, "log_std")
        assert log_std.min() >= -10 and log_std.max() <= 2
        # the default draw comes from the device generator: different calls, different samples, same mean
        a1, a2 = p.sample(torch.tensor(x).cuda())[2], p.sample(torch.tensor(x).cuda())[2]
        assert (a1 != a2).any()


def test_select_action_returns_the_reference_tuple():
    """reference core/agent.py:116-124: (squashed mean, log-prob scalar, sampled action, aux pose)"""
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from oracle import ref_step
    from oracle.detfill import fill_module_
    agent, cfg = make_agent("ddpg_td3_aux.yaml")
    oracle = ref_step.OracleAgent(load_cfg("ddpg_td3_aux.yaml").RL_TRAIN)
    for name in ("policy", "policy_target", "critic", "critic_target", "state_feature_extractor"):
        fill_module_(getattr(agent, name), name, SEED)
    for name, net in oracle.nets().items():
        fill_module_(net, name, SEED)
    mem = BaseMemory(300, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 300, seed=4)
    batch = sample_valid_batch(mem, 8, np.random.default_rng(2))
    one = batch["point_state_batch"][5]
    eps = np.random.default_rng(9).normal(size=6).astype(np.float32)
    action, logp, sample, aux = agent.select_action([[one, None]], remain_timestep=11, eps=eps)
    oracle.state_feature_extractor.eval()
    with torch.no_grad():
        z = oracle.state_feature_extractor(torch.tensor(one[None]), value=False)
        w = oracle.policy.sample(torch.cat([z, torch.tensor([[11.0]])], 1), torch.tensor(eps[None]))
    assert isinstance(logp, float) and action.shape == (6,) and sample.shape == (6,) and aux.shape == (7,)
    assert_close(action, w[0][0].numpy(), 1e-4, 2e-6, "action")
    assert_close(logp, float(w[1][0, 0]), 1e-4, 5e-4, "log-prob")
    assert_close(sample, w[2][0].numpy(), 1e-4, 2e-6, "action sample")
    assert_close(aux, w[3][0].numpy(), 1e-4, 2e-5, "aux pose")
    # without an injected draw the sample differs from the mean
    _, _, s2, _ = agent.select_action([[one, None]], remain_timestep=11)
    assert np.abs(s2 - action).max() > 0


def _small_cloud_cfg(name):
    from ga_ddpg_amd.experiments.config import load_cfg
    c = load_cfg(name)
    c.RL_TRAIN.uniform_num_pts = 128
    c.RL_SAVE_DATA_NAME = "replay_io_saved.npz"
    return c


def test_offline_loop_checkpoints_resume_and_migrate(golden_dir, tmp_path):
    """SURVEY A22 / N4, reference core/train_test_offline.py:107-161 + core/utils.py:319-334 as ONE loop:
    BaseMemory.load(file written by the reference's class) -> epochs of updates_per_step x (sample -> update_parameters ->
    step_scheduler) at OFFLINE_BATCH_SIZE = 100 -> save_model at a save_epoch step and at the end -> a FRESH agent
    load_model()s before its first update and reproduces the original's next step (losses, parameters, Adam state);
    a BC checkpoint is migrated to the DDPG file names and loaded with set_init_step."""
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core import train_test_offline as tto
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.core.utils import migrate_model
    data = tmp_path / "data"
    data.mkdir()
    shutil.copy(os.path.join(golden_dir, "replay_io_saved.npz"), data / "replay_io_saved.npz")
    cfg = _small_cloud_cfg("ddpg_td3_aux.yaml")
    assert cfg.OFFLINE_BATCH_SIZE == 100
    torch.manual_seed(3)
    agent, _ = make_agent(cfg)
    config = cfg.RL_TRAIN
    config.batch_size = cfg.OFFLINE_BATCH_SIZE
    config.updates_per_step = 4
    config.max_epoch = 9                      # stop once update_step >= 9: two epochs of four updates (update_step starts at 1)
    config.save_epoch = [6]
    mem = BaseMemory(48, cfg)
    mem.load(str(data))
    assert mem.upper_idx() > 20
    np.random.seed(11)
    out = tmp_path / "run"
    logs = []
    losses, epochs = tto.train_off_policy(agent, mem, config, str(out), save_model=True, log=logs.append)
    assert epochs == 2 and agent.update_step == 9
    assert all(np.isfinite(list(h)).all() for h in losses.values())
    assert len(losses["critic_loss"]) == 9                                               # deque([0]) + 8 updates
    assert any("updates: 9" in l for l in logs)
    assert os.path.exists(out / "DDPG_actor_PandaYCBEnv_epoch_6") and os.path.exists(out / "DDPG_state_feat_PandaYCBEnv_epoch_6")
    assert torch.load(out / "DDPG_state_feat_PandaYCBEnv_epoch_6", weights_only=False)["step"] == 6
    # ---- the same loop with run-ahead updates (losses read at the end of each epoch): same bookkeeping, finite losses
    torch.manual_seed(3)
    agent_b, _ = make_agent(cfg)
    np.random.seed(11)
    losses_b, epochs_b = tto.train_off_policy(agent_b, mem, config, str(tmp_path / "run_b"), save_model=True, run_ahead=True)
    assert epochs_b == 2 and agent_b.update_step == 9 and len(losses_b["critic_loss"]) == 9
    assert all(np.isfinite(list(h)).all() for h in losses_b.values())
    assert abs(list(losses_b["critic_loss"])[1] - list(losses["critic_loss"])[1]) <= 1e-3 * abs(list(losses["critic_loss"])[1]) + 1e-6
    assert torch.load(tmp_path / "run_b" / "DDPG_state_feat_PandaYCBEnv_epoch_6", weights_only=False)["step"] == 6
    agent.save_model(agent.update_step, output_dir=str(out))
    # ---- resume on a fresh agent BEFORE its first update (the Adam moments must come from the file, not from zero)
    torch.manual_seed(4)
    fresh, _ = make_agent(_small_cloud_cfg("ddpg_td3_aux.yaml"))
    assert fresh._rt is None
    assert fresh.load_model(str(out)) == 9 and fresh.update_step == 9
    from ga_ddpg_amd.core.utils import hard_update
    hard_update(agent.policy_target, agent.policy)          # load_model hard-copies the targets (reference :386,404)
    hard_update(agent.critic_target, agent.critic)
    batch = mem.sample(batch_size=100)
    u = np.random.default_rng(1).random((100, 6)).astype(np.float32)
    r1 = agent.update_parameters(batch, agent.update_step, 0, noise_u=u)
    r2 = fresh.update_parameters(batch, fresh.update_step, 0, noise_u=u)
    for k in r1:
        assert_close(r2[k], r1[k], 5e-3, 1e-6, "resumed step: " + k)
    for name in ("pol", "cr", "enc", "venc"):
        f1, f2 = getattr(agent._rt, name).flat, getattr(fresh._rt, name).flat
        assert f1.step_count == f2.step_count and f1.step_count > 1
        m1, m2 = f1.exp_avg.cpu().numpy(), f2.exp_avg.cpu().numpy()
        # a resume that dropped the moments would give exp_avg = (1 - beta1) * g: a tenth of the true value
        assert np.abs(m2 - m1).max() <= 2e-2 * np.abs(m1).max() + 1e-9, name + " exp_avg after resume"
        p1, p2 = f1.master.cpu().numpy(), f2.master.cpu().numpy()
        assert np.abs(p2 - p1).max() <= 3.0 * 1e-3 + 1e-7, name + " parameters after the resumed step"
    # ---- migrate a BC checkpoint into a DDPG run directory
    torch.manual_seed(5)
    bc, _ = make_agent(_small_cloud_cfg("bc_dagger_aux.yaml"))
    bc.update_parameters(mem.sample(batch_size=64), bc.update_step, 0)
    bc_dir, mig = tmp_path / "bc", tmp_path / "mig"
    bc.save_model(bc.update_step, output_dir=str(bc_dir))
    assert sorted(os.listdir(bc_dir)) == ["BC_actor_PandaYCBEnv_latest", "BC_state_feat_PandaYCBEnv_latest"]
    copied = migrate_model(str(bc_dir), str(mig))
    assert sorted(os.listdir(mig)) == ["DDPG_actor_PandaYCBEnv_latest", "DDPG_state_feat_PandaYCBEnv_latest"]
    assert len(copied) == 2
    d2, _ = tto.setup(_small_cloud_cfg("ddpg_td3_aux.yaml"), pretrained=str(bc_dir), output_dir=str(tmp_path / "mig2"))
    assert d2.update_step == bc.update_step and d2.init_step == bc.update_step
    for (n1, p1), (n2, p2) in zip(bc.policy.named_parameters(), d2.policy.named_parameters()):
        assert n1 == n2 and torch.equal(p1.detach().cpu(), p2.detach().cpu()), n1
    sd1, sd2 = bc.state_feature_extractor.state_dict(), d2.state_feature_extractor.state_dict()
    assert all(torch.equal(sd1[k].cpu(), sd2[k].cpu()) for k in sd1)
    assert np.isfinite(list(d2.update_parameters(mem.sample(batch_size=100), d2.update_step, 0).values())).all()


def test_reinit_optim_restarts_the_lr_schedule(tmp_path):
    """reference core/agent.py:375-383,394-401: with reinit_optim a checkpoint loaded with set_init_step gets lr =
    reinit_lr and fresh MultiStepLR schedules for policy and critic"""
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.experiments.config import load_cfg
    a1, _ = make_agent("ddpg_td3_aux.yaml")
    a1.save_model(1, output_dir=str(tmp_path))
    cfg = load_cfg("ddpg_td3_aux.yaml")
    cfg.RL_TRAIN.reinit_optim = True
    a2, _ = make_agent(cfg)
    a2.load_model(str(tmp_path), set_init_step=True)
    assert a2.policy_optim.param_groups[0]["lr"] == cfg.RL_TRAIN.reinit_lr == 1e-4
    assert a2.critic_optim.param_groups[0]["lr"] == 1e-4
    assert a2.policy_scheduler.base_lrs[0] == 1e-4 and list(a2.policy_scheduler.milestones) == list(cfg.RL_TRAIN.policy_milestones)
    a3, _ = make_agent(cfg)
    a3.load_model(str(tmp_path), set_init_step=False)            # without set_init_step the loaded lr stays
    assert a3.policy_optim.param_groups[0]["lr"] == 3e-4


def test_prefetched_pinned_batches_feed_the_update():
    """BASELINE configs[4] 'sample-prefetch': minibatches staged by the background sampler (pinned float32 tensors) go
    through update_parameters exactly like the host dict the reference's loop would have sampled"""
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.prefetch import PrefetchSampler
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from oracle.detfill import fill_module_
    res = {}
    for mode in ("host", "prefetch"):
        agent, cfg = make_agent("ddpg_td3_aux.yaml")
        for name in ("policy", "policy_target", "critic", "critic_target", "state_feature_extractor"):
            fill_module_(getattr(agent, name), name, 5)
        mem = BaseMemory(600, cfg, point_dtype=np.float32)
        fill_synthetic_buffer(mem, 600, seed=4)
        rng = np.random.default_rng(2)
        u = np.random.default_rng(3).random((16, 6)).astype(np.float32)
        if mode == "host":
            b = sample_valid_batch(mem, 16, rng)
            res[mode] = agent.update_parameters(b, agent.update_step, 0, noise_u=u)
        else:
            with PrefetchSampler(mem, 16, sample=lambda n: sample_valid_batch(mem, n, rng)) as s:
                b = s.next()
                assert b["point_state_batch"].is_pinned() and b["point_state_batch"].dtype == torch.float32
                res[mode] = agent.update_parameters(b, agent.update_step, 0, noise_u=u)
                for i in range(3):                                  # staging sets recycle while updates keep running
                    out = agent.update_parameters(s.next(), agent.update_step, i + 1)
                    assert np.isfinite(list(out.values())).all()
    for k in res["host"]:
        assert_close(res["prefetch"][k], res["host"][k], 1e-5, 1e-7, k)


def test_prefetch_sampler_under_run_ahead_steps():
    """run-ahead updates return before their uploads have run: a staging set must not be refilled until the event the
    runtime hangs on the batch (`uploaded_event`).  Learning rate 0 makes every step a function of ITS minibatch only, so a
    step that saw a half-overwritten staging set shows up against the synchronous loop over the same minibatches."""
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.prefetch import PrefetchSampler
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from oracle.detfill import fill_module_
    runs = {}
    for mode in ("sync", "ahead"):
        agent, cfg = make_agent("ddpg_td3_aux.yaml")
        for name in ("policy", "policy_target", "critic", "critic_target", "state_feature_extractor"):
            fill_module_(getattr(agent, name), name, 5)
        for opt in (agent.policy_optim, agent.critic_optim, agent.state_feat_encoder_optim, agent.state_feat_val_encoder_optim):
            for g in opt.param_groups:
                g["lr"] = 0.0
        mem = BaseMemory(600, cfg, point_dtype=np.float32)
        fill_synthetic_buffer(mem, 600, seed=4)
        rng = np.random.default_rng(2)
        noise = np.random.default_rng(3).random((10, 32, 6)).astype(np.float32)
        logs = []
        with PrefetchSampler(mem, 32, depth=1, sample=lambda n: sample_valid_batch(mem, n, rng)) as s:   # two staging sets circulate
            for i in range(10):
                logs.append(agent.update_parameters(s.next(), agent.update_step, i, noise_u=noise[i], sync=(mode == "sync")))
            agent.flush()
        runs[mode] = [dict(l) for l in logs]
    for i, (a, b) in enumerate(zip(runs["sync"], runs["ahead"])):
        for k in a:
            assert_close(b[k], a[k], 2e-4, 1e-6, "step %d %s" % (i, k))


@pytest.mark.parametrize("source", ["host", "device"])
def test_run_ahead_steps_equal_synchronous_steps(source):
    """update_parameters(sync=False) returns before the step has run; the host stages and enqueues the following steps
    meanwhile (pinned staging sets rotate, engine.HOST_RING).  Same launches in the same order as the synchronous loop:
    the first steps agree to the atomics' rounding, and after a flush the parameters do too."""
    from ga_ddpg_amd.core.agent import PendingLog
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.runtime import BATCH_KEYS
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from tests.test_gpu_step import _filled_agent
    c = load_cfg("ddpg_td3_aux.yaml")
    mem = BaseMemory(1500, c, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 1500, seed=5)
    rng = np.random.default_rng(3)
    batches = [sample_valid_batch(mem, 32, rng) for _ in range(7)]           # more steps than staging sets
    noise = [rng.random((32, 6)).astype(np.float32) for _ in batches]
    if source == "device":
        batches = [{k: torch.as_tensor(np.ascontiguousarray(b[k], dtype=np.float32)).cuda() for k in BATCH_KEYS} for b in batches]
    def run(mode, lr):
        agent, nets = _filled_agent("ddpg_td3_aux.yaml", 11)
        if lr is not None:
            for opt in (agent.policy_optim, agent.critic_optim, agent.state_feat_encoder_optim, agent.state_feat_val_encoder_optim):
                for g in opt.param_groups:
                    g["lr"] = lr
        logs = []
        for i, (b, u) in enumerate(zip(batches, noise)):
            if mode == "prefetch" and i + 1 < len(batches):
                # the reference-shaped loop with the NEXT minibatch staged ahead (Agent.prefetch -> FusedRuntime.prefetch_inputs):
                # before the first step there is no runtime yet (False), host batches are never staged ahead (False)
                staged = agent.prefetch(batches[i + 1])
                assert staged == (source == "device" and i >= 1), (i, staged)
            out = agent.update_parameters(b, agent.update_step, 0, noise_u=u, sync=(mode != "ahead"))
            agent.step_scheduler(agent.update_step)
            logs.append(out)
        if mode == "ahead":
            assert all(isinstance(l, PendingLog) for l in logs)
            assert not logs[-1].done()                                        # nothing was read yet
            agent.flush()
        return ([dict(l) for l in logs], {n: p.detach().clone() for nn, net in nets.items() for n, p in
                                          ((nn + "/" + k, v) for k, v in net.named_parameters())})
    # (a) the configured learning rates: the first steps agree to the atomics' rounding (later ones are separated float32
    # trajectories: two synchronous runs differ by 2x in critic_loss at step 5 of these B=32 batches)
    (la, pa), (lb, pb) = run("sync", None), run("ahead", None)
    assert set(la[0].keys()) == set(lb[0].keys())
    for s in range(2):          # step 1 follows an Adam step: the atomics' rounding (1e-6) has grown ~100x by then
        for k in la[s]:
            assert_close(lb[s][k], la[s][k], 1e-4 if s == 0 else 3e-3, 1e-6, "step %d %s" % (s, k))
    for n in pa:
        assert float((pa[n] - pb[n]).abs().max()) <= 2.2 * 1e-3 * len(batches), n
        assert bool(torch.isfinite(pb[n]).all())
    # (b) learning rate 0: weights stay put (the target networks still move), every step's numbers depend on ITS batch only -- a
    # step that read another step's inputs, geometry, noise or staging block (sets rotate every 2 / 4 steps) shows up at once
    (la, _), (lb, _) = run("sync", 0.0), run("ahead", 0.0)
    for s in range(len(batches)):
        for k in la[s]:
            assert_close(lb[s][k], la[s][k], 2e-4, 1e-6, "lr=0 step %d %s" % (s, k))
    # (c) synchronous steps with the next minibatch's inputs + geometry staged one step ahead: the same numbers, step by step
    (lc, _) = run("prefetch", 0.0)
    for s in range(len(batches)):
        for k in la[s]:
            assert_close(lc[s][k], la[s][k], 2e-4, 1e-6, "lr=0 prefetched step %d %s" % (s, k))
    assert abs(la[2]["critic_loss"] - la[3]["critic_loss"]) > 1e-3            # the batches do differ
