"""Results must not depend on WHAT ELSE runs on the GPU.  Round 5 finding ("co-running hazard"): with the split-bf16
GEMMs (v_mfma_f32_32x32x16_bf16) of step N in flight, the furthest-point-sampling kernel of step N + 1 -- enqueued on the prefetch
stream, so resident on the same CUs -- picked wrong points in ~1 % of the run-ahead steps: the first VALU consumer of an LDS read saw
stale data in lanes 48-63 (tools/diag_fps_corun.py, tools/ubench/victims.hip).  The kernel now takes the broadcast coordinates
through v_readfirstlane.  Two guards:
  * the geometry entry points beside back-to-back split GEMM launches on a second stream == the same launch alone, bit for bit;
  * the update step in run-ahead mode on ONE minibatch with learning rate 0: the losses that depend on the online networks only
    repeat exactly from step to step (a step whose geometry or activations were disturbed shows up as an outlier)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _aggressors():
    from tests import split_cases as sc
    return [sc.DxWide(27240, 128, 128, "act"), sc.FwdWide(27240, 128, 128, "act"), sc.FwdStream(213034, 64, 64, "act")]


@pytest.mark.parametrize("shape", [(32, 1024, 128), (8, 4096, 256)])
def test_geometry_kernels_beside_split_gemms_equal_the_launch_alone(shape):
    from ga_ddpg_amd import hip
    B, N, M = shape
    S, radius = 32, 0.12
    g = torch.Generator(device="cuda").manual_seed(3)
    xyz = torch.rand(B, N, 3, device="cuda", generator=g)

    def geometry(idx, new_xyz, nb, cnt):
        hip.call("gad_furthest_point_sampling", xyz, B, N, M, idx, new_xyz)
        hip.call("gad_ball_query", new_xyz, xyz, B, N, M, radius, S, nb, cnt)

    def bufs():
        return (torch.zeros(B, M, dtype=torch.int32, device="cuda"), torch.zeros(B, M, 3, device="cuda"),
                torch.zeros(B, M, S, dtype=torch.int32, device="cuda"), torch.zeros(B, M, dtype=torch.int32, device="cuda"))
    ref = bufs()
    geometry(*ref)
    torch.cuda.synchronize()
    cases = _aggressors()
    args = [(getattr(hip.lib(), c.entry), c.args()) for c in cases]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    R = 48
    outs = [bufs() for _ in range(R)]
    bad = []
    for rnd in range(4):
        with torch.cuda.stream(s2):
            for _ in range(120):
                for f, a in args:
                    hip.check(f(C.byref(a), hip.stream()), "gemm")
        with torch.cuda.stream(s1):
            for o in outs:
                geometry(*o)
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            for name, a, b in zip(("fps idx", "new_xyz", "ball idx", "ball cnt"), o, ref):
                if not torch.equal(a, b):
                    bad.append("round %d launch %d: %s differs in %d entries" % (rnd, i, name, int((a != b).sum())))
    assert not bad, "\n".join(bad[:10])


def test_run_ahead_steps_on_one_minibatch_repeat_exactly():
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from tests.test_gpu_step import _filled_agent
    c = load_cfg("ddpg_td3_aux.yaml")
    mem = BaseMemory(1500, c, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 1500, seed=5)
    rng = np.random.default_rng(3)
    batch = sample_valid_batch(mem, 32, rng)
    noise = rng.random((32, 6)).astype(np.float32)
    agent, _ = _filled_agent("ddpg_td3_aux.yaml", 11)
    for opt in (agent.policy_optim, agent.critic_optim, agent.state_feat_encoder_optim, agent.state_feat_val_encoder_optim):
        for grp in opt.param_groups:
            grp["lr"] = 0.0
    logs = [agent.update_parameters(batch, agent.update_step, 0, noise_u=noise, sync=False) for _ in range(400)]
    agent.flush()
    logs = [dict(l) for l in logs]
    for k in ("bc_loss", "policy_grasp_aux_loss", "critic_grasp_aux_loss"):
        v = np.array([l[k] for l in logs], dtype=np.float64)
        for par in (0, 1):                                   # (policy steps and the others log through different launches)
            w = v[par::2]
            med = np.median(w)
            out = np.nonzero(np.abs(w - med) > 2e-5 * abs(med) + 1e-7)[0]
            assert len(out) == 0, "%s: steps %s deviate from the repeated value %.8f: %s" % (k, (2 * out + par).tolist()[:8], med, w[out][:8])


def test_sa_stack_beside_split_gemms_equals_the_stack_alone():
    """the configs[3] two-module stack (eval-mode BatchNorm: no atomics' order in the result) on one stream while split GEMM launches
    run on another: sampled centroids and pooled features bit-equal to the run alone -- FPS, the cell-list ball query, the row
    scan / fill, the gathered first layers, the streaming and wide-tile GEMMs, the fused max-pool and its finalisation as victims"""
    from ga_ddpg_amd import hip
    from ga_ddpg_amd.pointnet2_ops import pointnet2_modules as pm
    from oracle.detfill import fill_module_
    sa = [fill_module_(pm.PointnetSAModule(npoint=512, radius=0.1, nsample=64, mlp=[4, 64, 64, 128]), "sa0", 41).cuda().eval(),
          fill_module_(pm.PointnetSAModule(npoint=128, radius=0.2, nsample=128, mlp=[128, 128, 128, 256]), "sa1", 41).cuda().eval()]
    g = torch.Generator(device="cuda").manual_seed(5)
    xyz = torch.rand(8, 4096, 3, device="cuda", generator=g)
    feats = torch.randn(8, 4, 4096, device="cuda", generator=g)

    def stack():
        with torch.no_grad():
            x1, f1 = sa[0](xyz, feats)
            x2, f2 = sa[1](x1, f1)
        return [t.clone() for t in (x1, f1, x2, f2)]
    ref = stack()
    torch.cuda.synchronize()
    cases = _aggressors()
    args = [(getattr(hip.lib(), c.entry), c.args()) for c in cases]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    s1.wait_stream(torch.cuda.current_stream())
    bad = []
    for rnd in range(3):
        with torch.cuda.stream(s2):
            for _ in range(150):
                for f, a in args:
                    hip.check(f(C.byref(a), hip.stream()), "gemm")
        with torch.cuda.stream(s1):
            outs = [stack() for _ in range(12)]
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            for name, a, b in zip(("new_xyz 1", "features 1", "new_xyz 2", "features 2"), o, ref):
                if not torch.equal(a, b):
                    bad.append("round %d pass %d: %s differs in %d entries (max %.3e)" % (rnd, i, name, int((a != b).sum()), float((a - b).abs().max())))
    assert not bad, "\n".join(bad[:10])


def test_run_ahead_loop_equals_synchronous_loop_over_many_steps():
    """300 steps on one minibatch with learning rate 0 (only the target networks move): every logged value of the run-ahead loop --
    the TD target path and max |critic.grad| included, which see the backward pass too -- equals the synchronous loop's at the same
    step (measured: bit for bit over 2000 steps; 1e-5 here leaves room for the order of the f64 atomics)"""
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from tests.test_gpu_step import _filled_agent
    c = load_cfg("ddpg_td3_aux.yaml")
    mem = BaseMemory(1500, c, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 1500, seed=5)
    rng = np.random.default_rng(3)
    batch = sample_valid_batch(mem, 32, rng)
    noise = rng.random((32, 6)).astype(np.float32)
    runs = {}
    for mode in ("sync", "ahead"):
        agent, _ = _filled_agent("ddpg_td3_aux.yaml", 11)
        for opt in (agent.policy_optim, agent.critic_optim, agent.state_feat_encoder_optim, agent.state_feat_val_encoder_optim):
            for grp in opt.param_groups:
                grp["lr"] = 0.0
        logs = [agent.update_parameters(batch, agent.update_step, 0, noise_u=noise, sync=(mode == "sync")) for _ in range(300)]
        agent.flush()
        runs[mode] = [dict(l) for l in logs]
    for k in runs["sync"][0]:
        a = np.array([l[k] for l in runs["sync"]], dtype=np.float64)
        b = np.array([l[k] for l in runs["ahead"]], dtype=np.float64)
        d = np.abs(a - b) / (np.abs(a) + 1e-9)
        assert float(d.max()) <= 1e-5, "%s: run-ahead differs from the synchronous loop at step %d (%.9g vs %.9g)" % (k, int(d.argmax()), b[d.argmax()], a[d.argmax()])
