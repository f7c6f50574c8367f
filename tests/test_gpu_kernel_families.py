"""Kernel families of the layer GEMMs against each other: the specialised kernels (streaming SA1 forward / dX / dW, skinny
split-K kernels of the FC head, wide-tile kernels of the mid-size layers) and the generic 64x64 tile kernels compute the same layers (gad_set_option switches the
routing); both orders of summation must agree to float32 rounding on every activation, statistic and gradient of an
encoder forward + backward.  Also here: the max-pool folded into the pooled layers' GEMM epilogue against the stand-alone
segment max-pool operator on the same raw activations."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(B, value, options, keep_slot=False, mutate=None):
    from ga_ddpg_amd import engine, hip
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from tests.test_gpu_encoder import _feature_net, _geometry, _run_encoder
    dev = torch.device("cuda")
    cfg = load_cfg("ddpg_td3_aux.yaml")
    mem = BaseMemory(400, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 400, seed=11)
    batch = sample_valid_batch(mem, B, np.random.default_rng(3))
    net = _feature_net()
    if mutate is not None:
        mutate(net)
    geo = _geometry(B)
    geo.run(torch.from_numpy(batch["point_state_batch"]).cuda())
    action = torch.from_numpy(batch["action_batch"]).cuda() if value else None
    probe = torch.from_numpy(np.random.default_rng(5).normal(size=(B, 512)).astype(np.float32)).cuda()
    options = dict(options)
    sched = {k: options.pop(k) for k in list(options) if k.isupper()}       # engine schedule switches (module constants)
    if options.get("fwd_stream", 1) == 0:                                   # mode 2 of gad_gemm_fwd is a streaming-kernel form
        sched.setdefault("RECOMP_SA1", False)
    sched_was = {k: getattr(engine, k) for k in sched}
    for k, v in sched.items():
        setattr(engine, k, v)
    for k, v in options.items():
        hip.set_option(k, v)
    fused_wide_was = engine.FUSED_WIDE_BWD
    engine.FUSED_WIDE_BWD = bool(options.get("bwd_wide", 0))         # (the plans take the fused call only when asked to)
    try:
        enc = engine.EncoderNet(net.value_encoder if value else net.encoder, dev)
        slot = engine.EncoderSlot(geo, enc, dev)
        z = _run_encoder(enc, slot, action, probe, value)
        n = [int(geo.rows[s]["n"].item()) for s in range(3)]
        out = {"z": z.clone(), "mean": slot.mean.clone(), "istd": slot.istd.clone(), "grad": enc.flat.grad.clone(),
               "running_mean": enc.running_mean.clone(), "running_var": enc.running_var.clone()}
        for s in range(3):
            for l in range(3):
                out["Z%d%d" % (s + 1, l + 1)] = slot.Z[s][l][:n[s]].clone()
            out["F%d" % (s + 1)] = slot.F[s].clone()
            out["dF%d" % (s + 1)] = slot.dF[s].clone()
            out["A%d" % (s + 1)] = slot.argmax[s].clone()
            out["zmax%d" % (s + 1)] = slot.zmax[s].clone()
        if value:
            out["daction"] = slot.daction.clone()
        out["rows"] = n
        if keep_slot:
            out.update(slot=slot, enc=enc, geo=geo)
    finally:
        engine.FUSED_WIDE_BWD = fused_wide_was
        for k, v in sched_was.items():
            setattr(engine, k, v)
        for k in options:
            hip.set_option(k, hip.get_option_default(k) if k == "mfma_split" else (0 if k == "bwd_wide" else 1))   # library defaults of the family switches
    return out


def _compare(a, b, keys, tol, mtol, what):
    bad = []
    for k in keys:
        x, y = a[k].double(), b[k].double()
        scale = float(y.abs().max()) + 1e-30
        err, med = float((x - y).abs().max()), float((x - y).abs().median())
        if err > tol * scale or med > mtol * scale:
            bad.append("%s %s: max diff %.3e, median %.3e (scale %.3e)" % (what, k, err, med, scale))
    return bad


FAMILIES = ("fwd_stream", "dx_stream", "dw_stream", "fwd_skinny", "dx_skinny", "dw_skinny", "fwd_wide", "dx_wide", "dw_wide", "bwd_fused")


@pytest.mark.parametrize("value", [False, True])
def test_specialised_and_tile_kernels_agree(value):
    """forward: streaming / skinny kernels vs the tile kernels on identical inputs -> every activation, pooled feature and
    BatchNorm statistic agrees to float32 summation-order rounding.  backward: with the SAME forward kernels (identical
    activations, hence identical arg-max routing and ReLU masks) the specialised dX / dW kernels reproduce the tile kernels'
    gradients to rounding; across different forward kernels gradients may differ in the few entries whose max-pool /
    ReLU decision sits within rounding of a tie, so that pairing is only held to the norm-wise median."""
    B = 96                                             # SA1 ~ 8e4 rows: above the streaming threshold
    default = _run(B, value, {})
    tile = _run(B, value, {k: 0 for k in FAMILIES})
    fwd_tile = _run(B, value, {"fwd_stream": 0, "fwd_skinny": 0, "fwd_wide": 0})
    bwd_tile = _run(B, value, {k: 0 for k in FAMILIES if not k.startswith("fwd")})
    assert default["rows"] == tile["rows"] == fwd_tile["rows"] and default["rows"][0] >= 32768
    acts = [k for k in tile if k[0] in "ZFzmir" and k != "rows"]
    grads = [k for k in tile if k not in acts and k != "rows"]
    bad = _compare(fwd_tile, default, acts, 2e-5, 2e-6, "forward tile vs specialised:")
    bad += _compare(bwd_tile, default, acts, 1e-6, 1e-7, "same forward kernels:")       # (BatchNorm sums: f64 atomics, order varies)
    bad += _compare(bwd_tile, default, grads, 2e-4, 1e-5, "backward tile vs specialised:")
    bad += _compare(tile, default, grads, 1.0, 2e-4, "all tile vs specialised (gradients, median):")
    # the fused SA1 dX + dW kernel (gad_gemm_bwd) against the two separate streaming kernels: same forward, same sums
    unfused = _run(B, value, {"bwd_fused": 0})
    bad += _compare(unfused, default, acts, 1e-6, 1e-7, "same forward kernels (unfused backward):")
    bad += _compare(unfused, default, grads, 2e-4, 1e-5, "separate dX / dW vs fused SA1 backward:")
    # round 4: the fused SA2 / SA3 dX + dW kernel (gemm_bwd_wide_kernel) against the separate wide-tile kernels
    fused_w = _run(B, value, {"bwd_wide": 1})
    bad += _compare(fused_w, default, acts, 1e-6, 1e-7, "same forward kernels (fused wide backward):")
    bad += _compare(fused_w, default, grads, 2e-4, 1e-5, "separate wide dX / dW vs fused wide backward:")
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("value", [False, True])
def test_split_bf16_option_stays_within_f32_summation_noise(value):
    """library option "mfma_split" (the default since round 5; 0 = FP32 MFMA throughout): every GEMM family of the encoder forms its
    FP32 products as six bf16 x bf16 term products (hi / mid / lo splits, 24 significand bits) on v_mfma_f32_32x32x16_bf16.  The
    two modes agree as tightly as two f32 summation orders do (the bounds of tile vs specialised kernels), i.e. the split form's
    error is f32-sized, not bf16-sized (which would show as 4e-3).  The per-family float64 gates are
    tests/test_gpu_split_families.py; the oracle-facing gates run in both modes (tests/conftest.py BOTH_MODES)."""
    B = 96
    ref = _run(B, value, {"mfma_split": 0})
    got = _run(B, value, {"mfma_split": 1})
    from ga_ddpg_amd import hip
    hip.set_option("mfma_split", hip.get_option_default("mfma_split"))
    assert got["rows"] == ref["rows"] and ref["rows"][0] >= 32768
    assert not torch.equal(got["Z12"], ref["Z12"]), "the option did not change the arithmetic"
    acts = [k for k in ref if k[0] in "ZFzmir" and k != "rows"]
    grads = [k for k in ref if k not in acts and k != "rows"]
    bad = _compare(got, ref, acts, 2e-5, 2e-6, "split-bf16 vs f32 MFMA, forward:")
    bad += _compare(got, ref, grads, 1.0, 2e-4, "split-bf16 vs f32 MFMA, gradients (median; ties reroute single entries):")
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("value", [False, True])
def test_recomputed_first_layer_is_bit_equal(value):
    """round 4: SA1 layer 2 recomputes layer 1's output from the gathered rows (gad_gemm_fwd mode 2) with the
    operands of layer 1's own MFMAs swapped: the same products in the same order, so layer 2's raw output equals the one computed
    from the stored z1 BIT FOR BIT, and with it everything downstream up to the order of the statistics' f64 atomics."""
    B = 96
    # (the recomputing form exists for the FP32-MFMA products only -- its premise is "the same products in the same order" -- so both
    # runs pin mfma_split = 0; with the split default a mode-2 launch takes the f32 kernel while the stored-z1 launch would not)
    ref = _run(B, value, dict(RECOMP_SA1=False, mfma_split=0))
    got = _run(B, value, dict(RECOMP_SA1=True, mfma_split=0))
    assert got["rows"] == ref["rows"] and ref["rows"][0] >= 32768
    assert torch.equal(got["Z11"], ref["Z11"]) and torch.equal(got["Z12"], ref["Z12"]), "recomputed z1 is not the stored z1"
    acts = [k for k in ref if k[0] in "ZFzmir" and k != "rows"]
    grads = [k for k in ref if k not in acts and k != "rows"]
    bad = _compare(got, ref, acts, 1e-6, 1e-7, "recomputed SA1 layer 1, forward:")
    bad += _compare(got, ref, grads, 2e-5, 1e-6, "recomputed SA1 layer 1, gradients:")
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("value", [False, True])
def test_batchnorm_prologues_equal_the_standalone_launches(value):
    """round 4: a layer's train-mode BatchNorm finalised in the prologue of the launch that consumes its output
    (gad_gemm_fwd_args.in_*) and the BatchNorm-backward coefficients formed in the dX / dW prologues (gad_dz_src.bn_*) against the
    stand-alone gad_bn_finalize / gad_bn_bwd_coef launches: the same arithmetic on the same sums, so every activation, published
    statistic, running statistic and gradient agrees to the order of the f64 atomics.  With the specialised families switched
    off the tile kernels -- which have no prologue -- take the blocks: the entry points then run the stand-alone launch
    themselves, and the result must not depend on the route."""
    B = 96
    launches = dict(DEFER_BN_WIDE=False, INLINE_BN_BWD=False)
    everywhere = dict(DEFER_BN_WIDE=True, DEFER_BN_STAGES=(0, 1, 2, 3), INLINE_BN_BWD=True)
    ref = _run(B, value, launches)
    acts = [k for k in ref if k[0] in "ZFzmir" and k != "rows"]
    grads = [k for k in ref if k not in acts and k != "rows"]
    bad = []
    for name, opts in (("default plans", {}), ("every stage + FC pair", everywhere)):
        got = _run(B, value, opts)
        assert got["rows"] == ref["rows"]
        bad += _compare(got, ref, acts, 1e-6, 1e-7, "%s, forward:" % name)
        bad += _compare(got, ref, grads, 2e-5, 1e-6, "%s, gradients:" % name)
    tile = {k: 0 for k in FAMILIES}
    ref_t = _run(B, value, dict(tile, **launches))
    got_t = _run(B, value, dict(tile, **everywhere))
    bad += _compare(got_t, ref_t, acts, 1e-6, 1e-7, "tile kernels (entry points run the launches), forward:")
    bad += _compare(got_t, ref_t, grads, 2e-5, 1e-6, "tile kernels (entry points run the launches), gradients:")
    # gad_set_option("fwd_bn_prologue", 0): gad_gemm_fwd finalises with a launch of its own on every route
    from ga_ddpg_amd import hip
    hip.set_option("fwd_bn_prologue", 0)
    try:
        got_o = _run(B, value, everywhere)
    finally:
        hip.set_option("fwd_bn_prologue", 1)
    bad += _compare(got_o, ref, acts, 1e-6, 1e-7, "fwd_bn_prologue = 0, forward:")
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("value", [False, True])
def test_fused_pool_equals_segment_pool_operator(value):
    """the max-pool folded into the third GEMM of every stage (packed arg-max keys + gad_pool_finalize) against
    gad_segment_pool run on the SAME stored raw activations with the same published scale / shift: pooled features
    bit-equal (a rounded fma is monotone, so relu(fma(max z)) == max relu(fma(z))); arg-max rows identical except where two
    different raw values of a group round to the same activation (the fused pool then keeps the larger raw value, torch
    the first row) -- such positions must carry bit-equal activations; zmax is the raw value at the reported row."""
    from ga_ddpg_amd import hip
    for B in (24, 96):
        out = _run(B, value, {}, keep_slot=True)
        slot, enc, geo = out["slot"], out["enc"], out["geo"]
        for s in range(3):
            m = enc.sa_mats[s][2]
            o = enc.bn_off[m.bn_index]
            r = geo.rows[s]
            F = torch.empty_like(slot.F[s])
            A = torch.empty_like(slot.argmax[s])
            hip.call("gad_segment_pool", slot.Z[s][2], m.n_out, m.n_out, slot.scale[o:o + m.n_out], slot.shift[o:o + m.n_out],
                     r["off"], r["G"], F, A)
            torch.cuda.synchronize()
            assert torch.equal(F, out["F%d" % (s + 1)]), "stage %d: pooled features differ" % (s + 1)
            Af = out["A%d" % (s + 1)]
            diff = (A != Af)
            z = slot.Z[s][2]
            cols = torch.arange(m.n_out, device=z.device).expand_as(A)
            sc, sh = slot.scale[o:o + m.n_out], slot.shift[o:o + m.n_out]
            ya = torch.relu(torch.addcmul(sh.double(), z[A.long(), cols].double(), sc.double()).float())
            yf = torch.relu(torch.addcmul(sh.double(), z[Af.long(), cols].double(), sc.double()).float())
            assert float(diff.float().mean()) <= 1e-4, "stage %d: %d arg-max rows differ" % (s + 1, int(diff.sum()))
            if bool(diff.any()):
                assert torch.equal(ya[diff], yf[diff]), "stage %d: a differing arg-max row does not tie" % (s + 1)
            assert bool((slot.key[s] == 0).all()), "stage %d: keys not reset" % (s + 1)
            zm = out["zmax%d" % (s + 1)]
            live = out["F%d" % (s + 1)] > 0
            assert torch.equal(zm[live], z[Af.long(), cols][live]), "stage %d: zmax is not the raw value at the arg-max row" % (s + 1)


def test_pooled_layers_with_zero_and_negative_gamma():
    """the fused max-pool reduces sgn(gamma) * z per group, so channels with gamma < 0 (the minimum of z wins) and
    gamma == 0 (the activation is constant over the rows: every row ties, torch keeps the FIRST row) are its special cases.
    Pooled features and arg-max rows must equal the stand-alone segment max-pool operator on the same raw activations, and
    dgamma of the pooled layers must equal sum_g dF[g, c] * x_hat[argmax[g, c], c] with that operator's arg-max rows -- for the
    gamma == 0 channels this is the first row of each group, not the row whose raw value won."""
    from ga_ddpg_amd import hip

    def mutate(net):
        with torch.no_grad():
            for enc in (net.encoder, net.value_encoder):
                for name, mod in enc.named_modules():
                    if isinstance(mod, torch.nn.BatchNorm2d) and name.endswith(".7"):      # BatchNorm of each stage's third conv
                        mod.weight[0:4] = 0.0
                        mod.bias[0:2] = 0.3                                               # relu(shift) > 0: the gradient flows
                        mod.bias[2:4] = -0.3
                        mod.weight[4:12] = -mod.weight[4:12].abs() - 0.1
    out = _run(48, True, {}, keep_slot=True, mutate=mutate)
    slot, enc, geo = out["slot"], out["enc"], out["geo"]
    grad = out["grad"]
    for s in range(3):
        m = enc.sa_mats[s][2]
        assert float(m.bn.weight[0]) == 0.0 and float(m.bn.weight[5]) < 0.0, "the mutation did not reach stage %d" % (s + 1)
        o = enc.bn_off[m.bn_index]
        r = geo.rows[s]
        F = torch.empty_like(slot.F[s])
        A = torch.empty_like(slot.argmax[s])
        sc, sh = slot.scale[o:o + m.n_out], slot.shift[o:o + m.n_out]
        hip.call("gad_segment_pool", slot.Z[s][2], m.n_out, m.n_out, sc, sh, r["off"], r["G"], F, A)
        torch.cuda.synchronize()
        assert torch.equal(F, out["F%d" % (s + 1)]), "stage %d: pooled features differ" % (s + 1)
        Af = out["A%d" % (s + 1)]
        assert torch.equal(A[:, :12], Af[:, :12]), "stage %d: arg-max rows of the gamma <= 0 channels differ" % (s + 1)
        first = r["off"][:r["G"]].long()
        assert torch.equal(Af[:, :4].long(), first[:, None].expand(-1, 4)), "gamma == 0: the first row of every group"
        # dgamma of the pooled layer from its definition, with the operator's arg-max rows
        z = slot.Z[s][2]
        cols = torch.arange(m.n_out, device=z.device).expand_as(A)
        zs = z[A.long(), cols].double()
        live = torch.addcmul(sh.double(), zs, sc.double()).float() > 0
        xhat = (zs - slot.mean[o:o + m.n_out].double()) * slot.istd[o:o + m.n_out].double()
        want = (out["dF%d" % (s + 1)].double() * live * xhat).sum(0)
        idx = [i for i, p in enumerate(enc.flat.params) if p is m.bn.weight][0]
        off = int(enc.flat.offsets[idx])
        got = grad[off:off + m.n_out].double()
        scale = float(want.abs().max()) + 1e-30
        err = (got - want).abs() / scale
        assert float(err.max()) <= 2e-4, "stage %d: dgamma of the pooled layer, worst channel %d: %.3e" % (s + 1, int(err.argmax()), float(err.max()))
