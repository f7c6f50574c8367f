"""Kernel families of the layer GEMMs against each other: the slab kernels of the mid-size layers (SA2 / SA3) and the
64x64 tile kernels compute the same layers (gad_set_option switches the routing); both orders of summation must agree
to float32 rounding on every activation, statistic and gradient of an encoder forward + backward."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(B, value, options):
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
    geo = _geometry(B)
    geo.run(torch.from_numpy(batch["point_state_batch"]).cuda())
    action = torch.from_numpy(batch["action_batch"]).cuda() if value else None
    probe = torch.from_numpy(np.random.default_rng(5).normal(size=(B, 512)).astype(np.float32)).cuda()
    for k, v in options.items():
        hip.set_option(k, v)
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
        if value:
            out["daction"] = slot.daction.clone()
        out["rows"] = n
    finally:
        for k in options:
            hip.set_option(k, 0)                       # library defaults of the two slab switches
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


@pytest.mark.parametrize("value", [False, True])
def test_slab_and_tile_kernels_agree(value):
    """forward: slab vs tile kernels on identical inputs -> every activation, pooled feature and BatchNorm statistic agrees
    to float32 summation-order rounding.  backward: with the SAME forward kernels (identical activations, hence identical
    arg-max routing and ReLU masks) the slab dX kernels -- pooled-gradient source included -- reproduce the tile kernels'
    gradients to rounding; across different forward kernels gradients may differ in the few entries whose max-pool /
    ReLU decision sits within rounding of a tie, so that pairing is only held to the norm-wise median."""
    B = 96                                             # SA2 ~ 1e4 rows, SA3 3072 rows: both above the slab threshold
    tile = _run(B, value, {"fwd_slab": 0, "dx_slab": 0})
    fwd = _run(B, value, {"fwd_slab": 1, "dx_slab": 0})
    dx = _run(B, value, {"fwd_slab": 0, "dx_slab": 2})
    assert tile["rows"] == fwd["rows"] == dx["rows"] and tile["rows"][2] >= 2048
    acts = [k for k in tile if k[0] in "ZFzmir" and k != "rows"]
    grads = [k for k in tile if k not in acts and k != "rows"]
    bad = _compare(fwd, tile, acts, 2e-5, 2e-6, "forward slab vs tile:")
    bad += _compare(dx, tile, acts, 1e-6, 1e-7, "same forward kernels:")       # (BatchNorm sums: f64 atomics, order varies)
    bad += _compare(dx, tile, grads, 2e-4, 1e-5, "dX slab vs tile:")
    bad += _compare(fwd, tile, grads, 1.0, 2e-4, "forward slab vs tile (gradients, median):")
    assert not bad, "\n".join(bad)
