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


@pytest.mark.parametrize("value", [False, True])
def test_slab_and_tile_kernels_agree(value):
    B = 96                                             # SA2 ~ 1e4 rows, SA3 3072 rows: both above the slab threshold
    a = _run(B, value, {"fwd_slab": 1, "dx_slab": 2})          # slab kernels for every layer they cover (pooled dX too)
    b = _run(B, value, {"fwd_slab": 0, "dx_slab": 0})
    assert a["rows"] == b["rows"] and a["rows"][2] >= 2048
    bad = []
    for k in a:
        if k == "rows":
            continue
        x, y = a[k].double(), b[k].double()
        scale = float(y.abs().max()) + 1e-30
        err = float((x - y).abs().max())
        med = float((x - y).abs().median())
        # activations / statistics: rounding of a different summation order; gradients: the same, amplified by the
        # BatchNorm-backward cancellation and by the rare ReLU / max-pool kink that the two roundings resolve
        # differently (norm-wise, as in tests/helpers.check_summaries: median tight, worst entry loose)
        tol, mtol = (2e-5, 2e-6) if k[0] in "ZFzmir" else (5e-3, 2e-5)
        print("%-12s max %.3e median %.3e (scale %.3e)" % (k, err, med, scale))
        if err > tol * scale or med > mtol * scale:
            bad.append("%s: max |slab - tile| = %.3e, median %.3e (scale %.3e)" % (k, err, med, scale))
    assert not bad, "\n".join(bad)
