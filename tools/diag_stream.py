"""Diagnostic: streaming vs tiled SA1 forward on one batch: per-tensor differences, and both against a float64
evaluation of the first SA1 layer computed with torch from the same gathered rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ga_ddpg_amd import engine, hip
from tests.test_gpu_encoder import _feature_net, _geometry, _run_encoder


def main(B=64):
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    dev = torch.device("cuda")
    cfg = load_cfg("ddpg_td3_aux.yaml")
    mem = BaseMemory(400, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 400, seed=11)
    batch = sample_valid_batch(mem, B, np.random.default_rng(3))
    net = _feature_net()
    geo = _geometry(B)
    geo.run(torch.from_numpy(batch["point_state_batch"]).cuda())
    action = torch.from_numpy(batch["action_batch"]).cuda()
    probe = torch.ones(B, 512, device=dev)
    outs = {}
    for mode in (1, 0):
        hip.set_option("fwd_stream", mode)
        enc = engine.EncoderNet(net.value_encoder, dev)
        slot = engine.EncoderSlot(geo, enc, dev)
        z = _run_encoder(enc, slot, action, probe, True)
        n1 = int(geo.rows[0]["n"].item())
        o = enc.bn_off[0]
        outs[mode] = dict(Z1=slot.Z[0][0][:n1].clone(), Z2=slot.Z[0][1][:n1].clone(), Z3=slot.Z[0][2][:n1].clone(),
                          F1=slot.F[0].clone(), F2=slot.F[1].clone(), F3=slot.F[2].clone(), z=z.clone(),
                          mean=slot.mean.clone(), istd=slot.istd.clone(), grad=enc.flat.grad.clone())
    hip.set_option("fwd_stream", 1)
    for k in outs[1]:
        a, b = outs[1][k].double(), outs[0][k].double()
        d = (a - b).abs()
        print("%-5s max|stream-tiled| %.3e  median %.3e   (max|.| %.3e)" % (k, d.max(), d.median(), b.abs().max()))
    # float64 truth of layer 1 from the gathered inputs
    r = geo.rows[0]
    n1 = int(r["n"].item())
    pt, grp, w = r["pt"][:n1].long(), r["grp"][:n1].long(), r["w"][:n1].double()
    xyz = geo.xyz.view(-1, 3).double()
    ctr = geo.new_xyz1.view(-1, 3).double()
    feat = geo.feat0.view(-1, geo.feat0.shape[-1]).double()
    m = enc.mats[0]
    Wp = enc.flat.packed[m.w_off:m.w_off + m.n_out * m.Kp].view(m.n_out, m.Kp).double() if hasattr(m, "w_off") else None
    if Wp is not None:
        X = torch.zeros(n1, m.Kp, dtype=torch.float64, device=dev)
        fc = feat.shape[1]
        X[:, :fc] = feat[pt]
        X[:, fc:fc + 3] = xyz[pt] - ctr[grp]
        X[:, fc + 3:fc + 9] = action.double()[grp // 32]
        Zt = X @ Wp.t()
        for mode in (1, 0):
            d = (outs[mode]["Z1"].double() - Zt).abs()
            print("Z1 vs float64: mode %d max %.3e median %.3e" % (mode, d.max(), d.median()))
        mu = (w[:, None] * Zt).sum(0) / w.sum()
        for mode in (1, 0):
            print("BN mean layer1 vs float64: mode %d max %.3e" % (mode, (outs[mode]["mean"][:64].double() - mu).abs().max()))


if __name__ == "__main__":
    main()
