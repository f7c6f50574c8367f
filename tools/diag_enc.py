"""Diagnostic: encoder backward vs golden for both encoders in either order; saves grads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os, sys
import numpy as np
import torch
from tests.test_gpu_encoder import _feature_net, _geometry, _run_encoder


def main(order):
    from ga_ddpg_amd import engine
    g = np.load("tests/golden/encoder_B16.npz")
    dev = torch.device("cuda")
    net = _feature_net()
    enc = engine.EncoderNet(net.encoder, dev)
    venc = engine.EncoderNet(net.value_encoder, dev)
    B = g["point_state"].shape[0]
    geo = _geometry(B)
    geo.run(torch.from_numpy(g["point_state"]).cuda())
    probe = torch.from_numpy(g["probe"]).cuda()
    action = torch.from_numpy(g["action"]).cuda()
    slot = engine.EncoderSlot(geo, enc, dev)
    vslot = engine.EncoderSlot(geo, venc, dev)
    for rep in range(2):
        for which in order:
            if which == "p":
                _run_encoder(enc, slot, None, probe, False)
            else:
                _run_encoder(venc, vslot, action, probe.flip(1).contiguous(), True)
        out = {n: p.grad.cpu().numpy() for n, p in net.named_parameters() if p.numel() < 40000}
        out["daction"] = vslot.daction.cpu().numpy()
        np.savez("gpurun_out/enc_grads_%s_%d.npz" % (order, rep), **out)


if __name__ == "__main__":
    os.makedirs("gpurun_out", exist_ok=True)
    main(sys.argv[1] if len(sys.argv) > 1 else "pv")
