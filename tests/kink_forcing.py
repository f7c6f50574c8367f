"""Test infrastructure: impose the ReLU / max-pool DECISIONS of a HIP encoder pass on the CPU oracle.

The update step is piecewise smooth: every ReLU and every max-pool is a kink, and two float32 evaluations that round a
pre-activation to different sides of one reroute a whole row's gradient.  With ~1e7 pre-activations per encoder pass a few
always sit within rounding of zero, and the layers with few rows per channel turn one such tie into a broad shift of every
gradient upstream (measured: one flip at fc[1], 32 rows, moved every value-encoder tensor by 2 - 4e-3 of its scale; one at
SA3.l2, 1024 rows, by 3e-4).  That caps any free-running gradient comparison at ~1e-3 -- for the reference's own float32
run as much as for ours.  Here the oracle is evaluated (in float64 and float32) with the decisions the HIP pass actually
took -- ReLU masks from fmaf(z, scale, shift) > 0 on the saved raw outputs, pool winners from the saved arg-max -- so both
sides differentiate the SAME smooth function and what remains is arithmetic: summation order, the de-duplicated rows and
their weights, BatchNorm forward / backward, the scatter epilogues.  Tolerances drop from 5e-3 to ~1e-5."""
import torch

from oracle.pointnet2_ops import pointnet2_utils as pu


def decisions_from_slot(enc, slot):
    """enc: engine.EncoderNet, slot: engine.EncoderSlot after a forward pass -> decisions for forced_forward()"""
    geo = slot.geo
    B = slot.B
    dec = {"sa": [], "fc": []}
    for s in range(3):
        r = geo.rows[s]
        G = r["G"]
        n = int(r["n"].item())
        off = r["off"][:G + 1].long().cpu()
        S = (geo.sa1.nsample, geo.sa2.nsample, geo.M2)[s]
        M = (geo.M1, geo.M2, 1)[s]
        cnt = off[1:] - off[:-1]
        sl = torch.arange(S)[None, :]
        row_of = off[:-1, None] + torch.where(sl < cnt[:, None], sl, torch.zeros_like(sl))      # (G,S): padded slot -> row
        relu = []
        for l, m in enumerate(enc.sa_mats[s]):
            o = enc.bn_off[m.bn_index]
            C = m.n_out
            y = slot.Z[s][l][:n].double() * slot.scale[o:o + C].double() + slot.shift[o:o + C].double()   # == fmaf's sign
            mask = (y > 0).cpu()                                                                # (rows, C)
            relu.append(mask[row_of.reshape(-1)].reshape(B, M, S, C).permute(0, 3, 1, 2).contiguous())   # (B,C,M,S)
        C = enc.sa_mats[s][2].n_out
        win = slot.argmax[s].long().cpu() - off[:-1, None]                                      # (G,C): winner's slot
        assert int(win.min()) >= 0 and bool((win < cnt[:, None]).all())
        dec["sa"].append({"relu": relu, "pool": win.reshape(B, M, C).permute(0, 2, 1).contiguous()})
    for l, m in enumerate(enc.fc_mats):
        o = enc.bn_off[m.bn_index]
        y = slot.Zfc[l].double() * slot.scale[o:o + m.n_out].double() + slot.shift[o:o + m.n_out].double()
        dec["fc"].append((y > 0).cpu())
    return dec


def _forward_forced(fe, pc, value, D):
    """oracle.ref_step.PointFeature.forward + upstream's SA-module forward, with every ReLU / max-pool decision taken from D"""
    x = pc[..., 6:] if pc.shape[-1] != 1024 else pc
    c = fe.critic_input_dim if value else fe.policy_input_dim
    feats = x[:, :c].contiguous()
    xyz = feats.transpose(1, -1)[..., :3].contiguous()
    enc = fe.value_encoder if value else fe.encoder
    for s, sa in enumerate(enc[0]):
        new_xyz = None
        if sa.npoint is not None:
            fps_idx = pu.furthest_point_sample(xyz, sa.npoint)
            new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), fps_idx).transpose(1, 2).contiguous()
        h = sa.groupers[0](xyz, new_xyz, feats)
        seq = sa.mlps[0]
        for l in range(3):
            h = seq[3 * l + 1](seq[3 * l](h))                       # conv, BatchNorm2d (train mode, padded duplicates counted)
            h = h * D["sa"][s]["relu"][l].to(h.dtype)               # ReLU as decided by the HIP pass
        feats = h.gather(3, D["sa"][s]["pool"].unsqueeze(-1)).squeeze(-1)      # max-pool winner as decided by the HIP pass
        xyz = new_xyz
    fc = enc[1]
    z = fc[1](fc[0](feats.squeeze(-1))) * D["fc"][0].to(h.dtype)
    return fc[4](fc[3](z)) * D["fc"][1].to(h.dtype)


class forced_forward(object):
    """context manager: `decisions` = {("value" | "policy", call index): decisions_from_slot(...)}; the selected calls of
    the oracle's feature extractor run with those decisions, every other call as usual"""

    def __init__(self, fe, decisions):
        self.fe, self.decisions = fe, decisions

    def __enter__(self):
        fe, calls = self.fe, {"value": 0, "policy": 0}
        self.orig = fe.forward

        def fwd(pc, value=False):
            tag = "value" if value else "policy"
            k = calls[tag]
            calls[tag] += 1
            D = self.decisions.get((tag, k))
            return self.orig(pc, value) if D is None else _forward_forced(fe, pc, value, D)
        fe.forward = fwd
        self.calls = calls
        return self

    def __exit__(self, *a):
        self.fe.forward = self.orig
