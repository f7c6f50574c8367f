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
        dec["sa"].append({"relu": relu, "pool": win.reshape(B, M, C).permute(0, 2, 1).contiguous(), "cnt": cnt.reshape(B, M)})
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


def _forward_recording(fe, pc, value, rec):
    """the same forward, free-running: every ReLU / max-pool takes its own decision, which is recorded together with the
    quantity it was taken on (post-BatchNorm pre-activation y; pooled activations) -> rec = decisions + "y" / "pooled" """
    x = pc[..., 6:] if pc.shape[-1] != 1024 else pc
    c = fe.critic_input_dim if value else fe.policy_input_dim
    feats = x[:, :c].contiguous()
    xyz = feats.transpose(1, -1)[..., :3].contiguous()
    enc = fe.value_encoder if value else fe.encoder
    rec["sa"], rec["fc"], rec["y_sa"], rec["y_fc"], rec["pooled"] = [], [], [], [], []
    for s, sa in enumerate(enc[0]):
        new_xyz = None
        if sa.npoint is not None:
            fps_idx = pu.furthest_point_sample(xyz, sa.npoint)
            new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), fps_idx).transpose(1, 2).contiguous()
        h = sa.groupers[0](xyz, new_xyz, feats)
        seq = sa.mlps[0]
        relu, ys = [], []
        for l in range(3):
            y = seq[3 * l + 1](seq[3 * l](h))
            ys.append(y.detach())
            relu.append((y > 0).detach())
            h = torch.relu(y)
        win = h.argmax(dim=3)                                       # first maximum (torch.max_pool2d's rule on CPU)
        rec["sa"].append({"relu": relu, "pool": win.detach()})
        rec["y_sa"].append(ys)
        rec["pooled"].append(h.detach())
        feats = h.max(dim=3).values
        xyz = new_xyz
    fc = enc[1]
    y1 = fc[1](fc[0](feats.squeeze(-1)))
    y2 = fc[4](fc[3](torch.relu(y1)))
    rec["fc"] = [(y1 > 0).detach(), (y2 > 0).detach()]
    rec["y_fc"] = [y1.detach(), y2.detach()]
    return torch.relu(y2)


class recording_forward(object):
    """context manager: every call of the oracle's feature extractor runs free and leaves its decisions in
    self.records[("value" | "policy", call index)]"""

    def __init__(self, fe):
        self.fe, self.records = fe, {}

    def __enter__(self):
        fe, calls = self.fe, {"value": 0, "policy": 0}
        self.orig = fe.forward

        def fwd(pc, value=False):
            tag = "value" if value else "policy"
            k = calls[tag]
            calls[tag] += 1
            rec = self.records[(tag, k)] = {}
            return _forward_recording(fe, pc, value, rec)
        fe.forward = fwd
        return self

    def __exit__(self, *a):
        self.fe.forward = self.orig


def decision_differences(hip_dec, rec):
    """compare the HIP pass's decisions with a free-running oracle pass's own: -> dict(n, n_diff, worst) for the ReLU
    masks (worst = largest |pre-activation| / channel scale among the differing entries, measured in the oracle) and for
    the pool winners (worst = largest relative gap between the oracle's maximum and its activation at the HIP winner)"""
    n = n_diff = 0
    worst = 0.0
    for s in range(3):
        for l in range(3):
            a, b, y = hip_dec["sa"][s]["relu"][l], rec["sa"][s]["relu"][l], rec["y_sa"][s][l]
            d = a != b
            n += d.numel()
            n_diff += int(d.sum())
            if bool(d.any()):
                scale = y.abs().amax(dim=(0, 2, 3), keepdim=True).expand_as(y)
                worst = max(worst, float((y.abs() / scale)[d].max()))
    for l in range(2):
        a, b, y = hip_dec["fc"][l], rec["fc"][l], rec["y_fc"][l]
        d = a != b
        n += d.numel()
        n_diff += int(d.sum())
        if bool(d.any()):
            scale = y.abs().amax(dim=0, keepdim=True).expand_as(y)
            worst = max(worst, float((y.abs() / scale)[d].max()))
    pn = pn_diff = 0
    pworst = 0.0
    detail = []
    for s in range(3):
        a, b, h = hip_dec["sa"][s]["pool"], rec["sa"][s]["pool"], rec["pooled"][s]
        # upstream pads a group to nsample slots with copies of its first hit: a winner among the copies IS slot 0 (the
        # float64 convolution is not bitwise identical across the copies, its arg-max lands on one of them now and then)
        b = torch.where(b >= hip_dec["sa"][s]["cnt"][:, None, :], torch.zeros_like(b), b)
        d = a != b
        pn += d.numel()
        pn_diff += int(d.sum())
        if bool(d.any()):
            top = h.max(dim=3).values
            at = h.gather(3, a.unsqueeze(-1)).squeeze(-1)
            gap = (top - at) / (top.abs() + 1e-30)
            # all-zero groups (every activation clipped by the ReLU): any winner is the reference's "first" only by index
            live = d & (top > 0)
            exact = d & (top == at)
            detail.append("stage %d: %d differ, %d with a positive maximum, %d exactly tied in float64" % (s + 1, int(d.sum()), int(live.sum()), int(exact.sum())))
            if bool(live.any()):
                pworst = max(pworst, float(gap[live].max()))
    return dict(relu=dict(n=n, n_diff=n_diff, worst=worst), pool=dict(n=pn, n_diff=pn_diff, worst=pworst, detail="; ".join(detail)))
