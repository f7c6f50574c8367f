"""Launch plans for the actor / critic heads (reference core/networks.py:253-300 QNetwork,
:303-371 GaussianPolicy) on top of an encoder slot.  The three critic trunks (Q1, Q2, aux) share
their input, so layer 1 is ONE GEMM over the concatenated (768 x 520) weight and layers 2/3 are
grouped GEMMs (blockIdx.z = trunk); the policy's mean and aux heads are one (13 x 264) GEMM."""
import torch

from . import hip
from .engine import FlatNet, MatSpec, Plan, _bn_vec, _dz, _fwd_args, _ptr, dw_workspace


class CriticNet(object):
    def __init__(self, module, device):
        self.module = module
        self.aux = getattr(module, "extra_pred_dim", 0) > 0
        m = module
        l1 = [MatSpec(m.linear1.weight, m.linear1.bias), MatSpec(m.linear4.weight, m.linear4.bias)]
        l2 = [MatSpec(m.linear2.weight, m.linear2.bias), MatSpec(m.linear5.weight, m.linear5.bias)]
        l3 = [MatSpec(m.linear3.weight, m.linear3.bias), MatSpec(m.linear6.weight, m.linear6.bias)]
        if self.aux:
            l1.append(MatSpec(m.linear7.weight, m.linear7.bias))
            l2.append(MatSpec(m.linear8.weight, m.linear8.bias))
            l3.append(MatSpec(m.extra_pred.weight, m.extra_pred.bias))
        self.l1, self.l2, self.l3 = l1, l2, l3
        self.flat = FlatNet(list(module.named_parameters()), l1 + l2 + l3, device)
        self.ng = len(l1)
        self.hidden = l1[0].n_out
        self.width = self.hidden * self.ng                  # 768
        self.n_last = [x.n_out for x in l3]                 # [1, 1, 7]
        self.out_off = [0, 1, 2][:self.ng]
        self.n_outputs = 9


class PolicyNet(object):
    def __init__(self, module, device):
        self.module = module
        m = module
        self.l1 = MatSpec(m.linear1.weight, m.linear1.bias)
        self.l2 = MatSpec(m.linear2.weight, m.linear2.bias)
        self.mean = MatSpec(m.mean.weight, m.mean.bias)
        self.extra = MatSpec(m.extra_pred.weight, m.extra_pred.bias)
        self.log_std = MatSpec(m.log_std_linear.weight, m.log_std_linear.bias)
        # log_std_linear feeds nothing on the update path: never receives a gradient, never stepped.  Without
        # policy_aux (extra_pred_dim 1, reference core/agent.py:31-36) the extra head feeds no loss either: its .grad
        # stays None in the reference and torch's Adam skips it (no weight decay) -- same here.
        never = ("log_std_linear",) + (("extra_pred",) if self.extra.n_out != 7 else ())
        self.flat = FlatNet(list(module.named_parameters()), [self.l1, self.l2, self.mean, self.extra, self.log_std],
                            device, never_trained=never)
        self.hidden = self.l1.n_out
        self.n_heads = self.mean.n_out + self.extra.n_out   # 6 + 7 = 13 (extra_pred_dim 7) or 6 + 1
        self.extra_dim = self.extra.n_out


class HeadSlot(object):
    """activations / gradients of one head evaluation on B rows"""

    def __init__(self, B, width, n_out, device):
        f32 = dict(dtype=torch.float32, device=device)
        self.B = B
        self.Z1 = torch.empty(B, width, **f32)
        self.Z2 = torch.empty(B, width, **f32)
        self.out = torch.zeros(B, n_out, **f32)
        self.g_out = torch.zeros(B, n_out, **f32)
        self.G2 = torch.empty(B, width, **f32)
        self.G1 = torch.empty(B, width, **f32)
        self.g_feat = torch.empty(B, 512, **f32)


def _feat_input(enc, eslot, time):
    """heads' layer-1 input: [relu(bn(Zfc2)) (512), time, 1]; a plain feature tensor (runtime._FeatureSource:
    `input_relu` 0, identity scale / shift) is taken as it is."""
    fc2 = enc.fc_mats[1]
    return dict(n_rows=eslot.B, mode=0, zin=_ptr(eslot.Zfc[1]), zin_pitch=fc2.n_out, c_in=fc2.n_out,
                scale=_bn_vec(eslot, enc, fc2, "scale"), shift=_bn_vec(eslot, enc, fc2, "shift"),
                relu=getattr(enc, "input_relu", 1), extra=_ptr(time), ones_col=fc2.n_out + 1)


def _hidden_input(B, z, pitch, hidden, offs):
    return dict(n_rows=B, mode=0, zin=_ptr(z), zin_pitch=pitch, c_in=hidden, relu=1, ones_col=hidden,
                zin_off=offs)


def plan_critic_forward(cr, hs, enc, eslot, time):
    plan = Plan()
    B, H, ng = hs.B, cr.hidden, cr.ng
    fl = cr.flat
    offs = [i * H for i in range(ng)]
    plan.call_struct("gad_gemm_fwd", _fwd_args(W=fl.p_w(cr.l1[0]), Kp=cr.l1[0].Kp, n_out=[cr.width], zout=_ptr(hs.Z1),
                                               zout_pitch=cr.width, **_feat_input(enc, eslot, time)))
    plan.call_struct("gad_gemm_fwd", _fwd_args(W=_ptr(fl.packed), Kp=cr.l2[0].Kp, n_groups=ng,
                                               w_off=[m.w_off for m in cr.l2], n_out=[H] * ng, out_off=offs,
                                               zout=_ptr(hs.Z2), zout_pitch=cr.width,
                                               **_hidden_input(B, hs.Z1, cr.width, H, offs)))
    plan.call_struct("gad_gemm_fwd", _fwd_args(W=_ptr(fl.packed), Kp=cr.l3[0].Kp, n_groups=ng,
                                               w_off=[m.w_off for m in cr.l3], n_out=cr.n_last, out_off=cr.out_off,
                                               zout=_ptr(hs.out), zout_pitch=cr.n_outputs,
                                               **_hidden_input(B, hs.Z2, cr.width, H, offs)))
    return plan


def plan_critic_backward(cr, hs, enc, eslot, time, want_dw=True, dw_lane=1, join_dw=False):
    """consumes hs.g_out (B,9); leaves dLoss/dfeature in hs.g_feat (B,512) and the BN-backward sums
    of the encoder's last BatchNorm in eslot.bstats (which the caller must have zeroed).
    dw_lane: the side stream the head's weight-gradient GEMMs are forked onto (they feed nothing but the optimiser: off the
    dX chain, like the encoder's) = the lane of the encoder backward that follows (critic 1, actor 2 -- the two backward
    passes may run concurrently); that plan's final join covers them.  join_dw: join here instead (a caller that reads the
    gradient arena right after the head)."""
    plan = Plan()
    B, H, ng = hs.B, cr.hidden, cr.ng
    fl = cr.flat
    offs = [i * H for i in range(ng)]
    tot = eslot.tot

    def dw(dz, dz_off, mats, inp):
        if not want_dw:
            return
        a = hip.GemmDwArgs()
        a.inp = _fwd_args(Kp=mats[0].Kp, n_groups=len(mats), w_off=[m.w_off for m in mats],
                          n_out=[m.n_out for m in mats], **inp)
        a.dz = dz
        for i, o in enumerate(dz_off):
            a.dz_off[i] = o
        a.gacc = _ptr(fl.gacc)
        ws = dw_workspace(fl.device, lane=100 + dw_lane)     # never shared with a forked encoder dW (lanes 1, 2)
        a.partial, a.partial_elems = _ptr(ws), ws.numel()
        lane = dw_lane if CONCURRENT_DW_HEADS() else 0
        if lane:
            plan.fork(lane)
        plan.call_struct("gad_gemm_dw", a, side=lane)

    def dx(dz, dz_off, mats, k_valid, gout, gout_off, **epi):
        a = hip.GemmDxArgs()
        a.n_rows = B
        a.dz = dz
        a.n_groups = len(mats)
        for i, m in enumerate(mats):
            a.dz_off[i] = dz_off[i]
            a.w_off[i] = m.w_off
            a.n_out[i] = m.n_out
            a.gout_off[i] = gout_off[i]
        a.W = _ptr(fl.packed)
        a.Kp = mats[0].Kp
        a.k_valid = k_valid
        a.epilogue = 0
        a.gout = _ptr(gout)
        a.gout_pitch = gout.shape[1]
        a.grp_per_sample = 1
        for k, v in epi.items():
            setattr(a, k, v)
        plan.call_struct("gad_gemm_dx", a)

    # layer 3 (no activation after it)
    d3 = _dz(z=None, z_pitch=0, relu=0, gmode=0, G=_ptr(hs.g_out), g_pitch=cr.n_outputs, c=cr.n_outputs)
    dw(d3, cr.out_off, cr.l3, _hidden_input(B, hs.Z2, cr.width, H, offs))
    dx(d3, cr.out_off, cr.l3, H, hs.G2, offs)
    # layer 2
    d2 = _dz(z=_ptr(hs.Z2), z_pitch=cr.width, relu=1, gmode=0, G=_ptr(hs.G2), g_pitch=cr.width, c=cr.width)
    dw(d2, offs, cr.l2, _hidden_input(B, hs.Z1, cr.width, H, offs))
    dx(d2, offs, cr.l2, H, hs.G1, offs)
    # layer 1 (concatenated): one group of width 768
    d1 = _dz(z=_ptr(hs.Z1), z_pitch=cr.width, relu=1, gmode=0, G=_ptr(hs.G1), g_pitch=cr.width, c=cr.width)
    cat = _Cat(cr.l1)
    dw(d1, [0], [cat], _feat_input(enc, eslot, time))
    fc2 = enc.fc_mats[1]
    o = enc.bn_off[fc2.bn_index]
    dx(d1, [0], [cat], fc2.n_out, hs.g_feat, [0], zprev=_ptr(eslot.Zfc[1]), zprev_pitch=fc2.n_out,
       prev_scale=_bn_vec(eslot, enc, fc2, "scale"), prev_shift=_bn_vec(eslot, enc, fc2, "shift"),
       prev_mean=_bn_vec(eslot, enc, fc2, "mean"), prev_istd=_bn_vec(eslot, enc, fc2, "istd"),
       prev_dbeta=_ptr(eslot.bstats, o, 8), prev_dgamma=_ptr(eslot.bstats, tot + o, 8), stat_stride=2 * tot,
       store_masked=1)          # g_feat carries the encoder's last ReLU mask (engine.plan_encoder_backward: premasked)
    if join_dw and want_dw and CONCURRENT_DW_HEADS():
        plan.join(dw_lane)
    return plan


def CONCURRENT_DW_HEADS():
    from . import engine
    return engine.CONCURRENT_DW            # (measured: 299 vs 296 steps/s with the heads' dW GEMMs in line on the main stream)


class _Cat(object):
    """consecutive packed matrices with the same Kp seen as one (sum n_out, Kp) matrix"""

    def __init__(self, mats):
        self.w_off, self.Kp = mats[0].w_off, mats[0].Kp
        self.n_out = sum(m.n_out for m in mats)
        for a, b in zip(mats[:-1], mats[1:]):
            assert b.w_off == a.w_off + a.n_out * a.Kp and a.Kp == b.Kp


def plan_policy_forward(po, hs, enc, eslot, time, with_log_std=False):
    """hs.out (B, 6 + extra) = [mean | extra]; with_log_std: (B, 6 + extra + 6) = [mean | extra | log_std] (the
    three output matrices are consecutive in the packed layout: one GEMM either way)"""
    plan = Plan()
    B, H = hs.B, po.hidden
    fl = po.flat
    plan.call_struct("gad_gemm_fwd", _fwd_args(W=fl.p_w(po.l1), Kp=po.l1.Kp, n_out=[H], zout=_ptr(hs.Z1), zout_pitch=H,
                                               **_feat_input(enc, eslot, time)))
    plan.call_struct("gad_gemm_fwd", _fwd_args(W=fl.p_w(po.l2), Kp=po.l2.Kp, n_out=[H], zout=_ptr(hs.Z2), zout_pitch=H,
                                               **_hidden_input(B, hs.Z1, H, H, [0])))
    cat = _Cat([po.mean, po.extra] + ([po.log_std] if with_log_std else []))
    assert hs.out.shape[1] >= cat.n_out
    plan.call_struct("gad_gemm_fwd", _fwd_args(W=_ptr(fl.packed, cat.w_off), Kp=cat.Kp, n_out=[cat.n_out],
                                               zout=_ptr(hs.out), zout_pitch=hs.out.shape[1],
                                               **_hidden_input(B, hs.Z2, H, H, [0])))
    return plan


def plan_policy_backward(po, hs, enc, eslot, time, dw_lane=2):
    """consumes hs.g_out (B,13); leaves dLoss/dfeature in hs.g_feat and fc[1]'s BN sums in eslot.bstats."""
    plan = Plan()
    B, H = hs.B, po.hidden
    fl = po.flat
    tot = eslot.tot
    cat = _Cat([po.mean, po.extra])
    nh = cat.n_out

    def dw(dz, m, inp):
        a = hip.GemmDwArgs()
        a.inp = _fwd_args(Kp=m.Kp, n_out=[m.n_out], w_off=[m.w_off], **inp)
        a.dz = dz
        a.gacc = _ptr(fl.gacc)
        ws = dw_workspace(fl.device, lane=100 + dw_lane)
        a.partial, a.partial_elems = _ptr(ws), ws.numel()
        lane = dw_lane if CONCURRENT_DW_HEADS() else 0
        if lane:
            plan.fork(lane)                       # weight gradients off the dX chain (the encoder backward's final join covers them)
        plan.call_struct("gad_gemm_dw", a, side=lane)

    def dx(dz, m, k_valid, gout, **epi):
        a = hip.GemmDxArgs()
        a.n_rows = B
        a.dz = dz
        a.n_groups = 1
        a.w_off[0] = m.w_off
        a.n_out[0] = m.n_out
        a.W = _ptr(fl.packed)
        a.Kp = m.Kp
        a.k_valid = k_valid
        a.epilogue = 0
        a.gout = _ptr(gout)
        a.gout_pitch = gout.shape[1]
        a.grp_per_sample = 1
        for k, v in epi.items():
            setattr(a, k, v)
        plan.call_struct("gad_gemm_dx", a)

    d3 = _dz(z=None, z_pitch=0, relu=0, gmode=0, G=_ptr(hs.g_out), g_pitch=hs.g_out.shape[1], c=nh)
    dw(d3, cat, _hidden_input(B, hs.Z2, H, H, [0]))
    dx(d3, cat, H, hs.G2)
    d2 = _dz(z=_ptr(hs.Z2), z_pitch=H, relu=1, gmode=0, G=_ptr(hs.G2), g_pitch=H, c=H)
    dw(d2, po.l2, _hidden_input(B, hs.Z1, H, H, [0]))
    dx(d2, po.l2, H, hs.G1)
    d1 = _dz(z=_ptr(hs.Z1), z_pitch=H, relu=1, gmode=0, G=_ptr(hs.G1), g_pitch=H, c=H)
    dw(d1, po.l1, _feat_input(enc, eslot, time))
    fc2 = enc.fc_mats[1]
    o = enc.bn_off[fc2.bn_index]
    dx(d1, po.l1, fc2.n_out, hs.g_feat, zprev=_ptr(eslot.Zfc[1]), zprev_pitch=fc2.n_out,
       prev_scale=_bn_vec(eslot, enc, fc2, "scale"), prev_shift=_bn_vec(eslot, enc, fc2, "shift"),
       prev_mean=_bn_vec(eslot, enc, fc2, "mean"), prev_istd=_bn_vec(eslot, enc, fc2, "istd"),
       prev_dbeta=_ptr(eslot.bstats, o, 8), prev_dgamma=_ptr(eslot.bstats, tot + o, 8), stat_stride=2 * tot,
       store_masked=1)          # g_feat carries the encoder's last ReLU mask (engine.plan_encoder_backward: premasked)
    return plan
