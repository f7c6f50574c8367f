"""Stand-alone evaluation of one `PointnetSAModule` through the fused libgaddpg path
(pointnet2_ops.pointnet2_modules.PointnetSAModule.forward): FPS -> ball query -> de-duplicated rows
-> gather + 3 x (1x1 conv, BatchNorm, ReLU) as FP32-MFMA GEMMs -> segment max-pool.

Forward only (train- or eval-mode BatchNorm, running statistics updated in train mode).  Training goes
through the fused update step (core.agent / runtime), which owns the backward plans; calling this with
tensors that require grad raises instead of silently detaching.
Upstream semantics mirrored: _PointnetSAModuleBase.forward: returns
(new_xyz (B,npoint,3) | None, new_features (B, mlp[-1], npoint | 1)).
"""
import torch

from . import engine, hip
from .engine import BN_EPS, BN_MOMENTUM, FlatNet, MatSpec, Plan, _fwd_args, _ptr


class _StageNet(object):
    """packed parameters of one SA module (same attribute names as engine.EncoderNet)"""

    def __init__(self, mod, device):
        seq = mod.mlps[0]
        c_feat = seq[0].weight.shape[1] - 3
        if c_feat % 4 != 0 or c_feat < 4:
            raise NotImplementedError("stand-alone PointnetSAModule.forward needs a feature count that is a "
                                      "positive multiple of 4 (got %d)" % c_feat)
        self.c_feat = c_feat
        self.mats = [MatSpec(seq[0].weight, None, seq[1], gather_feat_c=c_feat), MatSpec(seq[3].weight, None, seq[4]),
                     MatSpec(seq[6].weight, None, seq[7])]
        for i, m in enumerate(self.mats):
            m.bn_index = i
        self.flat = FlatNet(list(mod.named_parameters()), self.mats, device)
        tot = sum(m.n_out for m in self.mats)
        self.running_mean = torch.zeros(tot, dtype=torch.float32, device=device)
        self.running_var = torch.ones(tot, dtype=torch.float32, device=device)
        self.batches_tracked = torch.zeros(3, dtype=torch.int64, device=device)
        self.bn_off, o = [], 0
        for i, m in enumerate(self.mats):
            self.bn_off.append(o)
            with torch.no_grad():
                self.running_mean[o:o + m.n_out].copy_(m.bn.running_mean.to(device))
                self.running_var[o:o + m.n_out].copy_(m.bn.running_var.to(device))
            m.bn.running_mean = self.running_mean[o:o + m.n_out]
            m.bn.running_var = self.running_var[o:o + m.n_out]
            self.batches_tracked[i] = int(m.bn.num_batches_tracked)
            m.bn.num_batches_tracked = self.batches_tracked[i]
            o += m.n_out
        self.bn_total = tot


class _StageRun(object):
    def __init__(self, net, mod, B, N, device):
        f32 = dict(dtype=torch.float32, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        self.B, self.N = B, N
        self.group_all = mod.npoint is None
        M = 1 if self.group_all else mod.npoint
        S = N if self.group_all else mod.nsample
        self.M, self.S = M, S
        G = B * M
        cap = G * S
        self.xyz = torch.empty(B, N, 3, **f32)
        self.feat = torch.empty(B * N, net.c_feat, **f32)
        self.new_xyz = torch.empty(B, M, 3, **f32)
        self.fps = torch.empty(B, M, **i32)
        self.idx = torch.empty(B, M, S, **i32) if not self.group_all else None
        self.cnt = torch.empty(B, M, **i32) if not self.group_all else None
        self.rows = dict(G=G, cap=cap, off=torch.zeros(G + 1, **i32), pt=torch.zeros(cap, **i32),
                         grp=torch.zeros(cap, **i32), w=torch.zeros(cap, **f32), n=torch.zeros(1, **i32))
        if self.group_all:
            r = self.rows
            hip.call("gad_rows_group_all", B, N, r["off"], r["pt"], r["grp"], r["w"], r["n"])
        self.Z = [torch.empty(cap, m.n_out, **f32) for m in net.mats]
        c_out = net.mats[2].n_out
        self.F = torch.empty(G, c_out, **f32)
        self.argmax = torch.empty(G, c_out, dtype=torch.int32, device=device)
        tot = net.bn_total
        self.tot = tot
        self.stats = torch.zeros(hip.STAT_REPLICAS * 2 * tot, dtype=torch.float64, device=device)
        self.scale = torch.empty(tot, **f32)
        self.shift = torch.empty(tot, **f32)
        self.mean = torch.empty(tot, **f32)
        self.istd = torch.empty(tot, **f32)
        self.count = float(G * S)
        self.plans = {t: self._plan(net, mod, t) for t in (True, False)}

    def _plan(self, net, mod, train):
        plan = Plan()
        B, N, M, S, r, tot = self.B, self.N, self.M, self.S, self.rows, self.tot
        if not self.group_all:
            plan.call("gad_furthest_point_sampling", self.xyz, B, N, M, self.fps, self.new_xyz)
            plan.call("gad_ball_query", self.new_xyz, self.xyz, B, N, M, float(mod.radius), S, self.idx, self.cnt)
            plan.call("gad_rows_from_ball_query", self.idx, self.cnt, B * M, M, N, S, r["off"], r["pt"], r["grp"],
                      r["w"], r["n"])
        plan.zero(self.stats)
        for l, m in enumerate(net.mats):
            o = net.bn_off[l]
            rows_kw = dict(n_rows_dev=_ptr(r["n"]), n_rows=r["cap"], row_w=_ptr(r["w"]))
            if l == 0:
                inp = dict(mode=1, c_in=m.k_in, src_xyz=_ptr(self.xyz), ctr_xyz=None if self.group_all else _ptr(self.new_xyz),
                           feat=_ptr(self.feat), feat_c=net.c_feat, action=None, act_c=0, grp_per_sample=M,
                           row_pt=_ptr(r["pt"]), row_grp=_ptr(r["grp"]))
            else:
                pm = net.mats[l - 1]
                po = net.bn_off[l - 1]
                inp = dict(mode=0, zin=_ptr(self.Z[l - 1]), zin_pitch=pm.n_out, c_in=pm.n_out, scale=_ptr(self.scale, po),
                           shift=_ptr(self.shift, po), relu=1)
            a = _fwd_args(W=net.flat.p_w(m), Kp=m.Kp, n_out=[m.n_out], zout=_ptr(self.Z[l]), zout_pitch=m.n_out,
                          stat_sum=_ptr(self.stats, o, 8) if train else None,
                          stat_sq=_ptr(self.stats, tot + o, 8) if train else None, stat_stride=2 * tot, **rows_kw, **inp)
            plan.call_struct("gad_gemm_fwd", a)
            if train:
                plan.call("gad_bn_finalize", _ptr(self.stats, o, 8), _ptr(self.stats, tot + o, 8), 2 * tot,
                          net.flat.p_gamma(m), net.flat.p_beta(m), m.n_out, hip.Dbl(self.count), BN_EPS, BN_MOMENTUM,
                          _ptr(net.running_mean, o), _ptr(net.running_var, o), _ptr(self.scale, o), _ptr(self.shift, o),
                          _ptr(self.mean, o), _ptr(self.istd, o))
            else:
                plan.call("gad_bn_eval_affine", net.flat.p_gamma(m), net.flat.p_beta(m), _ptr(net.running_mean, o),
                          _ptr(net.running_var, o), m.n_out, BN_EPS, _ptr(self.scale, o), _ptr(self.shift, o))
        m = net.mats[2]
        o = net.bn_off[2]
        plan.call("gad_segment_pool", self.Z[2], m.n_out, m.n_out, _ptr(self.scale, o), _ptr(self.shift, o), r["off"],
                  r["G"], self.F, self.argmax)
        return plan


def sa_module_forward(mod, xyz, features):
    hip.require_cuda(xyz, features)
    if features is None:
        raise NotImplementedError("PointnetSAModule without input features is not used by GA-DDPG")
    if torch.is_grad_enabled() and features.requires_grad:
        raise RuntimeError("stand-alone PointnetSAModule.forward is inference-only; train through "
                           "Agent.update_parameters (fused forward+backward)")
    B, N, _ = xyz.shape
    dev = xyz.device
    rt = mod.__dict__.get("_gad_rt")
    if rt is None:
        rt = {"net": _StageNet(mod, dev)}
        object.__setattr__(mod, "_gad_rt", rt)
    net = rt["net"]
    if features.shape[1] != net.c_feat:
        raise RuntimeError("features have %d channels, module expects %d" % (features.shape[1], net.c_feat))
    key = (B, N)
    if key not in rt:
        rt[key] = _StageRun(net, mod, B, N, dev)
    run = rt[key]
    run.xyz.copy_(xyz)
    run.feat.view(B, N, net.c_feat).copy_(features.transpose(1, 2))      # (B,C,N) -> point-major
    run.plans[bool(mod.training)].run()
    if mod.training:
        net.batches_tracked += 1
    c_out = net.mats[2].n_out
    out = run.F.view(B, run.M, c_out).transpose(1, 2).contiguous()
    return (None if run.group_all else run.new_xyz.clone()), out
