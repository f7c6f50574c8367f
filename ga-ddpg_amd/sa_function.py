"""Stand-alone evaluation of one `PointnetSAModule` through the fused libgaddpg path
(pointnet2_ops.pointnet2_modules.PointnetSAModule.forward): FPS -> ball query -> de-duplicated rows
-> gather + 3 x (1x1 conv, BatchNorm, ReLU) as FP32-MFMA GEMMs -> segment max-pool.

Differentiable: the forward is wrapped in a torch.autograd.Function whose backward drives the same
dX / dW kernels as the fused update step (BatchNorm-backward with weighted statistics, max-pool routing
through the saved arg-max, scatter-add into the point features), so a network written against upstream's
`pointnet2_ops` -- the reference's own core/networks.py:66-81,217-220 -- trains through this package
with ordinary torch optimizers.  Gradients: wrt `features` and every parameter of the module; `xyz`
is geometry (FPS / ball-query indices are not differentiable upstream either; the recentred xyz channels'
gradient wrt the coordinates is not produced -- GA-DDPG never asks for it: xyz is detached).
Upstream semantics mirrored: _PointnetSAModuleBase.forward: returns
(new_xyz (B,npoint,3) | None, new_features (B, mlp[-1], npoint | 1)).
"""
import torch

from . import engine, hip
from .engine import BN_EPS, BN_MOMENTUM, FlatNet, MatSpec, Plan, _dz, _fwd_args, _ptr


class _StageNet(object):
    """packed parameters of one SA module (same attribute names as engine.EncoderNet)"""

    def __init__(self, mod, device):
        seq = mod.mlps[0]
        c_feat = seq[0].weight.shape[1] - 3
        if c_feat < 1:
            raise NotImplementedError("PointnetSAModule without input features is not used by GA-DDPG")
        self.c_feat = c_feat                                  # channels of the `features` argument
        self.c_pad = (c_feat + 3) // 4 * 4                    # point-major feature rows are padded to 16 bytes
        self.mats = [MatSpec(seq[0].weight, None, seq[1], gather_feat_c=self.c_pad), MatSpec(seq[3].weight, None, seq[4]),
                     MatSpec(seq[6].weight, None, seq[7])]
        for i, m in enumerate(self.mats):
            m.bn_index = i
        self.flat = FlatNet(list(mod.named_parameters()), self.mats, device)
        # split-bf16 weight mirrors (library option "mfma_split"): layers 2 / 3, and layer 1 when its feature block is whole K-tiles
        self.flat.enable_split([(m, (m.gather_feat_c if l == 0 else m.Kp)) for l, m in enumerate(self.mats)
                                if (l > 0 or m.gather_feat_c >= 32)])
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
        self.free = {}                                        # (B, N) -> activation sets returned by finished backwards


class _StageRun(object):
    """static buffers + launch plans of one forward (and its backward) at a fixed (B, N)"""

    def __init__(self, net, mod, B, N, device, with_backward=False):
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
        self.feat = torch.zeros(B * N, net.c_pad, **f32)      # padding columns stay zero
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
        # max-pool folded into the third GEMM's epilogue (packed arg-max keys, finished by gad_pool_finalize) where the entry point
        # takes it: rows a multiple of 4, channels a multiple of 64; else the separate segment max-pool operator
        self.fused_pool = cap % 4 == 0 and c_out % 64 == 0
        self.key = torch.zeros(G, c_out, dtype=torch.int64, device=device) if self.fused_pool else None
        # raw value of every arg-max row (the backward's BatchNorm sums read it instead of gathering z[argmax])
        self.zmax = torch.empty(G, c_out, **f32) if (self.fused_pool and with_backward) else None
        self.plans = {t: self._plan(net, mod, t) for t in (True, False)}
        self.bwd = None
        if with_backward:
            self.bstats = torch.zeros(hip.STAT_REPLICAS * 2 * tot, dtype=torch.float64, device=device)
            self.coef = torch.empty(3 * tot, **f32)
            self.G = [torch.empty(cap * net.mats[l].n_out, **f32) for l in (1, 0)]
            self.dF = torch.zeros(G, c_out, **f32)
            self.dfeat = torch.zeros(B * N, net.c_pad, **f32)
            self.grad = torch.zeros(net.flat.n, **f32)        # this call's parameter gradients (master layout)
            self.bwd = self._plan_backward(net)

    def _input(self, net, mod, l):
        """gad_gemm_fwd_args fields describing the input of layer l"""
        r = self.rows
        kw = dict(n_rows_dev=_ptr(r["n"]), n_rows=r["cap"], row_w=_ptr(r["w"]))
        if l == 0:
            kw.update(mode=1, c_in=net.c_pad + 3, src_xyz=_ptr(self.xyz),
                      ctr_xyz=None if self.group_all else _ptr(self.new_xyz), feat=_ptr(self.feat), feat_c=net.c_pad,
                      action=None, act_c=0, grp_per_sample=self.M, row_pt=_ptr(r["pt"]), row_grp=_ptr(r["grp"]))
        else:
            pm, po = net.mats[l - 1], net.bn_off[l - 1]
            kw.update(mode=0, zin=_ptr(self.Z[l - 1]), zin_pitch=pm.n_out, c_in=pm.n_out, scale=_ptr(self.scale, po),
                      shift=_ptr(self.shift, po), relu=1)
        return kw

    def _plan(self, net, mod, train):
        plan = Plan()
        B, N, M, S, r, tot = self.B, self.N, self.M, self.S, self.rows, self.tot
        if not self.group_all:
            plan.call("gad_furthest_point_sampling", self.xyz, B, N, M, self.fps, self.new_xyz)
            plan.call("gad_ball_query", self.new_xyz, self.xyz, B, N, M, float(mod.radius), S, self.idx, self.cnt)
            plan.call("gad_rows_from_ball_query", self.idx, self.cnt, B * M, M, N, S, r["off"], r["pt"], r["grp"],
                      r["w"], r["n"])
        plan.zero(self.stats)
        import ctypes as C
        for l, m in enumerate(net.mats):
            o = net.bn_off[l]
            kw = self._input(net, mod, l)
            kw.update(net.flat.split_fwd_kw(m))
            pooled = l == 2 and self.fused_pool
            if pooled:
                kw.update(pool_key=_ptr(self.key, 0, 8), pool_row_grp=_ptr(r["grp"]), pool_gamma=net.flat.p_gamma(m))
            a = _fwd_args(W=net.flat.p_w(m), Kp=m.Kp, n_out=[m.n_out], zout=_ptr(self.Z[l]), zout_pitch=m.n_out,
                          stat_sum=_ptr(self.stats, o, 8) if train else None,
                          stat_sq=_ptr(self.stats, tot + o, 8) if train else None, stat_stride=2 * tot, **kw)
            plan.call_struct("gad_gemm_fwd", a)
            if not train:
                plan.call("gad_bn_eval_affine", net.flat.p_gamma(m), net.flat.p_beta(m), _ptr(net.running_mean, o),
                          _ptr(net.running_var, o), m.n_out, BN_EPS, _ptr(self.scale, o), _ptr(self.shift, o))
            if pooled:
                # finalises the layer's train-mode BatchNorm (same arithmetic and outputs as gad_bn_finalize) or takes the eval-mode
                # scale / shift as given, then turns the keys into pooled features and arg-max rows
                stats = (_ptr(self.stats, o, 8), _ptr(self.stats, tot + o, 8), 2 * tot, hip.Dbl(self.count)) if train else (None, None, 0, hip.Dbl(1.0))
                run = (_ptr(net.running_mean, o), _ptr(net.running_var, o)) if train else (None, None)
                plan.call("gad_pool_finalize", _ptr(self.key, 0, 8), m.n_out, r["G"], r["off"], *stats, net.flat.p_gamma(m),
                          net.flat.p_beta(m), BN_EPS, BN_MOMENTUM, *run, _ptr(self.scale, o), _ptr(self.shift, o),
                          _ptr(self.mean, o), _ptr(self.istd, o), self.F, self.argmax, self.zmax)
            elif train:
                plan.call("gad_bn_finalize", _ptr(self.stats, o, 8), _ptr(self.stats, tot + o, 8), 2 * tot,
                          net.flat.p_gamma(m), net.flat.p_beta(m), m.n_out, hip.Dbl(self.count), BN_EPS, BN_MOMENTUM,
                          _ptr(net.running_mean, o), _ptr(net.running_var, o), _ptr(self.scale, o), _ptr(self.shift, o),
                          _ptr(self.mean, o), _ptr(self.istd, o))
        if not self.fused_pool:
            m = net.mats[2]
            o = net.bn_off[2]
            # (gad_segment_pool: torch's "first maximal activation" arg-max; the fused form takes the largest raw value, which is the
            # same row unless two different raw values round to one activation)
            plan.call("gad_segment_pool", self.Z[2], m.n_out, m.n_out, _ptr(self.scale, o), _ptr(self.shift, o),
                      r["off"], r["G"], self.F, self.argmax)
        return plan

    def _plan_backward(self, net):
        """consumes self.dF (G, c_out) = dLoss/d(pooled output); leaves dLoss/dfeatures in self.dfeat (point-major,
        padded) and this call's parameter gradients in self.grad.  Train-mode BatchNorm only (batch statistics)."""
        plan = Plan()
        r, tot = self.rows, self.tot
        fl = net.flat
        plan.zero_multi([fl.gacc, self.bstats, self.dfeat])
        m1, m2, m3 = net.mats
        o1, o2, o3 = net.bn_off

        def vec(which, o):
            return _ptr(getattr(self, which), o)

        def bn_dz(m, o, z, accumulate, G=None, pooled=False):
            d = dict(z=_ptr(z), z_pitch=m.n_out, scale=vec("scale", o), shift=vec("shift", o), relu=1, premasked=1,
                     row_w=_ptr(r["w"]), c=m.n_out)
            if not accumulate:            # first use of the layer's coefficients (the dW precedes the dX below)
                plan.call("gad_bn_bwd_coef", _ptr(self.bstats, o, 8), _ptr(self.bstats, tot + o, 8), 2 * tot,
                          vec("scale", o), vec("mean", o), vec("istd", o), m.n_out, hip.Dbl(self.count),
                          _ptr(self.coef, o), _ptr(self.coef, tot + o), _ptr(self.coef, 2 * tot + o),
                          _ptr(fl.gacc, m.g_off, 8), _ptr(fl.gacc, m.b_off, 8))
            d["coefP"], d["coefQ"], d["coefS"] = _ptr(self.coef, o), _ptr(self.coef, tot + o), _ptr(self.coef, 2 * tot + o)
            if pooled:
                d.update(gmode=1, argmax=_ptr(self.argmax), dout=_ptr(self.dF), row_grp=_ptr(r["grp"]))
            else:
                d.update(gmode=0, G=_ptr(G), g_pitch=m.n_out)
            return _dz(**d)

        def dw_args(l, dz, m):
            a = hip.GemmDwArgs()
            a.inp = _fwd_args(Kp=m.Kp, n_out=[m.n_out], w_off=[m.w_off], **self._input(net, None, l))
            a.dz = dz
            a.gacc = _ptr(fl.gacc)
            ws = engine.dw_workspace(fl.device, lane=0)
            a.partial, a.partial_elems = _ptr(ws), ws.numel()
            return a

        def dw(l, dz, m):
            plan.call_struct("gad_gemm_dw", dw_args(l, dz, m))

        def bwd(l, dz, m, k_valid, **epi):
            """dW and dX of layer l in one call (gad_gemm_bwd): SA1-like shapes (64 input channels, >= 32768 rows) stream dZ once
            for both products, every other shape runs the same two launches as dw() + dx()"""
            import ctypes as C
            aw, a = dw_args(l, dz, m), dx_args(dz, m, k_valid, **epi)
            plan.keep.extend([a, aw])
            plan.call("gad_gemm_bwd", a, aw)

        def dx(dz, m, k_valid, **epi):
            plan.call_struct("gad_gemm_dx", dx_args(dz, m, k_valid, **epi))

        def dx_args(dz, m, k_valid, **epi):
            a = hip.GemmDxArgs()
            a.n_rows_dev, a.n_rows = _ptr(r["n"]), r["cap"]
            a.dz = dz
            a.n_groups = 1
            a.n_out[0] = m.n_out
            a.W = fl.p_w(m)
            a.Kp = m.Kp
            a.k_valid = k_valid
            a.grp_per_sample = 1
            for k, v in dict(epi, **fl.split_t_kw(m)).items():
                setattr(a, k, v)
            return a

        def prev_stats(pm, po, zprev):
            return dict(zprev=_ptr(zprev), zprev_pitch=pm.n_out, prev_scale=vec("scale", po), prev_shift=vec("shift", po),
                        prev_mean=vec("mean", po), prev_istd=vec("istd", po), prev_dbeta=_ptr(self.bstats, po, 8),
                        prev_dgamma=_ptr(self.bstats, tot + po, 8), stat_stride=2 * tot, store_masked=1)

        # gradients travel ReLU-masked between the layers (store_masked / premasked)
        plan.call("gad_pool_bwd_stats", self.dF, self.argmax, r["G"], m3.n_out, self.Z[2], m3.n_out, vec("scale", o3),
                  vec("shift", o3), vec("mean", o3), vec("istd", o3), _ptr(self.bstats, o3, 8),
                  _ptr(self.bstats, tot + o3, 8), 2 * tot, 1, self.zmax)
        bwd(2, bn_dz(m3, o3, self.Z[2], False, pooled=True), m3, m2.n_out, epilogue=0, gout=_ptr(self.G[0]), gout_pitch=m2.n_out,
            **prev_stats(m2, o2, self.Z[1]))
        bwd(1, bn_dz(m2, o2, self.Z[1], False, G=self.G[0]), m2, m1.n_out, epilogue=0, gout=_ptr(self.G[1]), gout_pitch=m1.n_out,
            **prev_stats(m1, o1, self.Z[0]))
        dw(0, bn_dz(m1, o1, self.Z[0], False, G=self.G[1]), m1)
        dx(bn_dz(m1, o1, self.Z[0], True, G=self.G[1]), m1, net.c_pad, epilogue=1, dfeat=_ptr(self.dfeat), feat_c=net.c_pad,
           row_pt=_ptr(r["pt"]), row_grp=_ptr(r["grp"]), act_c=0, grp_per_sample=1)
        plan.call("gad_grad_from_arena", fl.gacc, fl.m2p, fl.n, self.grad, 0)
        return plan


def _stage_net(mod, dev):
    rt = mod.__dict__.get("_gad_rt")
    if rt is None:
        rt = {"net": _StageNet(mod, dev)}
        object.__setattr__(mod, "_gad_rt", rt)
    return rt


def _transpose(src, dst, B, R, C_, src_pitch, src_batch, dst_pitch, dst_batch):
    """dst[b][j][i] = src[b][i][j] (gad_transpose_batched): the (B, C, N) <-> point-major hand-over at the module boundary"""
    import ctypes as C
    src = src if (src.is_contiguous() and src.dtype == torch.float32) else src.contiguous().float()
    hip.call("gad_transpose_batched", src, dst, B, R, C_, src_pitch, C.c_longlong(src_batch), dst_pitch, C.c_longlong(dst_batch))


def _run_forward(mod, net, run, xyz, features):
    B, N = run.B, run.N
    net.flat.sync_packed()            # the parameters may have been stepped by an ordinary torch optimizer
    run.xyz.copy_(xyz)
    _transpose(features, run.feat, B, net.c_feat, N, N, net.c_feat * N, net.c_pad, N * net.c_pad)     # (B,C,N) -> point-major rows
    run.plans[bool(mod.training)].run()
    if mod.training:
        net.batches_tracked += 1
    c_out = net.mats[2].n_out
    out = torch.empty(B, c_out, run.M, dtype=torch.float32, device=xyz.device)
    _transpose(run.F, out, B, run.M, c_out, c_out, run.M * c_out, run.M, c_out * run.M)
    return (None if run.group_all else run.new_xyz.clone()), out


class _SAFunction(torch.autograd.Function):
    """forward / backward of one set-abstraction module over its own activation set (held by the graph node until the
    backward has run, then recycled): several forwards of the same module may be in flight, as in the reference's
    update step (value encoder on the current and on the next state)."""

    @staticmethod
    def forward(ctx, mod, xyz, features, *params):
        rt = _stage_net(mod, xyz.device)
        net = rt["net"]
        B, N, _ = xyz.shape
        pool = net.free.setdefault((B, N), [])
        run = pool.pop() if pool else _StageRun(net, mod, B, N, xyz.device, with_backward=True)
        new_xyz, out = _run_forward(mod, net, run, xyz, features)
        ctx.run, ctx.net, ctx.key = run, net, (B, N)
        ctx.feat_grad = features.requires_grad
        ctx.mark_non_differentiable(*([new_xyz] if new_xyz is not None else []))
        return (new_xyz if new_xyz is not None else xyz.new_zeros(0)), out

    @staticmethod
    def backward(ctx, g_xyz, g_out):
        run, net = ctx.run, ctx.net
        B, N = ctx.key
        c_out = net.mats[2].n_out
        _transpose(g_out, run.dF, B, c_out, run.M, run.M, c_out * run.M, c_out, run.M * c_out)
        run.bwd.run()
        g_feat = None
        if ctx.feat_grad:
            g_feat = torch.empty(B, net.c_feat, N, dtype=torch.float32, device=g_out.device)
            _transpose(run.dfeat, g_feat, B, N, net.c_feat, net.c_pad, N * net.c_pad, N, net.c_feat * N)
        grads = []
        # ONE copy of the call's parameter gradients; VIEWS of it go out.  Autograd may adopt such a view as p.grad (set_to_none /
        # first accumulation): the parameters' .grad tensors of one call then share one storage (disjoint slices: in-place optimizers
        # are unaffected; the flat buffer lives as long as any of them).  One clone per parameter would cost nine launches per call.
        gcopy = run.grad.clone()
        for p, o in zip(net.flat.params, net.flat.offsets[:-1]):
            o = int(o)
            grads.append(gcopy[o:o + p.numel()].view(p.shape) if p.requires_grad else None)
        net.free[ctx.key].append(run)                          # recycled: everything handed out above is a copy
        ctx.run = None
        return (None, None, g_feat) + tuple(grads)


def sa_module_forward(mod, xyz, features):
    hip.require_cuda(xyz, features)
    if features is None:
        raise NotImplementedError("PointnetSAModule without input features is not used by GA-DDPG")
    # xyz may arrive with requires_grad=True for bookkeeping reasons only (the reference slices it out of the tensor that
    # carries the broadcast action channels, core/networks.py:239): it is treated as geometry, its gradient is None
    xyz = xyz.detach()
    B, N, _ = xyz.shape
    dev = xyz.device
    rt = _stage_net(mod, dev)
    net = rt["net"]
    if features.shape[1] != net.c_feat:
        raise RuntimeError("features have %d channels, module expects %d" % (features.shape[1], net.c_feat))
    params = net.flat.params
    needs_grad = torch.is_grad_enabled() and (features.requires_grad or any(p.requires_grad for p in params))
    if needs_grad:
        if not mod.training:
            raise RuntimeError("PointnetSAModule: backward through eval-mode BatchNorm (running statistics) is not "
                               "implemented; call .train() or wrap the call in torch.no_grad()")
        new_xyz, out = _SAFunction.apply(mod, xyz, features, *params)
        return (None if new_xyz.numel() == 0 else new_xyz), out
    key = (B, N)
    if key not in rt:
        rt[key] = _StageRun(net, mod, B, N, dev)
    return _run_forward(mod, net, rt[key], xyz, features)
