"""Synthetic single-launch cases for the split-bf16 form of the layer GEMMs (library option "mfma_split"), one per kernel
family at its bench shape (B = 256: SA1 ~2.1e5 rows, SA2 ~2.7e4 rows, SA3 8192 rows).  Every case runs the SAME C-ABI call
twice -- FP32-MFMA path and split path -- and evaluates both against a float64 reference computed by torch from the same
operands (test infrastructure: the product never calls this).  Used by tests/test_gpu_split_families.py and
tools/ubench_split.py."""
import ctypes as C

import numpy as np
import torch

from ga_ddpg_amd import hip
from ga_ddpg_amd.engine import _dz, _fwd_args, _ptr


def _gen(seed):
    return torch.Generator(device="cuda").manual_seed(seed)


def fma32(z, s, t):
    """relu-less fmaf(z, s, t) as the kernels evaluate it: one rounding (the float64 product and sum are exact enough)"""
    return (z.double() * s.double() + t.double()).float()


class Mirror(object):
    """forward + transposed split-bf16 mirrors of one packed weight matrix (gad_split_weights)"""

    def __init__(self, W, Ks):
        n_out, Kp = W.shape
        self.plane = n_out * Ks
        self.buf = torch.zeros(6 * self.plane, dtype=torch.int16, device=W.device)
        lay = (hip.SplitLayer * 1)()
        lay[0].w_off, lay[0].n_out, lay[0].Kp, lay[0].Ks, lay[0].fwd_off, lay[0].t_off = 0, n_out, Kp, Ks, 0, 3 * self.plane
        hip.call("gad_split_weights", W, lay, 1, self.buf)
        self.Ks, self.n_out = Ks, n_out

    def fwd_kw(self):
        return dict(W_split=_ptr(self.buf, 0, 2), W_split_pitch=self.Ks, W_split_plane=self.plane)

    def t_kw(self):
        return dict(W_split_t=_ptr(self.buf, 3 * self.plane, 2), W_split_t_pitch=self.n_out, W_split_t_plane=self.plane)


def decode_mirror(buf, plane, rows, cols, transposed_sign_axis):
    """hi + mid + lo of a mirror as float64 (rows, cols), signs of the odd reduction blocks undone.  transposed_sign_axis:
    1 = the reduction index is the column (forward mirror), 0 = the row... of the ORIGINAL matrix orientation passed in."""
    u = buf.view(torch.int16).to(torch.int32) & 0xffff
    out = torch.zeros(rows, cols, dtype=torch.float64, device=buf.device)
    for p in range(3):
        bits = (u[p * plane:(p + 1) * plane] << 16).to(torch.int32)
        out += bits.view(torch.float32).double().view(rows, cols)
    idx = torch.arange(cols, device=buf.device)
    sign = torch.where(((idx >> 4) & 1) == 1, -1.0, 1.0).double()
    return out * sign[None, :]


def errors(got, ref):
    """(max |e|, mean |e|, signed mean e, max |ref|) in float64"""
    e = got.double() - ref
    return float(e.abs().max()), float(e.abs().mean()), float(e.mean()), float(ref.abs().max())


def time_call(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters          # us


class Case(object):
    """one launch: run(split) -> dict of output tensors; ref() -> dict of float64 references for the same keys"""
    family = 0
    entry = "gad_gemm_fwd"

    def run_mode(self, split):
        hip.set_option("mfma_split", self.family if split else 0)
        try:
            out = self.run()
            routed = hip.lib().gad_last_kernel().decode()
            torch.cuda.synchronize()
        finally:
            hip.set_option("mfma_split", hip.get_option_default("mfma_split"))
        return out, routed

    def time_mode(self, split, iters=50):
        """microseconds per launch, back-to-back launches of the call alone (outputs accumulate: timing only)"""
        hip.set_option("mfma_split", self.family if split else 0)
        try:
            a = self.args()
            name = self.entry
            f = getattr(hip.lib(), name)
            st = hip.stream()
            return time_call(lambda: hip.check(f(C.byref(a), st), name), iters)
        finally:
            hip.set_option("mfma_split", hip.get_option_default("mfma_split"))


class FwdWide(Case):
    """gad_gemm_fwd on the wide-tile route.  mode: "act" (BatchNorm + ReLU input), "pool" (+ fused max-pool epilogue),
    "gather" (first layer of SA2 / SA3: [feat[pt] | src_xyz[pt] - ctr_xyz[grp]])"""
    family = hip.SPLIT_FWD_WIDE

    def __init__(self, rows, K, N, mode="act", seed=1, a_scale=1.0, w_scale=0.05, ragged=True):
        dev = torch.device("cuda")
        g = _gen(seed)
        self.rows, self.K, self.N, self.mode = rows, K, N, mode
        cap = rows + (1000 if ragged else 0)                     # rows past the live count are never read into results
        self.cap = cap
        self.nrows = torch.tensor([rows], dtype=torch.int32, device=dev)
        self.row_w = torch.randint(1, 4, (cap,), device=dev, generator=g).float()
        self.zout = torch.full((cap, N), float("nan"), device=dev)
        self.stats = torch.zeros(hip.STAT_REPLICAS * 2 * N, dtype=torch.float64, device=dev)
        if mode == "gather":
            Kp = (K + 3 + 7) // 8 * 8
            npts, ngrp = max(rows // 3, 64), max(rows // 8, 8)
            self.feat = torch.randn(npts, K, device=dev, generator=g).abs() * a_scale          # pooled features are >= 0
            self.src = torch.rand(npts, 3, device=dev, generator=g)
            self.ctr = torch.rand(ngrp, 3, device=dev, generator=g)
            self.row_pt = torch.randint(0, npts, (cap,), device=dev, generator=g, dtype=torch.int32)
            self.row_grp = torch.sort(torch.randint(0, ngrp, (cap,), device=dev, generator=g, dtype=torch.int32)).values.contiguous()
            self.W = torch.randn(N, Kp, device=dev, generator=g) * w_scale
            self.W[:, K + 3:] = 0
        else:
            Kp = K
            self.zin = torch.randn(cap, K, device=dev, generator=g) * a_scale
            self.scale = torch.rand(K, device=dev, generator=g) + 0.5
            self.shift = torch.randn(K, device=dev, generator=g) * 0.3 * a_scale
            self.W = torch.randn(N, Kp, device=dev, generator=g) * w_scale
            if mode == "pool":
                self.gsz = 4
                self.row_grp = (torch.arange(cap, device=dev, dtype=torch.int32) // self.gsz).contiguous()
                self.ngrp = (cap + self.gsz - 1) // self.gsz
                self.key = torch.zeros(self.ngrp, N, dtype=torch.int64, device=dev)
                self.gamma = torch.ones(N, device=dev)
        self.Kp = Kp
        self.mirror = Mirror(self.W, K)

    def args(self):
        kw = dict(n_rows_dev=_ptr(self.nrows), n_rows=self.cap, row_w=_ptr(self.row_w), W=_ptr(self.W), Kp=self.Kp, n_out=[self.N],
                  zout=_ptr(self.zout), zout_pitch=self.N, stat_sum=_ptr(self.stats, 0, 8), stat_sq=_ptr(self.stats, self.N, 8),
                  stat_stride=2 * self.N)
        if self.mode == "gather":
            kw.update(mode=1, c_in=self.K + 3, src_xyz=_ptr(self.src), ctr_xyz=_ptr(self.ctr), feat=_ptr(self.feat), feat_c=self.K,
                      act_c=0, grp_per_sample=1, row_pt=_ptr(self.row_pt), row_grp=_ptr(self.row_grp))
        else:
            kw.update(mode=0, zin=_ptr(self.zin), zin_pitch=self.K, c_in=self.K, scale=_ptr(self.scale), shift=_ptr(self.shift), relu=1)
            if self.mode == "pool":
                kw.update(pool_key=_ptr(self.key, 0, 8), pool_row_grp=_ptr(self.row_grp), pool_gamma=_ptr(self.gamma))
        kw.update(self.mirror.fwd_kw())
        return _fwd_args(**kw)

    def run(self):
        self.stats.zero_()
        if self.mode == "pool":
            self.key.zero_()
        a = self.args()
        hip.call_struct("gad_gemm_fwd", a)
        out = {"z": self.zout[:self.rows]}
        out["stat_sum"] = self.stats.view(hip.STAT_REPLICAS, 2, self.N)[:, 0].sum(0)
        out["stat_sq"] = self.stats.view(hip.STAT_REPLICAS, 2, self.N)[:, 1].sum(0)
        if self.mode == "pool":
            out["key"] = self.key.clone()
        return {k: v.clone() for k, v in out.items()}

    def operand(self):
        r = self.rows
        if self.mode == "gather":
            pt, grp = self.row_pt[:r].long(), self.row_grp[:r].long()
            dx = (self.src[pt] - self.ctr[grp])                       # float32 subtraction, as the kernels round it
            return torch.cat([self.feat[pt], dx], 1).double(), self.W[:, :self.K + 3].double()
        return fma32(self.zin[:r], self.scale, self.shift).clamp_min(0).double(), self.W.double()

    def ref(self):
        A, W = self.operand()
        z = A @ W.t()
        w = self.row_w[:self.rows].double()[:, None]
        # "#abs": the sum of |terms| behind a reduced output -- a per-element bias b of z shows up as rows * b in a column sum,
        # so the bias floor of a sum is taken relative to this scale (tests/test_gpu_split_families.py)
        return {"z": z, "stat_sum": (w * z).sum(0), "stat_sq": (w * z * z).sum(0), "stat_sum#abs": (w * z.abs()).sum(0),
                "stat_sq#abs": (w * z * z).sum(0)}

    def flops(self):
        return 2.0 * self.rows * self.K * self.N


class FwdStream(FwdWide):
    """gad_gemm_fwd on the streaming route (SA1 layers 2 / 3: ~2e5 rows, 64 input channels, 64 / 128 outputs; "pool": the fused
    max-pool epilogue of layer 3).  The kernel splits W itself (no mirror involved)."""
    family = hip.SPLIT_FWD_STREAM


class DxWide(Case):
    """gad_gemm_dx on the wide-tile route: dZ = P*g - w*(Q + S*z) from z and a dense (or pooled) gradient, times W; epilogue
    0 stores the ReLU-masked gradient of the previous layer and its BatchNorm-backward sums ("act" / "pool"), epilogue 1
    scatters into the points' feature gradients ("scatter")."""
    family = hip.SPLIT_DX_WIDE
    entry = "gad_gemm_dx"

    def __init__(self, rows, N, K, mode="act", seed=2, g_scale=1.0, w_scale=0.05, slack=1000):
        dev = torch.device("cuda")
        g = _gen(seed)
        self.rows, self.N, self.K, self.mode = rows, N, K, mode
        cap = rows + slack                                       # rows past the live count: never read into results
        self.cap = cap
        self.nrows = torch.tensor([rows], dtype=torch.int32, device=dev)
        self.z = torch.randn(cap, N, device=dev, generator=g)
        # P and the row weights are powers of two: P*g and w*fmaf(S, z, Q) are then exact, so dZ = P*g - w*(Q + S*z) has ONE rounding
        # however the compiler contracts it, and the float64 reference below stages bit-identical dZ values
        self.row_w = torch.pow(2.0, torch.randint(0, 3, (cap,), device=dev, generator=g).float())
        self.vecN = [torch.pow(2.0, torch.randint(-1, 2, (N,), device=dev, generator=g).float()), torch.randn(N, device=dev, generator=g) * 0.1,
                     torch.randn(N, device=dev, generator=g) * 0.1]                          # P, Q, S
        self.sc = torch.ones(N, device=dev)
        Kp = K if mode != "scatter" else (K + 3 + 7) // 8 * 8
        self.Kp = Kp
        self.W = torch.randn(N, Kp, device=dev, generator=g) * w_scale
        if mode == "pool":
            self.gsz = 4
            self.row_grp = (torch.arange(cap, device=dev, dtype=torch.int32) // self.gsz).contiguous()
            ngrp = (cap + self.gsz - 1) // self.gsz
            self.argmax = (torch.arange(ngrp, device=dev, dtype=torch.int32)[:, None] * self.gsz +
                           torch.randint(0, self.gsz, (ngrp, N), device=dev, generator=g, dtype=torch.int32)).contiguous()
            self.dout = torch.randn(ngrp, N, device=dev, generator=g) * g_scale
        else:
            self.G = torch.randn(cap, N, device=dev, generator=g) * g_scale
        if mode == "scatter":
            self.npts = max(rows // 3, 64)
            self.row_pt = torch.randint(0, self.npts, (cap,), device=dev, generator=g, dtype=torch.int32)
            self.row_grp2 = torch.zeros(cap, dtype=torch.int32, device=dev)
            self.dfeat = torch.zeros(self.npts, K, device=dev)
        else:
            self.zprev = torch.randn(cap, K, device=dev, generator=g)
            self.vecK = [torch.rand(K, device=dev, generator=g) + 0.5, torch.randn(K, device=dev, generator=g) * 0.3,
                         torch.randn(K, device=dev, generator=g) * 0.1, torch.rand(K, device=dev, generator=g) + 0.5]   # scale, shift, mean, istd
            self.gout = torch.full((cap, K), float("nan"), device=dev)
            self.bst = torch.zeros(hip.STAT_REPLICAS * 2 * K, dtype=torch.float64, device=dev)
        self.mirror = Mirror(self.W, K)

    def dz_kw(self):
        d = dict(z=_ptr(self.z), z_pitch=self.N, scale=_ptr(self.sc), shift=_ptr(self.sc), relu=1, premasked=1, row_w=_ptr(self.row_w),
                 c=self.N, coefP=_ptr(self.vecN[0]), coefQ=_ptr(self.vecN[1]), coefS=_ptr(self.vecN[2]))
        if self.mode == "pool":
            d.update(gmode=1, argmax=_ptr(self.argmax), dout=_ptr(self.dout), row_grp=_ptr(self.row_grp))
        else:
            d.update(gmode=0, G=_ptr(self.G), g_pitch=self.N)
        return d

    def args(self):
        a = hip.GemmDxArgs()
        a.n_rows_dev, a.n_rows, a.dz, a.n_groups = _ptr(self.nrows), self.cap, _dz(**self.dz_kw()), 1
        a.n_out[0] = self.N
        a.W, a.Kp, a.k_valid, a.grp_per_sample = _ptr(self.W), self.Kp, self.K, 1
        if self.mode == "scatter":
            a.epilogue, a.dfeat, a.feat_c, a.row_pt, a.row_grp, a.act_c = 1, _ptr(self.dfeat), self.K, _ptr(self.row_pt), _ptr(self.row_grp2), 0
        else:
            a.epilogue, a.gout, a.gout_pitch, a.zprev, a.zprev_pitch = 0, _ptr(self.gout), self.K, _ptr(self.zprev), self.K
            a.prev_scale, a.prev_shift, a.prev_mean, a.prev_istd = (_ptr(v) for v in self.vecK)
            a.prev_dbeta, a.prev_dgamma, a.stat_stride, a.store_masked = _ptr(self.bst, 0, 8), _ptr(self.bst, self.K, 8), 2 * self.K, 1
        for k, v in self.mirror.t_kw().items():
            setattr(a, k, v)
        return a

    def run(self):
        a = self.args()
        if self.mode == "scatter":
            self.dfeat.zero_()
            hip.call_struct("gad_gemm_dx", a)
            return {"dfeat": self.dfeat.clone()}
        self.bst.zero_()
        hip.call_struct("gad_gemm_dx", a)
        K = self.K
        return {"gout": self.gout[:self.rows].clone(), "dbeta": self.bst.view(hip.STAT_REPLICAS, 2, K)[:, 0].sum(0).clone(),
                "dgamma": self.bst.view(hip.STAT_REPLICAS, 2, K)[:, 1].sum(0).clone()}

    def dz32(self):
        """the A operand exactly as the kernels form it (float32 P*g - w*fmaf(S, z, Q) with exact products: one rounding), in
        float64 BEFORE that rounding"""
        r = self.rows
        z = self.z[:r]
        if self.mode == "pool":
            grp = self.row_grp[:r].long()
            rowid = torch.arange(r, device=z.device, dtype=torch.int32)[:, None]
            g = torch.where(self.argmax[grp] == rowid, self.dout[grp], torch.zeros((), device=z.device))
        else:
            g = self.G[:r]
        P, Q, S = self.vecN
        inner = fma32(z, S, Q)
        return (P.double() * g.double() - self.row_w[:r].double()[:, None] * inner.double())

    def ref(self):
        dz = self.dz32()
        r = self.rows
        gx = dz.float().double() @ self.W[:, :self.K].double()
        if self.mode == "scatter":
            out = torch.zeros(self.npts, self.K, dtype=torch.float64, device=gx.device)
            out.index_add_(0, self.row_pt[:r].long(), gx)
            ab = torch.zeros_like(out)
            ab.index_add_(0, self.row_pt[:r].long(), gx.abs())
            return {"dfeat": out, "dfeat#abs": ab}
        ps, pt, pm, pi = self.vecK
        mask = fma32(self.zprev[:r], ps, pt) > 0
        ga = torch.where(mask, gx, torch.zeros((), dtype=torch.float64, device=gx.device))
        xhat = (self.zprev[:r].double() - pm.double()) * pi.double()
        return {"gout": ga, "dbeta": ga.sum(0), "dgamma": (ga * xhat).sum(0), "dbeta#abs": ga.abs().sum(0), "dgamma#abs": (ga * xhat).abs().sum(0)}

    def flops(self):
        return 2.0 * self.rows * self.K * self.N


class DwWide(Case):
    """gad_gemm_dw on the wide-tile route: dW[n][k] = sum_r dZ[r][n] * X[r][k] into the f64 arena; X = relu(bn(z_prev)) ("act",
    "pool": pooled gradient source) or the gathered rows [feat[pt] | src_xyz[pt] - ctr_xyz[grp]] of an SA2 / SA3 first layer
    ("gather")"""
    family = hip.SPLIT_DW_WIDE
    entry = "gad_gemm_dw"

    def __init__(self, rows, N, K, mode="act", seed=3):
        dev = torch.device("cuda")
        self.dx = DxWide(rows, N, K, mode="pool" if mode == "pool" else "act", seed=seed)
        self.rows, self.N, self.K, self.mode = rows, N, K, mode
        self.Kp = K if mode != "gather" else (K + 3 + 7) // 8 * 8
        if mode == "gather":
            g = _gen(seed + 100)
            cap = self.dx.cap
            npts, ngrp = max(rows // 3, 64), max(rows // 8, 8)
            self.feat = torch.randn(npts, K, device=dev, generator=g).abs()
            self.src = torch.rand(npts, 3, device=dev, generator=g)
            self.ctr = torch.rand(ngrp, 3, device=dev, generator=g)
            self.row_pt = torch.randint(0, npts, (cap,), device=dev, generator=g, dtype=torch.int32)
            self.row_grp = torch.sort(torch.randint(0, ngrp, (cap,), device=dev, generator=g, dtype=torch.int32)).values.contiguous()
        self.gacc = torch.zeros(N * self.Kp, dtype=torch.float64, device=dev)
        self.ws = torch.empty(48 * 1024 * 1024, device=dev)

    def args(self):
        d = self.dx
        a = hip.GemmDwArgs()
        kw = dict(n_rows_dev=_ptr(d.nrows), n_rows=d.cap, row_w=_ptr(d.row_w), Kp=self.Kp, n_out=[self.N], w_off=[0])
        if self.mode == "gather":
            kw.update(mode=1, c_in=self.K + 3, src_xyz=_ptr(self.src), ctr_xyz=_ptr(self.ctr), feat=_ptr(self.feat), feat_c=self.K,
                      act_c=0, grp_per_sample=1, row_pt=_ptr(self.row_pt), row_grp=_ptr(self.row_grp))
        else:
            kw.update(mode=0, zin=_ptr(d.zprev), zin_pitch=self.K, c_in=self.K, scale=_ptr(d.vecK[0]), shift=_ptr(d.vecK[1]), relu=1)
        a.inp = _fwd_args(**kw)
        a.dz = _dz(**d.dz_kw())
        a.gacc, a.partial, a.partial_elems = _ptr(self.gacc), _ptr(self.ws), self.ws.numel()
        return a

    def run(self):
        self.gacc.zero_()
        a = self.args()
        hip.call_struct("gad_gemm_dw", a)
        return {"dW": self.gacc.view(self.N, self.Kp)[:, :self.K + (3 if self.mode == "gather" else 0)].clone()}

    def ref(self):
        d = self.dx
        r = self.rows
        dz = d.dz32().float().double()
        if self.mode == "gather":
            pt, grp = self.row_pt[:r].long(), self.row_grp[:r].long()
            x = torch.cat([self.feat[pt], self.src[pt] - self.ctr[grp]], 1).double()
        else:
            x = fma32(d.zprev[:r], d.vecK[0], d.vecK[1]).clamp_min(0).double()
        return {"dW": dz.t() @ x, "dW#abs": dz.abs().t() @ x.abs()}

    def flops(self):
        return 2.0 * self.rows * self.K * self.N


class BwdStream(Case):
    """SA1's streaming backward: gad_gemm_bwd (dX + dW of one layer in one pass: fused=True) or gad_gemm_dx alone (fused=False) on
    ~2e5 rows, 64 input channels, N = 64 / 128 output channels; dense gradient ("act") or pooled source ("pool": layer 3)"""
    family = hip.SPLIT_BWD_STREAM

    def __init__(self, rows, N, mode="act", fused=True, seed=5, slack=1000):
        dev = torch.device("cuda")
        self.dx = DxWide(rows, N, 64, mode=mode, seed=seed, slack=slack)
        d = self.dx
        if mode == "pool":                                   # SA1-like groups: ~26 consecutive rows each
            g = _gen(seed + 7)
            ngrp = max(rows // 26, 4)
            cuts = torch.sort(torch.randint(1, d.cap, (ngrp - 1,), device=dev, generator=g)).values
            grp = torch.zeros(d.cap, dtype=torch.int32, device=dev)
            grp[cuts.long()] = 1
            d.row_grp = torch.cumsum(grp, 0).to(torch.int32).contiguous()
            ng = int(d.row_grp.max().item()) + 1
            first = torch.zeros(ng, dtype=torch.int64, device=dev)
            cnt = torch.bincount(d.row_grp.long(), minlength=ng)
            first[1:] = torch.cumsum(cnt, 0)[:-1]
            pick = (torch.rand(ng, N, device=dev, generator=g) * cnt[:, None].double()).long()
            pick = torch.minimum(pick, (cnt - 1).clamp_min(0)[:, None])
            d.argmax = (first[:, None] + pick).to(torch.int32).contiguous()
            d.dout = torch.randn(ng, N, device=dev, generator=g)
        self.rows, self.N, self.mode, self.fused = rows, N, mode, fused
        self.entry = "gad_gemm_bwd" if fused else "gad_gemm_dx"
        self.gacc = torch.zeros(N * 64, dtype=torch.float64, device=dev)
        self.ws = torch.empty(2 * 256 * 128 * 64, device=dev)

    def dw_args(self):
        d = self.dx
        a = hip.GemmDwArgs()
        a.inp = _fwd_args(mode=0, zin=_ptr(d.zprev), zin_pitch=64, c_in=64, scale=_ptr(d.vecK[0]), shift=_ptr(d.vecK[1]), relu=1,
                          n_rows_dev=_ptr(d.nrows), n_rows=d.cap, row_w=_ptr(d.row_w), Kp=64, n_out=[self.N], w_off=[0])
        a.dz = _dz(**d.dz_kw())
        a.gacc, a.partial, a.partial_elems = _ptr(self.gacc), _ptr(self.ws), self.ws.numel()
        return a

    def run(self):
        d = self.dx
        d.bst.zero_()
        self.gacc.zero_()
        ax = d.args()
        if self.fused:
            aw = self.dw_args()
            hip.check(hip.lib().gad_gemm_bwd(C.byref(ax), C.byref(aw), hip.stream()), "gad_gemm_bwd")
        else:
            hip.call_struct("gad_gemm_dx", ax)
        K = 64
        out = {"gout": d.gout[:self.rows].clone(), "dbeta": d.bst.view(hip.STAT_REPLICAS, 2, K)[:, 0].sum(0).clone(),
               "dgamma": d.bst.view(hip.STAT_REPLICAS, 2, K)[:, 1].sum(0).clone()}
        if self.fused:
            out["dW"] = self.gacc.view(self.N, 64).clone()
        return out

    def time_mode(self, split, iters=50):
        hip.set_option("mfma_split", self.family if split else 0)
        try:
            ax = self.dx.args()
            st = hip.stream()
            if self.fused:
                aw = self.dw_args()
                f = hip.lib().gad_gemm_bwd
                return time_call(lambda: hip.check(f(C.byref(ax), C.byref(aw), st), "gad_gemm_bwd"), iters)
            f = hip.lib().gad_gemm_dx
            return time_call(lambda: hip.check(f(C.byref(ax), st), "gad_gemm_dx"), iters)
        finally:
            hip.set_option("mfma_split", hip.get_option_default("mfma_split"))

    def ref(self):
        d = self.dx
        out = d.ref()
        if self.fused:
            dz = d.dz32().float().double()
            x = fma32(d.zprev[:self.rows], d.vecK[0], d.vecK[1]).clamp_min(0).double()
            out["dW"] = dz.t() @ x
            out["dW#abs"] = dz.abs().t() @ x
        return out

    def flops(self):
        return (4.0 if self.fused else 2.0) * self.rows * 64 * self.N
