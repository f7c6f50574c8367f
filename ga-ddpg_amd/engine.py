"""Host-side engine of the fused update-step path: flat parameter storage in the packed compute
layout, per-pass activation workspaces, and the launch plans (sequences of C-ABI calls) for the
PointNet++ encoder, the actor/critic heads and their backward passes.

Nothing here computes: every arithmetic step is a libgaddpg kernel (ga-ddpg_amd/hip.py).  torch
only owns device memory and the stream.  A *plan* is a list of pre-built calls over static buffers.

Reference structure mirrored: core/networks.py:65-92 (base_network = SA1, SA2, SA3, FC head),
:253-300 (QNetwork), :303-371 (GaussianPolicy); upstream PointnetSAModule.forward.
"""
import numpy as np
import torch

from . import hip

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def round8(k):
    return (k + 7) // 8 * 8


# ----------------------------------------------------------------------------------------------
# flat parameter storage
# ----------------------------------------------------------------------------------------------
class MatSpec(object):
    """One Conv2d-1x1 / Linear weight (n_out, k_in) [+ bias] and an optional BatchNorm after it."""

    def __init__(self, weight, bias=None, bn=None, gather_feat_c=None):
        """gather_feat_c: for an SA stage's first conv (input [xyz(3), feat, action]) the number of
        feature channels; its packed columns are permuted to [feat, xyz, action] (16-B aligned gather).
        It may exceed the real feature count (stand-alone modules pad the point features to a multiple of 4
        with zero columns): the padding columns of the packed weight stay zero."""
        self.weight, self.bias, self.bn = weight, bias, bn
        self.gather_feat_c = gather_feat_c
        self.n_out = weight.shape[0]
        self.k_in = int(np.prod(weight.shape[1:]))
        self.ones_col = self.k_in if bias is not None else -1
        width = self.k_in
        if gather_feat_c is not None and gather_feat_c + 3 > width:
            width = gather_feat_c + 3                       # padded feature block: real inputs are [xyz, feat (< gather_feat_c)]
        self.k_packed = width
        self.Kp = round8(width + (1 if bias is not None else 0))
        self.w_off = self.g_off = self.b_off = -1          # packed offsets (filled by FlatNet)
        self.bn_index = -1                                 # index into the per-pass BN vectors


HOST_RING = 4             # steps the host may run ahead of the GPU (pinned staging blocks per FlatNet / FusedRuntime)


class FlatNet(object):
    """Flat float32 master buffer holding every parameter of a network (the nn.Parameters become
    views into it), a packed/padded compute copy for the GEMMs, flat .grad / Adam state, an f64
    gradient-accumulation arena in the packed layout, and the master->packed index map."""

    def __init__(self, named_params, mats, device, never_trained=()):
        """named_params: ordered (name, nn.Parameter); mats: MatSpecs in PACKED order."""
        self.device = device
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        sizes = [p.numel() for p in self.params]
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.n = int(self.offsets[-1])
        self.mats = list(mats)
        off_of = {id(p): int(o) for p, o in zip(self.params, self.offsets[:-1])}

        m2p = np.full(self.n, -1, dtype=np.int32)
        pk = 0
        for m in self.mats:
            m.w_off = pk
            rows = np.arange(m.n_out, dtype=np.int64)[:, None]
            cols = np.arange(m.k_in, dtype=np.int64)
            if m.gather_feat_c is not None:
                fc = m.gather_feat_c
                cols = np.where(cols < 3, cols + fc, np.where(cols < 3 + fc, cols - 3, cols))
                assert m.bias is None or m.k_packed == m.k_in
            cols = cols[None, :]
            mo = off_of[id(m.weight)]
            m2p[mo:mo + m.n_out * m.k_in] = (pk + rows * m.Kp + cols).reshape(-1)
            if m.bias is not None:
                mo = off_of[id(m.bias)]
                m2p[mo:mo + m.n_out] = pk + np.arange(m.n_out) * m.Kp + m.k_in
            pk += m.n_out * m.Kp
        for m in self.mats:
            if m.bn is not None:
                c = m.n_out
                m.g_off, m.b_off = pk, pk + c
                mo = off_of[id(m.bn.weight)]
                m2p[mo:mo + c] = pk + np.arange(c)
                mo = off_of[id(m.bn.bias)]
                m2p[mo:mo + c] = pk + c + np.arange(c)
                pk += 2 * c
        assert (m2p >= 0).all(), "every parameter must have a packed slot"
        self.packed_n = pk

        f32 = dict(dtype=torch.float32, device=device)
        self.master = torch.zeros(self.n, **f32)
        self.grad = torch.zeros(self.n, **f32)
        self.exp_avg = torch.zeros(self.n, **f32)
        self.exp_avg_sq = torch.zeros(self.n, **f32)
        self.packed = torch.zeros(self.packed_n, **f32)
        self.gacc = torch.zeros(self.packed_n, dtype=torch.float64, device=device)
        self.m2p = torch.from_numpy(m2p).to(device)
        active = np.ones(self.n, dtype=np.uint8)
        for name, p in named_params:
            if any(name.startswith(s) for s in never_trained):
                o = off_of[id(p)]
                active[o:o + p.numel()] = 0
        self.active = torch.from_numpy(active).to(device)
        self.hyper = torch.zeros(8, **f32)                 # {lr,b1,b2,eps,wd,bc1,sqrt(bc2),grad_scale}
        # pinned host blocks for the scalars, one per in-flight step (runtime.HOST_RING): the host may enqueue step
        # N+1 while the copy node of step N has not run yet
        self.hyper_ring = torch.zeros(HOST_RING, 8, dtype=torch.float32)
        if device.type == "cuda":
            self.hyper_ring = self.hyper_ring.pin_memory()
        self.hyper_host = self.hyper_ring[0]
        self.step_count = 0
        # re-home the parameters into the flat buffers
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets[:-1]):
                o = int(o)
                self.master[o:o + p.numel()].copy_(p.detach().reshape(-1).to(device))
                p.data = self.master[o:o + p.numel()].view(p.shape)
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self.sync_packed()

    def rebind_grad(self, view):
        """move the flat gradient (and the parameters' .grad views) into `view` (n floats of a shared bucket)"""
        view.copy_(self.grad)
        self.grad = view
        for p, o in zip(self.params, self.offsets[:-1]):
            o = int(o)
            p.grad = view[o:o + p.numel()].view(p.shape)

    def segment(self, prefix):
        """(lo, hi) master range of the parameters whose name starts with prefix (must be contiguous)."""
        idx = [i for i, n in enumerate(self.names) if n.startswith(prefix)]
        assert idx and idx == list(range(idx[0], idx[-1] + 1))
        return int(self.offsets[idx[0]]), int(self.offsets[idx[-1] + 1])

    def sync_packed(self):
        """master -> packed (after load_state_dict / external edits)."""
        hip.call("gad_pack_params", self.master, self.m2p, self.n, self.packed)
        self.refresh_split()

    # ---- split-bf16 weight mirrors (include/gaddpg.h gad_split_weights; library option "mfma_split") ----
    split = None

    def enable_split(self, layers):
        """layers: [(MatSpec, Ks)] -- the matrices whose launches may take the split-bf16 form and the number of leading packed
        columns they multiply on the MFMA (Kp of an ACT layer, feat_c of a gathered first layer); Ks % 32 == 0.  Allocates the
        forward + transposed mirrors (3 bf16 planes each) and fills them; refresh_split() must follow every change of `packed`."""
        layers = [(m, ks) for m, ks in layers if ks % 32 == 0 and 32 <= ks <= m.Kp and m.w_off % 4 == 0 and m.n_out % 4 == 0]
        if not layers:
            return
        assert len(layers) <= hip.MAX_SPLIT_LAYERS
        arr = (hip.SplitLayer * len(layers))()
        off = 0
        for y, (m, ks) in zip(arr, layers):
            plane = m.n_out * ks
            y.w_off, y.n_out, y.Kp, y.Ks, y.fwd_off, y.t_off = m.w_off, m.n_out, m.Kp, ks, off, off + 3 * plane
            m.split = dict(fwd=off, t=off + 3 * plane, Ks=ks, plane=plane)
            off += 6 * plane
        self.split = torch.zeros(off, dtype=torch.int16, device=self.device)
        self._split_layers = arr
        self.refresh_split()

    def refresh_split(self):
        if self.split is not None:
            hip.call("gad_split_weights", self.packed, self._split_layers, len(self._split_layers), self.split)

    def split_fwd_kw(self, m):
        """gad_gemm_fwd_args fields of layer m's forward mirror ({} if it has none)"""
        sp = getattr(m, "split", None)
        if self.split is None or sp is None:
            return {}
        return dict(W_split=hip.Ptr(self.split.data_ptr() + 2 * sp["fwd"]), W_split_pitch=sp["Ks"], W_split_plane=sp["plane"])

    def split_t_kw(self, m):
        """gad_gemm_dx_args fields of layer m's transposed mirror"""
        sp = getattr(m, "split", None)
        if self.split is None or sp is None:
            return {}
        return dict(W_split_t=hip.Ptr(self.split.data_ptr() + 2 * sp["t"]), W_split_t_pitch=m.n_out, W_split_t_plane=sp["plane"])

    def p_w(self, m):
        return hip.Ptr(self.packed.data_ptr() + 4 * m.w_off)

    def p_gamma(self, m):
        return hip.Ptr(self.packed.data_ptr() + 4 * m.g_off)

    def p_beta(self, m):
        return hip.Ptr(self.packed.data_ptr() + 4 * m.b_off)

    def set_adam_hyper(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0, upload=True):
        """this step's Adam scalars -> the pinned host block (and, unless upload=False, on to the device block: a step
        replayed from a HIP graph carries that copy as a node that reads the pinned block at replay time)"""
        self.step_count += 1
        t = self.step_count
        h = self.hyper_host.numpy()
        h[0], h[1], h[2], h[3], h[4] = lr, betas[0], betas[1], eps, weight_decay
        h[5] = 1.0 - betas[0] ** t
        h[6] = float(np.sqrt(1.0 - betas[1] ** t))
        h[7] = grad_scale
        if upload:
            self.hyper.copy_(self.hyper_host, non_blocking=True)


# ----------------------------------------------------------------------------------------------
# geometry workspace (depends on the xyz of a cloud set only; shared by every pass over it)
# ----------------------------------------------------------------------------------------------
class SAConfig(object):
    def __init__(self, npoint, radius, nsample):
        self.npoint, self.radius, self.nsample = npoint, radius, nsample


class Geometry(object):
    """FPS / ball-query / de-duplicated rows of SA1 and SA2 plus the static GroupAll rows of SA3."""

    def __init__(self, B, N, sa1, sa2, device):
        self.B, self.N, self.sa1, self.sa2 = B, N, sa1, sa2
        i32 = dict(dtype=torch.int32, device=device)
        f32 = dict(dtype=torch.float32, device=device)
        M1, M2 = sa1.npoint, sa2.npoint
        self.M1, self.M2 = M1, M2
        self.xyz = torch.empty(B, N, 3, **f32)
        self.feat0 = torch.empty(B * N, 4, **f32)
        self.fps1 = torch.empty(B, M1, **i32)
        self.new_xyz1 = torch.empty(B, M1, 3, **f32)
        self.idx1 = torch.empty(B, M1, sa1.nsample, **i32)
        self.cnt1 = torch.empty(B, M1, **i32)
        self.fps2 = torch.empty(B, M2, **i32)
        self.new_xyz2 = torch.empty(B, M2, 3, **f32)
        self.idx2 = torch.empty(B, M2, sa2.nsample, **i32)
        self.cnt2 = torch.empty(B, M2, **i32)
        self.rows = []
        self.rows_n = torch.zeros(4, **i32)        # live rows of SA1, SA2, SA3 side by side: one 8-byte D2H copy brings SA1 + SA2 to the host
        for s_, (G, cap) in enumerate(((B * M1, B * M1 * sa1.nsample), (B * M2, B * M2 * sa2.nsample), (B, B * M2))):
            self.rows.append(dict(G=G, cap=cap, off=torch.zeros(G + 1, **i32), pt=torch.zeros(cap, **i32),
                                  grp=torch.zeros(cap, **i32), w=torch.zeros(cap, **f32),
                                  n=self.rows_n[s_:s_ + 1]))
        r3 = self.rows[2]
        hip.call("gad_rows_group_all", B, M2, r3["off"], r3["pt"], r3["grp"], r3["w"], r3["n"])
        self.counts = (float(B * M1 * sa1.nsample), float(B * M2 * sa2.nsample), float(B * M2))
        # expected live rows of SA1 / SA2 (host ints, 0 = unknown): grid-size hints for the tile launches over de-duplicated
        # rows (gad_grid_rows_hint); the owner of the geometry keeps them current from the row counts of earlier minibatches
        self.rows_hint = np.zeros(2, dtype=np.int32)

    def _calls(self, point_state):
        B, N = self.B, self.N
        NP = point_state.shape[2]
        skip = NP - N
        r0, r1 = self.rows[0], self.rows[1]
        return [("gad_prep_points", point_state, B, point_state.shape[1], NP, skip, self.xyz, self.feat0),
                ("gad_furthest_point_sampling", self.xyz, B, N, self.M1, self.fps1, self.new_xyz1),
                ("gad_ball_query", self.new_xyz1, self.xyz, B, N, self.M1, float(self.sa1.radius), self.sa1.nsample, self.idx1,
                 self.cnt1),
                ("gad_rows_from_ball_query", self.idx1, self.cnt1, B * self.M1, self.M1, N, self.sa1.nsample, r0["off"], r0["pt"],
                 r0["grp"], r0["w"], r0["n"]),
                ("gad_furthest_point_sampling", self.new_xyz1, B, self.M1, self.M2, self.fps2, self.new_xyz2),
                ("gad_ball_query", self.new_xyz2, self.new_xyz1, B, self.M1, self.M2, float(self.sa2.radius), self.sa2.nsample,
                 self.idx2, self.cnt2),
                ("gad_rows_from_ball_query", self.idx2, self.cnt2, B * self.M2, self.M2, self.M1, self.sa2.nsample, r1["off"],
                 r1["pt"], r1["grp"], r1["w"], r1["n"])]

    def run(self, point_state):
        """point_state (B,4,NP) f32 device tensor in the replay layout (gripper points first)."""
        for c in self._calls(point_state):
            hip.call(*c)

    def plan(self, point_state):
        """the same seven launches as a Plan over a static input buffer (embedded into the step's replay list)"""
        p = Plan()
        for c in self._calls(point_state):
            p.call(*c)
        p.keep.append(point_state)
        return p


# ----------------------------------------------------------------------------------------------
# encoder description and per-pass activation slot
# ----------------------------------------------------------------------------------------------
class EncoderNet(object):
    """Packed view of one `base_network`: 3 SA stages of 3 (conv, BN) pairs + 2 (Linear, BN1d)."""

    def __init__(self, module, device, prefix=""):
        """module = nn.ModuleList([ModuleList(SA1,SA2,SA3), Sequential(fc)]) as core.networks builds."""
        sa_mods, fc = module[0], module[1]
        self.sa_mats = []
        feat_cs = (4, sa_mods[0].mlps[0][6].weight.shape[0], sa_mods[1].mlps[0][6].weight.shape[0])
        for sa, nfeat in zip(sa_mods, feat_cs):
            seq = sa.mlps[0]
            self.sa_mats.append([MatSpec(seq[0].weight, None, seq[1], gather_feat_c=nfeat),
                                 MatSpec(seq[3].weight, None, seq[4]), MatSpec(seq[6].weight, None, seq[7])])
        self.fc_mats = [MatSpec(fc[0].weight, fc[0].bias, fc[1]), MatSpec(fc[3].weight, fc[3].bias, fc[4])]
        self.mats = [m for st in self.sa_mats for m in st] + self.fc_mats
        for i, m in enumerate(self.mats):
            m.bn_index = i
        self.flat = FlatNet(list(module.named_parameters()), self.mats, device)
        # split-bf16 mirrors: every SA layer behind a BatchNorm + ReLU (K = the channel count) and the gathered first layers of
        # SA2 / SA3 (their feature columns; the three coordinate columns stay a rank-3 f32 update)
        self.flat.enable_split([(m, (m.gather_feat_c if l == 0 else m.Kp)) for st in self.sa_mats for l, m in enumerate(st)
                                if (l > 0 or m.gather_feat_c >= 32)])
        self.c_feat = self.sa_mats[0][0].k_in - 3              # 4 (policy) or 10 (critic: + 6 action channels)
        self.act_c = self.c_feat - 4
        # BatchNorm running statistics: one flat buffer per kind, module buffers become views
        tot = sum(m.n_out for m in self.mats)
        self.running_mean = torch.zeros(tot, dtype=torch.float32, device=device)
        self.running_var = torch.ones(tot, dtype=torch.float32, device=device)
        self.bn_off = []
        self.batches_tracked = torch.zeros(len(self.mats), dtype=torch.int64, device=device)
        o = 0
        for i_m, m in enumerate(self.mats):
            self.bn_off.append(o)
            with torch.no_grad():
                self.running_mean[o:o + m.n_out].copy_(m.bn.running_mean.to(device))
                self.running_var[o:o + m.n_out].copy_(m.bn.running_var.to(device))
            m.bn.running_mean = self.running_mean[o:o + m.n_out]
            m.bn.running_var = self.running_var[o:o + m.n_out]
            self.batches_tracked[i_m] = int(m.bn.num_batches_tracked)
            m.bn.num_batches_tracked = self.batches_tracked[i_m]      # 0-d view of the flat counter buffer
            o += m.n_out
        self.bn_total = tot

    def bump_batches_tracked(self, k=1):
        self.batches_tracked += k           # one launch for all BatchNorm counters of the encoder


class EncoderSlot(object):
    """Activations of one encoder pass (raw pre-BN layer outputs, pooled features, arg-max, the
    per-layer BN vectors) + the scratch the backward pass needs.  Sized for the worst case
    (every neighbourhood full)."""

    def __init__(self, geo, enc, device, with_backward=True):
        """with_backward=False (a pass that is never back-propagated: the TD-target passes): the pooled layers' raw
        outputs are not stored at all -- their max-pool is folded into the GEMM epilogue and nothing else reads them"""
        f32 = dict(dtype=torch.float32, device=device)
        B = geo.B
        self.geo, self.B = geo, B
        self.with_backward = with_backward
        caps = [geo.rows[0]["cap"], geo.rows[1]["cap"], geo.rows[2]["cap"]]
        self.Z = []
        for s, st in enumerate(enc.sa_mats):
            self.Z.append([torch.empty(caps[s], m.n_out, **f32) if (with_backward or l < 2) else None
                           for l, m in enumerate(st)])
        self.F = [torch.empty(geo.rows[s]["G"], enc.sa_mats[s][2].n_out, **f32) for s in range(3)]
        self.argmax = [torch.empty(geo.rows[s]["G"], enc.sa_mats[s][2].n_out, dtype=torch.int32, device=device)
                       for s in range(3)]
        # fused max-pool: packed (value, ~row) keys per (group, channel), 0 between passes (gad_pool_finalize resets them)
        self.key = [torch.zeros(geo.rows[s]["G"], enc.sa_mats[s][2].n_out, dtype=torch.int64, device=device) for s in range(3)]
        # raw value of every arg-max row (the pooled layer's BatchNorm-backward sums read it instead of gathering z[argmax])
        self.zmax = [torch.empty(geo.rows[s]["G"], enc.sa_mats[s][2].n_out, **f32) if with_backward else None
                     for s in range(3)]
        self.Zfc = [torch.empty(B, m.n_out, **f32) for m in enc.fc_mats]
        tot = enc.bn_total
        R = hip.STAT_REPLICAS
        self.stats = torch.zeros(R * 2 * tot, dtype=torch.float64, device=device)  # [replica][sum | sq][channel]
        self.scale = torch.empty(tot, **f32)
        self.shift = torch.empty(tot, **f32)
        self.mean = torch.empty(tot, **f32)
        self.istd = torch.empty(tot, **f32)
        if with_backward:
            self.bstats = torch.zeros(R * 2 * tot, dtype=torch.float64, device=device)  # [replica][dbeta | dgamma]
            self.coef = torch.empty(3 * tot, **f32)                                  # P | Q | S
            # dY scratch: one pair per SA stage + one for the FC head, so a weight-gradient GEMM running on the side
            # stream never reads a buffer that the dX chain of the NEXT stage is already overwriting
            self.G = [[torch.empty(caps[s] * enc.sa_mats[s][l].n_out, **f32) for l in (1, 0)] for s in range(3)]
            self.Gfc = torch.empty(B * 1024, **f32)
            self.dF = [torch.zeros(geo.rows[s]["G"], enc.sa_mats[s][2].n_out, **f32) for s in range(3)]
            self.daction = torch.zeros(B, 6, dtype=torch.float64, device=device)
        self.tot = tot


def slot_view(slot, geo):
    """the same activation buffers bound to another Geometry of the same shape (the runtime alternates between two input /
    geometry sets; the multi-GB activation scratch exists once)"""
    import copy
    assert [r["cap"] for r in geo.rows] == [r["cap"] for r in slot.geo.rows] and geo.B == slot.B
    v = copy.copy(slot)
    v.geo = geo
    return v


def _ptr(t, off_elems=0, size=4):
    """raw device address of element `off_elems` of tensor t (element size in bytes)"""
    return None if t is None else hip.Ptr(t.data_ptr() + size * off_elems)


def _fwd_args(**kw):
    a = hip.GemmFwdArgs()
    a.n_groups = 1
    a.ones_col = -1
    a.grp_per_sample = 1
    for k, v in kw.items():
        if k in ("zin_off", "w_off", "out_off", "n_out"):
            arr = getattr(a, k)
            for i, x in enumerate(v):
                arr[i] = int(x)
        else:
            setattr(a, k, v)
    return a


def _dz(**kw):
    d = hip.DzSrc()
    for k, v in kw.items():
        setattr(d, k, v)
    return d


# bench.py / diagnostics: durations of tagged launches, stamped by the kernels themselves (gad_timing_slot: first
# wavefront start -> last wavefront end on the device wall clock -- what a profiler reports as the dispatch duration;
# HIP events around a 25 us launch inside a five-stream step read 8 - 20 us high: event packets, queue waits)
TIMING = {"enabled": False, "tag": None, "slots": None, "next": 0, "tags": [], "routed": {}}
_TIMED_CALLS = ("gad_gemm_fwd", "gad_gemm_dx", "gad_gemm_dw", "gad_gemm_bwd", "gad_segment_pool")      # (entry points that take a timing slot)


TIMING_WAVES = 16384          # include/gaddpg.h GAD_TIMING_WAVES


def timing_start(tag="*", capacity=2048):
    """time every launch whose plan tag is `tag` ("*": all, or a set of tags) until timing_stop(); capacity = launches
    (256 KB of stamps each)"""
    dev = torch.device("cuda", torch.cuda.current_device())
    slots = torch.zeros(capacity, TIMING_WAVES, 2, dtype=torch.int64, device=dev)
    slots[:, :, 0] = torch.iinfo(torch.int64).max
    TIMING.update(enabled=True, tag=tag, slots=slots, next=0, tags=[], routed={})


def timing_stop(spans=False):
    """-> {tag: [milliseconds per launch, ...]} of the launches timed since timing_start(); spans=True: a list of
    (tag, start_ms, end_ms) in launch order instead (one clock for every stream: a timeline of the step)"""
    TIMING["enabled"] = False
    torch.cuda.synchronize()
    khz = hip.lib().gad_wall_clock_khz()
    n, slots = TIMING["next"], TIMING["slots"]
    out = {}
    if slots is not None and n:
        t0 = slots[:n, :, 0].min(dim=1).values.cpu().numpy()
        t1 = slots[:n, :, 1].max(dim=1).values.cpu().numpy()
        if spans:
            out = [(tag, float(a) / float(khz), float(b) / float(khz)) for tag, a, b in zip(TIMING["tags"], t0, t1) if b > 0 and a < b]
        for tag, a, b in (() if spans else zip(TIMING["tags"], t0, t1)):
            if b > 0 and a < b:
                out.setdefault(tag, []).append(float(b - a) / float(khz))          # ticks / kHz = ms
    TIMING.update(slots=None, next=0, tags=[])
    return out


def coalesce_grads(flats):
    """Put the gradients of the networks that one optimiser phase updates into ONE buffer (each network's slice padded to
    64 floats), so that a data-parallel run needs a single all-reduce per phase.  Must run before any plan that bakes
    the .grad pointers is built; idempotent for the same group of networks (cached FlatNets are shared by runtimes)."""
    key = tuple(id(f) for f in flats)
    have = getattr(flats[0], "_grad_bucket", None)
    if have is not None:
        assert have[0] == key, "a FlatNet can be part of one gradient bucket only"
        return have[1]
    assert all(getattr(f, "_grad_bucket", None) is None for f in flats)
    pad = lambda n: (n + 63) // 64 * 64
    buf = torch.zeros(sum(pad(f.n) for f in flats), dtype=torch.float32, device=flats[0].device)
    off = 0
    for f in flats:
        f.rebind_grad(buf[off:off + f.n])
        off += pad(f.n)
    for f in flats:
        f._grad_bucket = (key, buf)
    return buf


_DW_WS = {}


def dw_workspace(device, elems=None, lane=0):
    """scratch for the weight-gradient split-K partial tiles: one per device and per stream lane (launches of a
    lane are stream-ordered; lane 0 = the main stream, 1.. = the dW side streams)"""
    key = (str(device), lane)
    if key not in _DW_WS:
        if elems is None:                    # head lanes (>= 100): 16 splits of the widest head matrix (768 x 520) at most
            elems = 48 * 1024 * 1024 if lane < 100 else 8 * 1024 * 1024
        _DW_WS[key] = torch.empty(elems, dtype=torch.float32, device=device)
    return _DW_WS[key]


_SIDE = {}
SERIAL = False            # bench.py / diagnostics: run the whole step on ONE stream (per-kernel durations without contention)


# Logical stream -> physical HIP stream.  MEASURED (MI355X, ROCm 7.2, tools/bisect_bench.sh): ROCm runs the streams of a
# process on GPU_MAX_HW_QUEUES = 4 hardware queues; a fifth active queue (GPU_MAX_HW_QUEUES >= 5, or one stream created with a
# priority) drops the step rate from ~265 to ~140 steps/s, and streams beyond the fourth silently SHARE a queue with an earlier
# one -- which of the step's chains then serialise depends on stream creation order (two extra prefetch streams cost 7 %).
# So the step uses exactly three side streams besides the caller's, and says which logical lanes ride together:
#   A: value pass of the critic phase (1) and, later in the step, the critic backward's weight-gradient lane (11)
#   B: actor pass (2: policy forward, and the whole actor phase on steps without the actor-critic term)
#   C: the actor backward's weight-gradient lane (12), small initialisations (3)
#   next step's uploads + geometry (20, 21) ride on A: behind the critic backward's weight gradients of the step before, they
#   start while that step's tail (actor phase / optimiser) still runs; on C they sat behind the actor backward's weight
#   gradients, i.e. until the very end of the step (round 3, same box: 316 -> 320 steps/s; on B: no change)
import os as _os
_PHYS = {1: "A", 11: "A", 2: "B", 12: "C", 3: "C", 20: "A", 21: "A"}
if _os.environ.get("GAD_STREAM_MAP"):                     # e.g. "1:A,11:A,2:B,12:C,3:C,20:D,21:D"
    _PHYS = dict((int(kv.split(":")[0]), kv.split(":")[1]) for kv in _os.environ["GAD_STREAM_MAP"].split(","))


def _masked_stream(dev, letter):
    """A/B switch GAD_CU_MASK_<letter> = "<n>" (the n lowest CU bits) or "even" / "odd" (every second CU) / hex words "w0,w1,...":
    the physical side stream `letter` is created with hipExtStreamCreateWithCUMask -- its kernels (e.g. the weight-gradient lanes)
    may then only occupy those CUs and leave the others to the chains (profiles/README.md round 5: measured, not kept)."""
    spec = _os.environ.get("GAD_CU_MASK_" + letter)
    if not spec:
        return None
    import ctypes as C
    if spec in ("even", "odd"):
        words = [0x55555555 if spec == "even" else 0xAAAAAAAA] * 8
    elif "," in spec or spec.startswith("0x"):
        words = [int(w, 16) for w in spec.split(",")]
    else:
        n = int(spec)
        words = [(0xFFFFFFFF if n >= 32 * (i + 1) else ((1 << max(0, n - 32 * i)) - 1)) & 0xFFFFFFFF for i in range(8)]
    rt = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    with torch.cuda.device(dev):
        rc = rt.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(len(words)), arr)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed (%d)" % rc)
    return torch.cuda.ExternalStream(st.value, device=dev)


def side_stream(device=None, which=0):
    """auxiliary HIP streams (per device, created lazily): 1 / 2 = whole encoder passes overlapped by runtime.FusedRuntime,
    3 = small initialisations, 10 + lane = weight-gradient GEMMs forked off a dX chain, 20 / 21 = input / geometry prefetch;
    several logical streams share a physical one (_PHYS)"""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if SERIAL:                                   # diagnostics: every fork / join degenerates to the caller's stream
        return torch.cuda.current_stream(dev)
    key = (dev, _PHYS.get(which, which))
    if key not in _SIDE:
        for k in ("A", "B", "C"):                # fixed creation order: the first three side streams get queues of their own
            if (dev, k) not in _SIDE:
                _SIDE[(dev, k)] = _masked_stream(dev, k) or torch.cuda.Stream(device=dev)
        if key not in _SIDE:
            _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


# Wavefront issue priority per lane (include/gaddpg.h: gad_stream_priority): "main:3,B:1" raises the GEMM-family launches of the
# caller's stream (the TD-target / critic chain the step waits for) and of the actor lane above the weight-gradient lanes'
# wavefronts they share SIMDs with.  s_setprio inside the kernels: no stream priority, the four-queue mapping is untouched.
LANE_PRIO = _os.environ.get("GAD_LANE_PRIO", "")
_PRIO_DONE = set()


def apply_lane_priorities(main):
    """once per caller's stream: hand the GAD_LANE_PRIO table to the library"""
    if not LANE_PRIO or SERIAL:
        return
    dev = main.device.index
    key = (dev, int(main.cuda_stream))
    if key in _PRIO_DONE:
        return
    _PRIO_DONE.add(key)
    from . import hip
    for kv in LANE_PRIO.split(","):
        name, pr = kv.split(":")
        st = main if name == "main" else side_stream(dev, {"A": 1, "B": 2, "C": 3}[name])
        hip.check(hip.lib().gad_stream_priority(hip.C.c_void_p(int(st.cuda_stream)), int(pr)), "gad_stream_priority")


CONCURRENT_DW = _os.environ.get("GAD_CONCURRENT_DW", "1") == "1"      # fork dW GEMMs onto side streams (they feed nothing but the optimiser)
FUSED_SA1_BWD = _os.environ.get("GAD_FUSED_SA1_BWD", "1") == "1"     # SA1 l3 / l2 backward: dX + dW in one kernel (gad_gemm_bwd)
RECOMP_SA1 = _os.environ.get("GAD_RECOMP_SA1", "0") == "1"   # SA1 layer 2 recomputes layer 1's output from the gathered rows (gad_gemm_fwd mode 2); layer 1 stores nothing in a pass that is never
                                                             # back-propagated.  OFF: bit-equal, measured neutral (layer 2 +1.9 us, layer 1 -2.6 us: the re-gather costs what z1's read cost)
INLINE_BN_BWD = _os.environ.get("GAD_INLINE_BN_BWD", "1") == "1"     # BatchNorm-backward coefficients formed in the dX / dW prologues
DEFER_BN_STAGES = tuple(int(c) for c in _os.environ.get("GAD_DEFER_BN_STAGES", "012"))     # SA stages it applies to (A/B)
DEFER_BN_WIDE = _os.environ.get("GAD_DEFER_BN_WIDE", "1") == "1"     # SA2 / SA3 layers 1, 2: BatchNorm finalised in the consumer GEMM's prologue
DEFER_DW = _os.environ.get("GAD_DEFER_DW", "0") == "1"              # A/B: weight-gradient GEMMs of a pass launched after its dX chain
FUSED_WIDE_BWD = _os.environ.get("GAD_FUSED_WIDE_BWD", "0") == "1"   # SA2 / SA3 backward: dX + dW in one kernel (round 4; the reduce of its
                                                                     # partial dW blocks forked onto the weight-gradient lane).  OFF: measured 4 - 6 % slower at B = 256 and 512 --
                                                                     # the step follows the length of its dX chain, and dW on its own lane is nearly free
WIDE_SLAB_ELEMS = 9 * 1024 * 1024        # floats per fused wide layer's partial-dW workspace (library option bwd_wide_slab <= 8)
DW_REDUCE_LATER = -2                     # include/gaddpg.h GAD_DW_REDUCE_LATER
DW_LANES = 1              # number of dW side streams (2 measured no faster: the overlapped kernels already saturate the GPU) the layers alternate between (each with its own partial workspace)


USE_C_PLANS = _os.environ.get("GAD_PLAN_C", "1") == "1"     # replay plans through gad_plan_run (one foreign call per plan segment);
                                                            # 0: the item-by-item Python walk (the same launches in the same order)
_ROUTES = hip.ROUTES       # tag -> kernel family (gad_last_kernel), learnt by the Python walk; cleared when a library option changes


class _Item(object):
    """one entry of a Plan: kind in {"call", "wait", "record", "wait_event", "zero", "memcpy", "py"}; `on` = the logical stream
    (0 = the stream that is current when the plan runs, else a side_stream `which`)"""
    __slots__ = ("kind", "on", "name", "f", "argv", "words", "kinds", "on2", "event", "holder", "tensor", "tag", "cpos", "clones")

    def __init__(self, kind, on=0):
        self.kind, self.on = kind, on
        self.name = self.f = self.argv = self.words = self.kinds = self.on2 = self.event = self.holder = self.tensor = self.tag = None
        self.cpos = []         # [(compiled plan, item index)]: where this item sits in gad_plans (an item may be shared by plans)
        self.clones = []       # copies made by Plan.extend(on=...): a patch of this item reaches them too


class EventHolder(object):
    """a hipEvent_t that both the host language (torch.cuda.Event) and a replayed plan can record / wait for: the torch event is
    recorded once at construction, which makes torch create the raw handle the plan items carry"""

    def __init__(self):
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream())
        self.handle = int(self.event.cuda_event)
        assert self.handle, "torch did not create the event"


def _pack_words(argv):
    """ctypes arguments -> (uint64 words, GAD_ARG_* kinds) of a gad_plan_add_call item"""
    import ctypes as C
    import struct
    words, kinds = [], []
    for x in argv:
        if isinstance(x, C.c_void_p):
            words.append(int(x.value or 0)); kinds.append(1)
        elif isinstance(x, (C.Structure, C.Array)):
            words.append(C.addressof(x)); kinds.append(1)
        elif isinstance(x, C.c_float):
            words.append(struct.unpack("<I", struct.pack("<f", x.value))[0]); kinds.append(2)
        elif isinstance(x, C.c_double):
            words.append(struct.unpack("<Q", struct.pack("<d", x.value))[0]); kinds.append(3)
        elif isinstance(x, (C.c_longlong, C.c_ulonglong, C.c_size_t)):
            words.append(int(x.value) & 0xFFFFFFFFFFFFFFFF); kinds.append(1)
        elif isinstance(x, (C.c_int, C.c_uint)):
            words.append(int(x.value) & 0xFFFFFFFFFFFFFFFF); kinds.append(0)
        else:
            raise TypeError("cannot pack %r into a plan item" % (x,))
    return words, kinds


def _float_word(v):
    import struct
    return struct.unpack("<I", struct.pack("<f", float(v)))[0]


class _Compiled(object):
    """a Plan lowered to gad_plans: the segments between host callbacks, each replayed by one gad_plan_run"""

    def __init__(self, plan):
        import ctypes as C
        L = hip.lib()
        self.handles = []          # one gad_plan per segment
        self.steps = []            # [("c", handle index) | ("py", item)]
        self.whichs = [0]          # dense lane -> logical stream
        self.timed = []            # [(segment handle index, item index in it, tag)] of the tagged launches
        lane_of = {0: 0}

        def lane(w):
            if w not in lane_of:
                lane_of[w] = len(self.whichs)
                self.whichs.append(w)
            return lane_of[w]

        cur = None
        for it in plan.calls:
            if it.kind == "py":
                cur = None
                self.steps.append(("py", it))
                continue
            if cur is None:
                h = C.c_void_p()
                hip.check(L.gad_plan_create(C.byref(h)), "gad_plan_create")
                self.handles.append(h)
                self.steps.append(("c", len(self.handles) - 1))
                cur = h
            if it.kind == "call":
                if it.words is None:
                    it.words, it.kinds = _pack_words(it.argv)
                n = len(it.words)
                rc = L.gad_plan_add_call(cur, it.name.encode(), (C.c_uint64 * n)(*it.words), (C.c_uint8 * n)(*it.kinds), n, lane(it.on))
                if rc >= 0 and it.tag is not None and it.name in _TIMED_CALLS:
                    self.timed.append((len(self.handles) - 1, rc, it.tag))
            elif it.kind == "wait":
                rc = L.gad_plan_add_wait(cur, lane(it.on), lane(it.on2))
            elif it.kind == "record":
                rc = L.gad_plan_add_record(cur, lane(it.on), C.c_void_p(it.holder.handle if it.holder is not None else 0))
            elif it.kind == "wait_event":
                rc = L.gad_plan_add_wait_event(cur, lane(it.on), C.c_void_p(it.holder.handle if it.holder is not None else 0))
            elif it.kind == "zero":
                t = it.tensor
                rc = L.gad_plan_add_memset(cur, C.c_void_p(t.data_ptr()), C.c_longlong(t.numel() * t.element_size()), lane(it.on))
            elif it.kind == "memcpy":
                rc = L.gad_plan_add_memcpy(cur, C.c_void_p(it.words[0]), C.c_void_p(it.words[1]), C.c_longlong(it.words[2]), lane(it.on))
            else:
                raise RuntimeError("unknown plan item %r" % it.kind)
            if rc < 0:
                raise RuntimeError("plan item %s failed to compile (status %d): %s" % (it.name or it.kind, rc, L.gad_last_error().decode()))
            it.cpos.append((cur, rc))
        self.n_items = len(plan.calls)
        self._tables = {}

    def table(self, main_handle):
        """lane -> hipStream_t table for a run whose current stream is main_handle"""
        import ctypes as C
        key = (int(main_handle or 0), SERIAL)
        t = self._tables.get(key)
        if t is None:
            hs = [int(main_handle or 0)] + [int(side_stream(which=w).cuda_stream) for w in self.whichs[1:]]
            t = self._tables[key] = (C.c_void_p * len(hs))(*hs)
        return t

    def __del__(self):
        import sys
        if sys is None or sys.is_finalizing():      # interpreter exit: the HIP runtime may be gone already, the process frees everything
            return
        try:
            L = hip.lib()
            for h in self.handles:
                L.gad_plan_destroy(h)
        except Exception:
            pass


class Plan(object):
    """A recorded sequence of C-ABI calls over static buffers, with the stream each one is enqueued on.  `on` = 0 names the
    stream that is current when the plan runs, any other value a logical side stream (side_stream(which=on)).  Calls marked
    `side=k` run on weight-gradient lane k (= logical stream 10 + k) between a fork (the lane waits for everything recorded so
    far on the main stream) and the next join (main waits for the lane): the weight-gradient GEMMs are off the dX critical path.
    run() replays the list through gad_plan_run (include/gaddpg.h section H: one foreign call per plan, or per segment between
    host callbacks); the item-by-item Python walk remains for the serialised / timing-probe diagnostics."""

    def __init__(self):
        self.calls = []      # _Items
        self.keep = []       # keeps ctypes structs / tensors alive
        self._c = None

    # ---- recording -------------------------------------------------------------------------------------------------
    def _add(self, it):
        self.calls.append(it)
        self._c = None
        return it

    def tag_last(self, tag):
        self.calls[-1].tag = tag

    def call(self, name, *a, side=False, on=None):
        """-> the item (for patch()); arguments as hip.call takes them, plus ctypes structures / arrays (passed by address)"""
        import ctypes as C
        it = _Item("call", (10 + int(side)) if side else (on or 0))
        it.name, it.f = name, getattr(hip.lib(), name)
        it.argv = hip._args(*a)
        self.keep.append(a)
        return self._add(it)

    def call_struct(self, name, s, side=False, on=None):
        return self.call(name, s, side=side, on=on)

    def zero(self, t, on=0):
        it = _Item("zero", on)
        it.tensor = t
        return self._add(it)

    def zero_multi(self, tensors, on=0):
        """clear several buffers with ONE launch per six of them (gad_zero_buffers) instead of a fill kernel each"""
        import ctypes as C
        ts = [t for t in tensors if t is not None]
        for i in range(0, len(ts), 6):
            grp = ts[i:i + 6]
            a = []
            for t in grp:
                a += [t, C.c_longlong(t.numel() * t.element_size())]
            a += [None, C.c_longlong(0)] * (6 - len(grp))
            self.call("gad_zero_buffers", *a, on=on)

    def memcpy(self, dst, src, nbytes, on=0):
        """hipMemcpyAsync(dst, src, nbytes) on the lane's stream; dst / src: raw addresses (device, or pinned host); patchable
        words 0 (dst) and 1 (src)"""
        it = _Item("memcpy", on)
        it.words = [int(dst), int(src), int(nbytes)]
        return self._add(it)

    def fn(self, f, side=False, on=None):
        """host callback at this point of the plan; on a side lane: called with that lane's stream current (whatever it
        enqueues -- a collective -- is ordered after the lane's launches, not after the main stream's)"""
        it = _Item("py", (10 + int(side)) if side else (on or 0))
        it.f = f
        return self._add(it)

    def wait(self, waiter, signal):
        """logical stream `waiter` waits for everything enqueued so far on `signal` (plan-owned event)"""
        it = _Item("wait", waiter)
        it.on2 = signal
        it.event = torch.cuda.Event()
        return self._add(it)

    def fork(self, k=1):
        return self.wait(10 + k, 0)

    def join(self, k=1):
        return self.wait(0, 10 + k)

    def record(self, holder, on=0):
        """record the caller-owned event (EventHolder, or None = no-op until patched) on the lane's stream"""
        it = _Item("record", on)
        it.holder = holder
        return self._add(it)

    def wait_event(self, holder, on=0):
        it = _Item("wait_event", on)
        it.holder = holder
        return self._add(it)

    def extend(self, other, on=0):
        """append another plan's items; on != 0: the other plan's main stream becomes logical stream `on` here"""
        import copy
        for it in other.calls:
            if on and (it.on == 0 or (it.kind == "wait" and it.on2 == 0)):
                src, it = it, copy.copy(it)
                src.clones.append(it)
                it.cpos, it.clones = [], []
                if it.kind == "wait":
                    it.event = torch.cuda.Event()
                    it.on2 = on if it.on2 == 0 else it.on2
                it.on = on if it.on == 0 else it.on
            self.calls.append(it)
        self.keep += other.keep
        self.keep.append(other)
        self._c = None

    # ---- per-run edits -----------------------------------------------------------------------------------------------
    @staticmethod
    def patch(it, index, value):
        """replace argument `index` of a call item (a Python float / int / hip.Ptr / tensor / None, converted like hip.call
        does), or word `index` of a memcpy item (an address), or the event of a record / wait_event item (EventHolder | None)"""
        import ctypes as C
        if it.kind == "call":
            new = hip._args(value)[0]
            it.argv[index] = new
            w, k = _pack_words([new])
            if it.words is not None:
                assert it.kinds[index] == k[0], "patch changes the argument's kind"
                it.words[index] = w[0]
            word = w[0]
        elif it.kind == "memcpy":
            it.words[index] = word = int(value)
        else:
            it.holder = value
            word = value.handle if value is not None else 0
        L = hip.lib()
        for h, i in it.cpos:
            hip.check(L.gad_plan_patch(h, i, index, C.c_uint64(word)), "gad_plan_patch")
        for cl in it.clones:           # (argv / words lists are shared with the clones; their compiled words and holders are not)
            if cl.kind not in ("call", "memcpy"):
                cl.holder = it.holder
            for h, i in cl.cpos:
                hip.check(L.gad_plan_patch(h, i, index, C.c_uint64(word)), "gad_plan_patch")

    # ---- replay -------------------------------------------------------------------------------------------------------
    def run(self):
        timed = TIMING["enabled"]
        if not USE_C_PLANS or (timed and any(it.tag is not None and it.name in _TIMED_CALLS and it.tag not in _ROUTES
                                              for it in self.calls)):
            return self.run_py()
        import ctypes as C
        c = self._c
        if c is None or c.n_items != len(self.calls):
            c = self._c = _Compiled(self)
        L = hip.lib()
        main = torch.cuda.current_stream()
        table = c.table(main.cuda_stream)
        if timed:
            for hi, idx, tag in c.timed:
                if TIMING["next"] < TIMING["slots"].shape[0] and _timing_wants(tag):
                    k = TIMING["next"]
                    TIMING["next"] = k + 1
                    TIMING["tags"].append(tag)
                    TIMING["routed"][tag] = _ROUTES[tag]
                    hip.check(L.gad_plan_arm_timing(c.handles[hi], idx, C.c_void_p(TIMING["slots"].data_ptr() + 16 * TIMING_WAVES * k)),
                              "gad_plan_arm_timing")
        n = len(table)
        for kind, x in c.steps:
            if kind == "c":
                rc = L.gad_plan_run(c.handles[x], table, n, 0, -1)
                if rc != 0:
                    hip.check(rc, "gad_plan_run")
            elif x.on:
                with torch.cuda.stream(side_stream(which=x.on)):
                    x.f()
            else:
                x.f()

    def run_py(self):
        """the same launches, item by item from Python (diagnostics; also learns the kernel family of every tagged launch)"""
        import ctypes as C
        main = torch.cuda.current_stream()
        st = hip.stream()
        sides = {}            # logical stream -> (torch stream, raw handle)
        timed = TIMING["enabled"]

        def stream_of(w):
            if w == 0:
                return main, st
            if w not in sides:
                so = side_stream(which=w)
                sides[w] = (so, C.c_void_p(so.cuda_stream))
            return sides[w]

        for it in self.calls:
            kind = it.kind
            if kind == "wait":
                a, b = stream_of(it.on2)[0], stream_of(it.on)[0]
                it.event.record(a)
                if a.cuda_stream != b.cuda_stream:
                    b.wait_event(it.event)
                continue
            so, q = stream_of(it.on)
            if kind == "call":
                took_slot = False
                if timed and it.name in _TIMED_CALLS and it.tag is not None and TIMING["next"] < TIMING["slots"].shape[0] and \
                        _timing_wants(it.tag):
                    k = TIMING["next"]
                    TIMING["next"] = k + 1
                    TIMING["tags"].append(it.tag)
                    hip.lib().gad_timing_slot(C.c_void_p(TIMING["slots"].data_ptr() + 16 * TIMING_WAVES * k))   # consumed by the call below
                    took_slot = True
                argv = [C.byref(x) if isinstance(x, (C.Structure, C.Array)) else x for x in it.argv]
                hip.check(it.f(*(argv + [q])), it.name)
                if it.tag is not None and it.name in _TIMED_CALLS:
                    _ROUTES[it.tag] = hip.lib().gad_last_kernel().decode()      # kernel family the call routed to
                    if took_slot:
                        TIMING["routed"][it.tag] = _ROUTES[it.tag]
            elif kind == "zero":
                with torch.cuda.stream(so):
                    it.tensor.zero_()
            elif kind == "py":
                if it.on:
                    with torch.cuda.stream(so):
                        it.f()
                else:
                    it.f()
            elif kind == "record":
                if it.holder is not None:
                    it.holder.event.record(so)
            elif kind == "wait_event":
                if it.holder is not None:
                    so.wait_event(it.holder.event)
            elif kind == "memcpy":
                rt = _hip_runtime()
                rc = rt.hipMemcpyAsync(C.c_void_p(it.words[0]), C.c_void_p(it.words[1]), C.c_size_t(it.words[2]), 4, q)
                if rc != 0:
                    raise RuntimeError("hipMemcpyAsync failed (%d)" % rc)


_HIPRT = []


def _hip_runtime():
    if not _HIPRT:
        import ctypes as C
        _HIPRT.append(C.CDLL("libamdhip64.so"))
    return _HIPRT[0]


def _timing_wants(tag):
    t = TIMING["tag"]
    return t == "*" or tag == t or (isinstance(t, (set, frozenset, tuple, list)) and tag in t)


def timing_routes():
    """{tag: kernel family} of the launches timed since the last timing_start() (gad_last_kernel after each)"""
    return dict(TIMING["routed"])


# ----------------------------------------------------------------------------------------------
# encoder forward / backward plans
# ----------------------------------------------------------------------------------------------
def _bn_vec(slot, enc, m, which):
    return _ptr(getattr(slot, which), enc.bn_off[m.bn_index])


def _finalize(plan, enc, slot, m, count, train=True, update_running=True):
    o, tot = enc.bn_off[m.bn_index], slot.tot
    if not train:        # eval mode: affine from the running statistics (torch BatchNorm eval semantics)
        plan.call("gad_bn_eval_affine", enc.flat.p_gamma(m), enc.flat.p_beta(m), _ptr(enc.running_mean, o),
                  _ptr(enc.running_var, o), m.n_out, BN_EPS, _bn_vec(slot, enc, m, "scale"), _bn_vec(slot, enc, m, "shift"))
        return
    plan.call("gad_bn_finalize", _ptr(slot.stats, o, 8), _ptr(slot.stats, tot + o, 8), 2 * tot, enc.flat.p_gamma(m),
              enc.flat.p_beta(m), m.n_out, hip.Dbl(count), BN_EPS, BN_MOMENTUM,
              _ptr(enc.running_mean, o) if update_running else None, _ptr(enc.running_var, o) if update_running else None,
              _bn_vec(slot, enc, m, "scale"), _bn_vec(slot, enc, m, "shift"), _bn_vec(slot, enc, m, "mean"),
              _bn_vec(slot, enc, m, "istd"))


def _gather_src(geo, slot, s, action):
    """input description of SA stage s's first layer"""
    r = geo.rows[s]
    if s == 0:
        return dict(src_xyz=_ptr(geo.xyz), ctr_xyz=_ptr(geo.new_xyz1), feat=_ptr(geo.feat0), feat_c=4,
                    action=_ptr(action), act_c=6 if action is not None else 0, grp_per_sample=geo.M1,
                    row_pt=_ptr(r["pt"]), row_grp=_ptr(r["grp"]))
    if s == 1:
        return dict(src_xyz=_ptr(geo.new_xyz1), ctr_xyz=_ptr(geo.new_xyz2), feat=_ptr(slot.F[0]),
                    feat_c=slot.F[0].shape[1], action=None, act_c=0, grp_per_sample=geo.M2,
                    row_pt=_ptr(r["pt"]), row_grp=_ptr(r["grp"]))
    return dict(src_xyz=_ptr(geo.new_xyz2), ctr_xyz=None, feat=_ptr(slot.F[1]), feat_c=slot.F[1].shape[1],
                action=None, act_c=0, grp_per_sample=1, row_pt=_ptr(r["pt"]), row_grp=_ptr(r["grp"]))


def _layer_input(enc, slot, geo, s, l, action, fin=None, forward=False):
    """gad_gemm_fwd_args fields describing the INPUT of SA stage s layer l (s==3: FC layer l).
    fin = (count, update_running): the input layer's train-mode BatchNorm is finalised by THIS launch (gad_gemm_fwd_args
    in_*: no gad_bn_finalize launch between the two GEMMs)"""
    if s < 3:
        r = geo.rows[s]
        base = dict(n_rows_dev=_ptr(r["n"]), n_rows=r["cap"], row_w=_ptr(r["w"]))
        if l == 0:
            base.update(mode=1, c_in=enc.sa_mats[s][0].k_in, **_gather_src(geo, slot, s, action))
        else:
            pm = enc.sa_mats[s][l - 1]
            base.update(mode=0, zin=_ptr(slot.Z[s][l - 1]), zin_pitch=pm.n_out, c_in=pm.n_out,
                        scale=_bn_vec(slot, enc, pm, "scale"), shift=_bn_vec(slot, enc, pm, "shift"), relu=1)
            if forward and recomputed_input(enc, geo, s, l):       # (the backward pass reads the stored z1)
                # layer 1's output is recomputed from the gathered rows inside this launch (bit-equal to what layer 1 stored)
                base.update(mode=2, pre_W=enc.flat.p_w(pm), pre_Kp=pm.Kp, **_gather_src(geo, slot, s, action))
            if fin is not None:
                base.update(_input_bn(enc, slot, pm, fin))
        return base
    B = slot.B
    if l == 0:
        return dict(n_rows=B, mode=0, zin=_ptr(slot.F[2]), zin_pitch=slot.F[2].shape[1], c_in=slot.F[2].shape[1],
                    relu=0, ones_col=enc.fc_mats[0].ones_col)
    pm = enc.fc_mats[0]
    base = dict(n_rows=B, mode=0, zin=_ptr(slot.Zfc[0]), zin_pitch=pm.n_out, c_in=pm.n_out,
                scale=_bn_vec(slot, enc, pm, "scale"), shift=_bn_vec(slot, enc, pm, "shift"), relu=1,
                ones_col=enc.fc_mats[1].ones_col)
    if fin is not None:
        base.update(_input_bn(enc, slot, pm, fin))
    return base


def recomputed_input(enc, geo, s, l):
    """SA1 layer 2 on the streaming kernel's shapes: gad_gemm_fwd mode 2"""
    if not (RECOMP_SA1 and s == 0 and l == 1):
        return False
    if hip.get_option("mfma_split"):
        # the recomputing launch (gad_gemm_fwd mode 2) exists on the f32 MFMA only: beside split-bf16 layers the recomputed and the
        # stored forms of layer 1's output would come from different arithmetic
        raise RuntimeError("GAD_RECOMP_SA1=1 needs GAD_OPT_mfma_split=0 (the recomputing SA1 launch has no split-bf16 form)")
    m0, m1 = enc.sa_mats[0][0], enc.sa_mats[0][1]
    return m0.n_out == 64 and m1.n_out == 64 and m1.Kp == 64 and m0.Kp in (8, 16) and geo.rows[0]["cap"] >= 32768


def _input_bn(enc, slot, pm, fin):
    """gad_gemm_fwd_args in_* block: layer pm's train-mode BatchNorm, finalised by the launch that consumes its output"""
    o, tot = enc.bn_off[pm.bn_index], slot.tot
    return dict(in_stat_sum=_ptr(slot.stats, o, 8), in_stat_sq=_ptr(slot.stats, tot + o, 8), in_stat_stride=2 * tot,
                in_count=float(fin[0]), in_gamma=enc.flat.p_gamma(pm), in_beta=enc.flat.p_beta(pm), in_eps=BN_EPS,
                in_momentum=BN_MOMENTUM, in_running_mean=_ptr(enc.running_mean, o) if fin[1] else None,
                in_running_var=_ptr(enc.running_var, o) if fin[1] else None,
                in_mean=_bn_vec(slot, enc, pm, "mean"), in_istd=_bn_vec(slot, enc, pm, "istd"))


def plan_running_update(enc, slot):
    """apply the momentum update of a forward pass planned with update_running=False (one launch, all layers)"""
    plan = Plan()
    if not hasattr(slot, "bn_count"):
        cnt = torch.empty(slot.tot, dtype=torch.float32)
        for s in range(3):
            for m in enc.sa_mats[s]:
                o = enc.bn_off[m.bn_index]
                cnt[o:o + m.n_out] = float(slot.geo.counts[s])
        for m in enc.fc_mats:
            o = enc.bn_off[m.bn_index]
            cnt[o:o + m.n_out] = float(slot.B)
        slot.bn_count = cnt.to(slot.mean.device)
    plan.call("gad_bn_running_update", slot.mean, slot.istd, slot.bn_count, slot.tot, BN_EPS, BN_MOMENTUM,
              enc.running_mean, enc.running_var)
    return plan


def plan_encoder_forward(enc, slot, action=None, train=True, update_running=True):
    """SA1 -> SA2 -> SA3 -> FC.  Every GEMM is followed by its gad_bn_finalize launch (train mode) and the consumers read the
    published scale / shift.  The segment max-pool of each stage is folded into the epilogue of its third GEMM (packed
    arg-max keys, gad_gemm_fwd_args.pool_key); gad_pool_finalize then finalises that layer's BatchNorm and turns the keys
    into pooled features, arg-max rows and their raw values.  A slot built with with_backward=False does not store the
    third layers' outputs at all.
    update_running=False: batch statistics only; the running-statistics momentum update is applied later by
    plan_running_update (for a pass that overlaps another pass of the same network on a second stream)."""
    geo = slot.geo
    plan = Plan()
    plan.zero(slot.stats)
    tot = slot.tot
    if not train and slot.with_backward:
        # eval-mode BatchNorm under a training step (update_parameters(test=True), reference core/agent.py:276-280): the backward
        # pass normalises with the RUNNING statistics too -- the vectors it reads as the layers' mean / inverse deviation
        def running_vectors():
            slot.mean.copy_(enc.running_mean)
            slot.istd.copy_(torch.rsqrt(enc.running_var + BN_EPS))
        plan.fn(running_vectors)

    def deferred(s, l):
        """layer (s, l)'s BatchNorm is finalised in the prologue of its consumer (s, l + 1): the SA2 / SA3 layers, whose
        consumers are the wide-tile forward kernel"""
        if s == 3:
            return DEFER_BN_WIDE and train and 3 in DEFER_BN_STAGES and l == 0      # FC 1, finalised by FC 2 (skinny kernel)
        return DEFER_BN_WIDE and train and s in DEFER_BN_STAGES and l < 2

    def gemm(m, zout, s, l, tag, pool=None):
        o = enc.bn_off[m.bn_index]
        fin = ((geo.counts[s] if s < 3 else float(slot.B)), update_running) if (l > 0 and deferred(s, l - 1)) else None
        kw = _layer_input(enc, slot, geo, s, l, action, fin=fin, forward=True)
        if pool is not None:
            kw.update(pool_key=_ptr(slot.key[s], 0, 8), pool_row_grp=_ptr(geo.rows[s]["grp"]), pool_gamma=enc.flat.p_gamma(m))
        if s == 0 and l == 0 and not slot.with_backward and recomputed_input(enc, geo, 0, 1):
            zout = None         # statistics only: layer 2 recomputes this output and nothing else reads it in a forward-only pass
        kw.update(enc.flat.split_fwd_kw(m))
        a = _fwd_args(W=enc.flat.p_w(m), Kp=m.Kp, n_out=[m.n_out], zout=_ptr(zout), zout_pitch=m.n_out,
                      stat_sum=_ptr(slot.stats, o, 8), stat_sq=_ptr(slot.stats, tot + o, 8), stat_stride=2 * tot, **kw)
        if s < 2:
            plan.call("gad_grid_rows_hint", hip.Ptr(geo.rows_hint.ctypes.data + 4 * s))
        plan.call_struct("gad_gemm_fwd", a)
        plan.tag_last(tag)

    def pool_finalize(s, m, count):
        o = enc.bn_off[m.bn_index]
        r = geo.rows[s]
        if train:
            stats = (_ptr(slot.stats, o, 8), _ptr(slot.stats, tot + o, 8), 2 * tot, hip.Dbl(count))
            run = (_ptr(enc.running_mean, o), _ptr(enc.running_var, o)) if update_running else (None, None)
        else:           # eval mode: scale / shift from the running statistics, computed before the keys are decoded
            _finalize(plan, enc, slot, m, 0.0, False)
            stats, run = (None, None, 0, hip.Dbl(1.0)), (None, None)
        plan.call("gad_pool_finalize", _ptr(slot.key[s], 0, 8), m.n_out, r["G"], r["off"], *stats, enc.flat.p_gamma(m),
                  enc.flat.p_beta(m), BN_EPS, BN_MOMENTUM, *run, _bn_vec(slot, enc, m, "scale"), _bn_vec(slot, enc, m, "shift"),
                  _bn_vec(slot, enc, m, "mean"), _bn_vec(slot, enc, m, "istd"), slot.F[s], slot.argmax[s], slot.zmax[s])
        plan.tag_last("pool.sa%d" % (s + 1))

    for s in range(3):
        for l, m in enumerate(enc.sa_mats[s]):
            gemm(m, slot.Z[s][l], s, l, "fwd.sa%d.l%d" % (s + 1, l + 1), pool=(s if l == 2 else None))
            if l < 2 and not deferred(s, l):
                _finalize(plan, enc, slot, m, geo.counts[s], train, update_running)
        pool_finalize(s, enc.sa_mats[s][2], geo.counts[s])
    for l, m in enumerate(enc.fc_mats):
        gemm(m, slot.Zfc[l], 3, l, "fwd.fc%d" % (l + 1))
        if not deferred(3, l):
            _finalize(plan, enc, slot, m, float(slot.B), train, update_running)
    return plan


def _coef_ptrs(slot, enc, m):
    o, tot = enc.bn_off[m.bn_index], slot.tot
    return _ptr(slot.coef, o), _ptr(slot.coef, tot + o), _ptr(slot.coef, 2 * tot + o)


def _bn_coef(plan, enc, slot, m, count, want_dw):
    o, tot = enc.bn_off[m.bn_index], slot.tot
    P, Q, S = _coef_ptrs(slot, enc, m)
    gacc = enc.flat.gacc
    plan.call("gad_bn_bwd_coef", _ptr(slot.bstats, o, 8), _ptr(slot.bstats, tot + o, 8), 2 * tot,
              _bn_vec(slot, enc, m, "scale"), _bn_vec(slot, enc, m, "mean"), _bn_vec(slot, enc, m, "istd"),
              m.n_out, hip.Dbl(count), P, Q, S, _ptr(gacc, m.g_off, 8) if want_dw else None,
              _ptr(gacc, m.b_off, 8) if want_dw else None)


def plan_encoder_backward(enc, slot, g_fc2, action=None, want_dw=True, want_daction=False, dw_lane=1, zero_scatter=True,
                          early_hook=None, train=True):
    # zero_scatter=False: the caller has cleared slot.dF[0], slot.dF[1] (and slot.daction) at the head of its plan
    """Backward of plan_encoder_forward.  g_fc2 (B, 512) is dLoss/d(relu(bn(Zfc[1]))) WITH the ReLU mask applied, as
    produced by the consumer head's dX kernel (store_masked), whose epilogue must also have filled this slot's bstats
    for fc[1].  No BatchNorm launch: every layer's backward coefficients P, Q, S are formed in the prologue of its
    dX / dW kernels (gad_bn_bwd); every dX stores the next gradient already masked by the previous layer's ReLU.
    Weight gradients accumulate (f64) into enc.flat.gacc when want_dw; their GEMMs are forked onto side stream
    `dw_lane` (two backward passes that may run concurrently -- critic and actor -- get different lanes: each lane has
    its own split-K workspace and is stream-ordered).
    early_hook(plan): called once the weight gradients of everything but SA1 are complete (FC head, SA3, SA2 = 99 % of
    the encoder's parameters; the dW lanes are joined first): a data-parallel run converts and all-reduces that bucket
    there, under the SA1 backward -- the longest stage of the pass."""
    geo = slot.geo
    plan = Plan()
    B = slot.B
    tot = slot.tot
    # train=False: the backward of EVAL-mode BatchNorm (y = scale * z + shift with constants from the running statistics):
    # dZ = scale * dY, no batch-statistics terms.  The coefficient kernels form Q and S as (sums) / count, so an infinite count gives
    # exactly that (P = scale, Q = S = 0) through the same launches; dgamma / dbeta are the same sums as in train mode, taken with the
    # running mean / inverse deviation that plan_encoder_forward(train=False) put into slot.mean / slot.istd
    INF = float("inf")
    n_fc = float(B) if train else INF

    def prev_stats(pm, zprev):
        o = enc.bn_off[pm.bn_index]
        return dict(zprev=_ptr(zprev), zprev_pitch=pm.n_out, prev_scale=_bn_vec(slot, enc, pm, "scale"),
                    prev_shift=_bn_vec(slot, enc, pm, "shift"), prev_mean=_bn_vec(slot, enc, pm, "mean"),
                    prev_istd=_bn_vec(slot, enc, pm, "istd"), prev_dbeta=_ptr(slot.bstats, o, 8),
                    prev_dgamma=_ptr(slot.bstats, tot + o, 8), stat_stride=2 * tot, store_masked=1)

    def bn_dz(m, z, count, accumulate, G=None, pooled=None, row_w=None, inline=False):
        d = dict(z=_ptr(z), z_pitch=m.n_out, scale=_bn_vec(slot, enc, m, "scale"),
                 shift=_bn_vec(slot, enc, m, "shift"), relu=1, premasked=1, row_w=row_w, c=m.n_out)
        d["coefP"], d["coefQ"], d["coefS"] = _coef_ptrs(slot, enc, m)
        if inline:
            # P, Q, S formed in the consuming launch's prologue (no gad_bn_bwd_coef launch); the launch given `accumulate` adds
            # dgamma / dbeta to the gradient arena
            o = enc.bn_off[m.bn_index]
            d.update(bn_dbeta=_ptr(slot.bstats, o, 8), bn_dgamma=_ptr(slot.bstats, tot + o, 8), bn_stride=2 * tot, bn_count=float(count),
                     bn_mean=_bn_vec(slot, enc, m, "mean"), bn_istd=_bn_vec(slot, enc, m, "istd"))
            if accumulate and want_dw:
                d.update(gacc_gamma=_ptr(enc.flat.gacc, m.g_off, 8), gacc_beta=_ptr(enc.flat.gacc, m.b_off, 8))
        if pooled is None:
            d.update(gmode=0, G=_ptr(G), g_pitch=m.n_out)
        else:
            d.update(gmode=1, argmax=_ptr(pooled[0]), dout=_ptr(pooled[1]), row_grp=_ptr(pooled[2]))
        return _dz(**d)

    def dx(rows_kw, dz, m, k_valid, **epi):
        a = hip.GemmDxArgs()
        a.n_rows_dev = rows_kw.get("n_rows_dev")
        a.n_rows = rows_kw["n_rows"]
        a.dz = dz
        a.n_groups = 1
        a.n_out[0] = m.n_out
        a.W = enc.flat.p_w(m)
        a.Kp = m.Kp
        a.k_valid = k_valid
        a.grp_per_sample = 1
        for k, v in dict(epi, **enc.flat.split_t_kw(m)).items():
            setattr(a, k, v)
        stage = {"sa1": 0, "sa2": 1}.get(rows_kw.get("name"))
        if stage is not None:
            plan.call("gad_grid_rows_hint", hip.Ptr(geo.rows_hint.ctypes.data + 4 * stage))
        if fused_dw:                       # dX and dW in one pass on this stream (gad_gemm_bwd): SA1 l3 / l2, SA2, SA3
            aw = fused_dw.pop()
            plan.call("gad_gemm_bwd", a, aw)
            plan.tag_last("bwd.%s.l%d" % (rows_kw.get("name", "fc"), rows_kw.get("layer", 0)))
            if aw.row_splits == DW_REDUCE_LATER:
                # the f64 sum of the kernel's partial dW blocks leaves the dX chain: forked onto the weight-gradient lane
                # (each layer has a workspace of its own, so the next layer's kernel cannot overwrite blocks still to be summed)
                lane = dw_lane if CONCURRENT_DW else 0
                dw_lanes.append(lane)
                if lane:
                    plan.fork(lane)
                plan.call("gad_gemm_dw_reduce", a, aw, side=lane)
            return
        plan.call_struct("gad_gemm_dx", a)
        plan.tag_last("dx.%s.l%d" % (rows_kw.get("name", "fc"), rows_kw.get("layer", 0)))

    dw_lanes = []
    fused_dw = []
    deferred_dw = []
    has_dx_now = [True]                # (set by layer(): a layer without a dX launch cannot take the fused call)

    def dw(s, l, dz, m, action):
        if not want_dw:
            return
        a = hip.GemmDwArgs()
        a.inp = _fwd_args(Kp=m.Kp, n_out=[m.n_out], w_off=[m.w_off], **_layer_input(enc, slot, geo, s, l, action))
        a.dz = dz
        a.gacc = _ptr(enc.flat.gacc)
        if FUSED_SA1_BWD and s == 0 and l > 0:
            # the two SA1 layers whose dX and dW both stream the same ~1e5-row tensors: side by side on two streams they
            # slow each other down to 2x their stand-alone durations; one kernel reads the tensors once for both
            # (workspace per backward pass: the critic's and the actor's may run at the same time)
            ws = dw_workspace(enc.flat.device, elems=256 * 128 * 64, lane=200 + dw_lane)
            a.partial, a.partial_elems = _ptr(ws), ws.numel()
            fused_dw.append(a)
            return
        if FUSED_WIDE_BWD and s in (1, 2) and has_dx_now[0]:
            ws = dw_workspace(enc.flat.device, elems=WIDE_SLAB_ELEMS, lane=300 + 10 * dw_lane + 3 * s + l)
            a.partial, a.partial_elems = _ptr(ws), ws.numel()
            a.row_splits = DW_REDUCE_LATER
            fused_dw.append(a)
            return
        lane = dw_lane + (len(dw_lanes) % DW_LANES) if CONCURRENT_DW else 0
        dw_lanes.append(lane)
        ws = dw_workspace(enc.flat.device, lane=lane)
        a.partial, a.partial_elems = _ptr(ws), ws.numel()
        tag = "dw.%s.l%d" % ("sa%d" % (s + 1) if s < 3 else "fc", l + 1)
        if DEFER_DW and lane:                 # A/B: every weight-gradient GEMM of the pass behind its dX chain (one fork at the end)
            deferred_dw.append((a, lane, tag))
            return
        if lane:
            plan.fork(lane)
        plan.call_struct("gad_gemm_dw", a, side=lane)
        plan.tag_last(tag)

    def layer(s, l, m, z, count, has_dx, **src):
        """(dz for the dW, dz for the dX) of one layer: the dX carries the arena accumulation of dgamma / dbeta when
        there is one, else the dW does"""
        # SA3, SA2 and SA1 layers 3 / 2: the coefficients are formed in the dX / dW launches' own prologues (round 4); SA1
        # layer 1 and the FC layers keep the launch (their weight-gradient kernels read P / Q / S per lane)
        inline = INLINE_BN_BWD and s < 3 and not (s == 0 and l == 0)
        if not inline:
            _bn_coef(plan, enc, slot, m, count, want_dw)
        has_dx_now[0] = bool(has_dx)
        d_dw = bn_dz(m, z, count, not has_dx, inline=inline, **src)
        dw(s, l, d_dw, m, action)
        return bn_dz(m, z, count, True, inline=inline, **src) if has_dx else None

    # ---- FC head ----
    fc1, fc2 = enc.fc_mats
    d = layer(3, 1, fc2, slot.Zfc[1], n_fc, True, G=g_fc2)
    dx(dict(n_rows=B, layer=2), d, fc2, fc1.n_out, epilogue=0, gout=_ptr(slot.Gfc), gout_pitch=fc1.n_out,
       **prev_stats(fc1, slot.Zfc[0]))
    d = layer(3, 0, fc1, slot.Zfc[0], n_fc, True, G=slot.Gfc)
    dx(dict(n_rows=B, layer=1), d, fc1, slot.F[2].shape[1], epilogue=0, gout=_ptr(slot.dF[2]), gout_pitch=slot.F[2].shape[1])
    # ---- SA3 -> SA1 ----
    for s in (2, 1, 0):
        if s == 0 and early_hook is not None and want_dw:
            # on the dW lane itself: its launches so far are the weight gradients of everything but SA1 (each fork made
            # the lane wait for the main stream up to that point, the heads' dW GEMMs included); no join, the dX chain
            # on the main stream goes straight on into SA1
            early_hook(plan, dw_lanes[-1] if dw_lanes else 0)
        r = geo.rows[s]
        rows_kw = dict(n_rows_dev=_ptr(r["n"]), n_rows=r["cap"], name="sa%d" % (s + 1))
        m1, m2, m3 = enc.sa_mats[s]
        gbuf = slot.G[s]
        o3 = enc.bn_off[m3.bn_index]
        cnt = geo.counts[s] if train else INF
        # dbeta / dgamma of the pooled layer; the pooled gradient is ReLU-masked in place for its consumers
        plan.call("gad_pool_bwd_stats", slot.dF[s], slot.argmax[s], r["G"], m3.n_out, slot.Z[s][2], m3.n_out,
                  _bn_vec(slot, enc, m3, "scale"), _bn_vec(slot, enc, m3, "shift"), _bn_vec(slot, enc, m3, "mean"),
                  _bn_vec(slot, enc, m3, "istd"), _ptr(slot.bstats, o3, 8), _ptr(slot.bstats, tot + o3, 8), 2 * tot, 1,
                  slot.zmax[s])
        d = layer(s, 2, m3, slot.Z[s][2], cnt, True, pooled=(slot.argmax[s], slot.dF[s], r["grp"]), row_w=_ptr(r["w"]))
        dx(dict(rows_kw, layer=3), d, m3, m2.n_out, epilogue=0, gout=_ptr(gbuf[0]), gout_pitch=m2.n_out,
           **prev_stats(m2, slot.Z[s][1]))
        d = layer(s, 1, m2, slot.Z[s][1], cnt, True, G=gbuf[0], row_w=_ptr(r["w"]))
        dx(dict(rows_kw, layer=2), d, m2, m1.n_out, epilogue=0, gout=_ptr(gbuf[1]), gout_pitch=m1.n_out,
           **prev_stats(m1, slot.Z[s][0]))
        has_dx = s > 0 or (want_daction and action is not None)
        d = layer(s, 0, m1, slot.Z[s][0], cnt, has_dx, G=gbuf[1], row_w=_ptr(r["w"]))
        if s > 0:
            fc = slot.F[s - 1].shape[1]
            if zero_scatter:
                plan.zero(slot.dF[s - 1])
            dx(dict(rows_kw, layer=1), d, m1, fc, epilogue=1, dfeat=_ptr(slot.dF[s - 1]), feat_c=fc, row_pt=_ptr(r["pt"]),
               row_grp=_ptr(r["grp"]), act_c=0, grp_per_sample=1)
        elif has_dx:
            if zero_scatter:
                plan.zero(slot.daction)
            dx(dict(rows_kw, layer=1), d, m1, m1.k_in, epilogue=1, dfeat=None, feat_c=4, row_pt=_ptr(r["pt"]), row_grp=_ptr(r["grp"]),
               daction=_ptr(slot.daction), act_c=6, grp_per_sample=geo.M1)
    for lane in sorted(set(l for _, l, _ in deferred_dw)):
        plan.fork(lane)
    for a, lane, tag in deferred_dw:
        plan.call_struct("gad_gemm_dw", a, side=lane)
        plan.tag_last(tag)
    for lane in sorted(set(dw_lanes) - {0}):
        plan.join(lane)
    return plan


def plan_zero_backward(enc, slot):
    """clear the backward statistics of a slot (call before the consumer head's dX fills fc[1]'s)."""
    plan = Plan()
    plan.zero(slot.bstats)
    return plan
