"""Fused runtime of one agent: owns the engine objects (flat nets, geometry, activation slots, head
slots, static batch buffers) and runs the DDPG / BC update step as a sequence of libgaddpg calls.

Step structure mirrored from the reference: DDPG.update_parameters (core/ddpg.py:146-185) =
critic phase (extract_feature -> target_value -> compute_critic_loss -> critic_optimize) then
actor phase (extract_feature -> policy.sample -> [actor-critic term every policy_update_gap
steps] -> compute_loss -> optimize -> target updates) and BC.update_parameters (core/bc.py:71-87).
Host <-> device traffic per step: one pinned upload of the minibatch, one 32-float download.
"""
import numpy as np
import torch

from . import engine, heads, hip
from .engine import Plan

BATCH_KEYS = ("point_state_batch", "next_point_state_batch", "action_batch", "expert_action_batch", "reward_batch",
              "return_batch", "mask_batch", "time_batch", "goal_batch", "expert_flag_batch", "perturb_flag_batch")


OVERLAP_PASSES = True      # run independent encoder passes of the DDPG step on side streams
import os as _os
# (measured and removed, same box, 287-288 steps/s for the schedule below: HIP-graph replay of the step 240; the actor pass
# started beside t1 272; the next step's value pass under this step's actor backward: no gain; the actor pass after t2, i.e.
# beside the critic backward only: 275; the value pass after t1, beside t2 and the actor pass: 282; both side passes
# swapped: 278 -- profiles/README.md round 2)
ROW_HINTS = _os.environ.get("GAD_ROW_HINTS", "1") == "1"     # grids of the SA1 / SA2 tile launches sized for the expected live rows
EARLY_ZERO = _os.environ.get("GAD_EARLY_ZERO", "1") == "1"   # backward buffers cleared at the end of the forward plans; t2's running update on the value stream
INPUT_SETS = int(_os.environ.get("GAD_INPUT_SETS", "2"))      # 1: uploads + geometry in front of every step (round-1 schedule)
STEP_PLAN = _os.environ.get("GAD_STEP_PLAN", "1") == "1"      # the whole update step as ONE replayed launch list (FusedRuntime._step_plan);
                                                              # 0: enqueued call by call from Python (_ddpg_enqueue: the same launches)


def _dev_f32(x, dev):
    """float32 device copy of a small host / device vector (module attributes such as action_scale live wherever
    GaussianPolicy.to() last put them)"""
    if torch.is_tensor(x):
        return x.detach().to(device=dev, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.asarray(x, dtype=np.float32)).to(dev)


def _fold_stats(s, fused):
    """private copy of a step's result block; the max-abs statistics of a fused optimiser launch arrive spread over 8 slots"""
    s = s.copy()
    if fused:
        s[10], s[11], s[12] = s[32:40].max(), s[40:48].max(), s[48:56].max()
    return s


class PendingStep(object):
    """result block of a step that was enqueued without waiting for it (ddpg_step(sync=False)); wait() blocks until the
    step has run and returns a private copy of the 32 floats"""

    def __init__(self, event, view, fused=False):
        self.event, self._view, self._value, self._fused = event, view, None, fused

    def done(self):
        return self._value is not None or self.event.query()

    def wait(self):
        if self._value is None:
            self.event.synchronize()
            self._value = _fold_stats(self._view, self._fused)
            self._view = None
        return self._value


class FusedRuntime(object):
    def __init__(self, agent, B, NP, device=None):
        dev = self.dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.agent, self.B, self.NP = agent, B, NP
        self.has_critic = agent.has_critic
        fe = agent.state_feature_extractor.module
        self.N = NP - 6 if NP != 1024 else NP
        # one packed view per module, shared with the module-level forward helpers below
        encs = _encoder_nets(fe, dev)
        self.enc, self.venc = encs[False], encs[True]
        self.pol = _head_net(agent.policy, "policy", dev)
        self.pol_t = _head_net(agent.policy_target, "policy", dev)
        if self.has_critic:
            self.cr = _head_net(agent.critic, "critic", dev)
            self.cr_t = _head_net(agent.critic_target, "critic", dev)
        # one gradient buffer per optimiser phase (a data-parallel run all-reduces it in one call); before any plan exists
        # encoder first: its leading SA1 slice is the late bucket of a data-parallel run, [encoder rest | head] the early one
        self.bucket_a = engine.coalesce_grads([self.enc.flat, self.pol.flat])
        self.bucket_c = engine.coalesce_grads([self.venc.flat, self.cr.flat]) if self.has_critic else None
        sa1 = engine.SAConfig(fe.pointnet_nclusters, fe.pointnet_radius, 64)
        sa2 = engine.SAConfig(32, 0.04, 128)
        self.geo = engine.Geometry(B, self.N, sa1, sa2, dev)
        self.slot_p = engine.EncoderSlot(self.geo, self.enc, dev)
        f32 = dict(dtype=torch.float32, device=dev)
        if self.has_critic:
            self.geo_next = engine.Geometry(B, self.N, sa1, sa2, dev)
            self.slot_v = engine.EncoderSlot(self.geo, self.venc, dev)
            self.slot_t = engine.EncoderSlot(self.geo_next, self.enc, dev, with_backward=False)
            self.hs_c = heads.HeadSlot(B, self.cr.width, 9, dev)
            self.hs_ct = heads.HeadSlot(B, self.cr.width, 9, dev)
            self.hs_cpi = heads.HeadSlot(B, self.cr.width, 9, dev)      # Q(s, pi(s)) of the actor phase
            self.hs_pt = heads.HeadSlot(B, self.pol.hidden, self.pol.n_heads, dev)
            self.pi_t = torch.zeros(B, 6, **f32)
            self.a_next = torch.zeros(B, 6, **f32)
            self.noise_u = torch.zeros(B, 6, **f32)
            self.y = torch.zeros(B, **f32)
            self.critic_aux_norm = torch.zeros(B, 7, **f32)
            self.clip_sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
            sel = np.zeros(self.cr.flat.n, dtype=np.uint8)              # polyak map (core/utils.py:757-770)
            for name, p, o in zip(self.cr.flat.names, self.cr.flat.params, self.cr.flat.offsets[:-1]):
                k = 1 if name[:7] in ("linear1", "linear2", "linear3") else (2 if name[:7] in ("linear4", "linear5", "linear6") else 0)
                sel[int(o):int(o) + p.numel()] = k
            self.critic_sel = torch.from_numpy(sel).to(dev)
        self.hs_p = heads.HeadSlot(B, self.pol.hidden, self.pol.n_heads, dev)
        self.pi = torch.zeros(B, 6, **f32)
        self.aux_pred = torch.zeros(B, 7, **f32) if self.pol.n_heads == 13 else self.hs_p.out[:, 6:]
        self.action_scale = _dev_f32(agent.policy.action_scale, dev)
        # (high + low) / 2 of the action space (core/networks.py:329-337); None when the bounds are symmetric (PandaTaskSpace6D)
        ab = _dev_f32(agent.policy.action_bias, dev).reshape(-1)
        self.action_bias = (ab.expand(6).contiguous() if ab.numel() == 1 else ab) if bool((ab != 0).any().item()) else None
        # static batch buffers + pinned staging
        shapes = {"point_state_batch": (B, 4, NP), "next_point_state_batch": (B, 4, NP), "action_batch": (B, 6),
                  "expert_action_batch": (B, 6), "goal_batch": (B, 7)}
        self.dbuf, self.hbuf = {}, {}
        self._hshapes = {}
        for k in BATCH_KEYS + ("time_m1",):
            shp = shapes.get(k, (B,))
            self._hshapes[k] = shp
            self.dbuf[k] = torch.zeros(*shp, **f32)
            self.hbuf[k] = torch.zeros(*shp, dtype=torch.float32).pin_memory()
        # Two input / geometry sets (INPUT_SETS): the uploads and the geometry of step N+1 are enqueued on their own stream
        # into the set that step N is not using, so with a host that runs ahead (sync=False) they execute under step N's
        # backward pass instead of in front of step N+1's critical chain.  Activation scratch is shared (engine.slot_view).
        self._sets = [dict(dbuf=self.dbuf, geo=self.geo, geo_next=getattr(self, "geo_next", None), slot_p=self.slot_p,
                           slot_v=getattr(self, "slot_v", None), slot_t=getattr(self, "slot_t", None))]
        if INPUT_SETS > 1 and self.has_critic:
            geo2, geon2 = engine.Geometry(B, self.N, sa1, sa2, dev), engine.Geometry(B, self.N, sa1, sa2, dev)
            self._sets.append(dict(dbuf={k: torch.zeros_like(v) for k, v in self.dbuf.items()}, geo=geo2, geo_next=geon2,
                                   slot_p=engine.slot_view(self.slot_p, geo2), slot_v=engine.slot_view(self.slot_v, geo2),
                                   slot_t=engine.slot_view(self.slot_t, geon2)))
        for st in self._sets:
            st["rows_pin"] = torch.zeros(4, dtype=torch.int32).pin_memory()    # live rows of [geo SA1, SA2, geo_next SA1, SA2]
            st.update(ev_in=torch.cuda.Event(), ev_gn=torch.cuda.Event(), ev_g=torch.cuda.Event(), ev_up=torch.cuda.Event(), ev_free=None,
                      plans=None)
        self._set = 0
        self._rows_seen = [[], []]            # recent live-row counts of SA1 / SA2 (grid-size hints, engine.Geometry.rows_hint)
        self.scal = torch.zeros(64, **f32)          # [0, 32): the step's result block; [32, 56): 3 x 8 spread max-abs slots
        self._one = torch.ones(1, **f32)
        self._minus_one = -torch.ones(1, **f32)
        # host-side staging, one set per in-flight step (ddpg_step(sync=False) lets the host run ahead of the GPU)
        R = engine.HOST_RING
        self._scal_ring = torch.zeros(R, 64, dtype=torch.float32).pin_memory()
        self._noise_ring = torch.zeros(R, B, 6, dtype=torch.float32).pin_memory()
        self._hbuf_ring = [self.hbuf] + [None] * (R - 1)       # further sets are allocated when a host batch needs them
        self._ev_done = [None] * R
        self._pending = [None] * R
        self._nsteps = 0
        self._slot = 0
        self.scal_host = self._scal_ring[0]
        self.bucketed = False
        self.dp = None                   # parallel.DataParallelContext (set by its attach())
        self.allreduce = None            # callable(list of flat grad tensors)
        # One launch per optimiser phase (gad_optim_jobs: arena -> .grad, Adam, target update, log statistics, BatchNorm
        # counters) instead of ~6 / ~12 small ones; a data-parallel run exchanges gradients between the conversion and the
        # Adam step: the backward plans then convert arena -> .grad themselves and the launch starts from .grad.
        self.fused_optim = self.has_critic and _os.environ.get("GAD_FUSED_OPTIM", "1") == "1"
        # this step's Adam scalars of every network: one pinned block per in-flight step, ONE upload
        self._opt_nets = [self.pol, self.enc] + ([self.venc, self.cr] if self.has_critic else [])
        self.hyper_all = torch.zeros(len(self._opt_nets), 8, **f32)
        self._hyper_ring = torch.zeros(R, len(self._opt_nets), 8, dtype=torch.float32).pin_memory()
        for k, net in enumerate(self._opt_nets):
            net.flat.hyper = self.hyper_all[k]
        self.seg = {}
        for nm, fl in (("pol", self.pol.flat),) + ((("cr", self.cr.flat),) if self.has_critic else ()):
            self.seg[nm] = torch.tensor([0, fl.n], dtype=torch.int32, device=dev)
        self._build_all_plans()
        self.world_size = 1
        self.inv_n = None
        self.resident = False            # True: the static batch buffers were filled on the device
        self._ev = [torch.cuda.Event() for _ in range(5)]
        self._ev_counts = torch.cuda.Event()
        self._ev_in = torch.cuda.Event()
        self._ev_pre = torch.cuda.Event()
        self._ev_run = torch.cuda.Event()
        self.noise_host = self._noise_ring[0]

    # ------------------------------------------------------------------ plans over static buffers
    def set_fused_optim(self, on):
        """switch between one optimiser launch per phase and the separate conversion / Adam / target / statistics launches
        (rebuilds the backward plans: the fused launch takes over their arena -> .grad tail)"""
        on = bool(on) and self.has_critic
        if on != self.fused_optim:
            self.fused_optim = on
            self._build_all_plans()

    def enable_bucketed_reduce(self):
        """data-parallel runs: rebuild the backward plans so that each optimiser phase's gradients leave in two buckets --
        [head | encoder FC + SA3 + SA2] as soon as the SA2 backward is done (99 % of the bytes, all-reduced under the SA1
        backward) and [encoder SA1] at the end -- instead of one exchange after the whole pass (parallel.py)"""
        self.bucketed = True
        self._build_all_plans()

    def _grad_tail(self, plan, head, enc, tag, early):
        """arena (f64, packed) -> flat .grad (f32, master order) at the end of a backward plan; bucketed: only what the
        early hook has not converted yet (the encoder's SA1 parameters, which lead its flat buffer)"""
        if self.fused_optim and self.has_critic and self.dp is None:
            # the optimiser launch converts (gad_optim_jobs); only the critic's gradient is needed before it: clip_grad_norm_
            # (data-parallel runs convert here instead: the exchange sits between the conversion and the optimiser launch)
            if tag == "c":             # ... with the sum of squares of the result in the same launch (self.clip_sumsq is cleared with
                plan.call("gad_grad_from_arena_sumsq", head.flat.gacc, head.flat.m2p, head.flat.n, head.flat.grad, 0, self.clip_sumsq)   # the backward buffers)
            return
        if not (self.bucketed and early):
            plan.call("gad_grad_from_arena", head.flat.gacc, head.flat.m2p, head.flat.n, head.flat.grad, 0)
            plan.call("gad_grad_from_arena", enc.flat.gacc, enc.flat.m2p, enc.flat.n, enc.flat.grad, 0)
            return
        lo, hi = enc.flat.segment("0.0.")                       # base_network[0][0] = SA1 (core/networks.py:65-92)
        assert lo == 0
        plan.call("gad_grad_from_arena", enc.flat.gacc, enc.flat.m2p, hi, enc.flat.grad, 0)

    def _early_hook(self, head, enc, tag):
        if not self.bucketed:
            return None
        lo, hi = enc.flat.segment("0.0.")
        n_rest = enc.flat.n - hi
        bucket = self.bucket_a if tag == "a" else self.bucket_c
        assert bucket.data_ptr() == enc.flat.grad.data_ptr()       # [encoder (SA1 first) | pad | head]: the early bucket is one slice

        def hook(plan, lane):
            plan.call("gad_grad_from_arena", head.flat.gacc, head.flat.m2p, head.flat.n, head.flat.grad, 0, side=lane)
            plan.call("gad_grad_from_arena", enc.flat.gacc, engine._ptr(enc.flat.m2p, hi), n_rest, engine._ptr(enc.flat.grad, hi),
                      0, side=lane)
            plan.fn(lambda: self._reduce_early(tag, [bucket[hi:]]), side=lane)
        return hook

    def _reduce_early(self, tag, tensors):
        if self.dp is not None:
            self.dp.reduce_early(tag, tensors)

    def _bind_set(self, i):
        """make input / geometry set i the current one (self.dbuf, self.geo, self.slot_*, self.plans)"""
        st = self._sets[i]
        self._set = i
        self.dbuf, self.geo, self.slot_p = st["dbuf"], st["geo"], st["slot_p"]
        if self.has_critic:
            self.geo_next, self.slot_v, self.slot_t = st["geo_next"], st["slot_v"], st["slot_t"]
        self.plans = st["plans"]
        return st

    def _build_all_plans(self):
        self._plans_gen = getattr(self, "_plans_gen", 0) + 1          # (invalidates the replayed step lists built over the old plans)
        for i in reversed(range(len(self._sets))):
            self._bind_set(i)
            self._build_plans()
            self._sets[i]["plans"] = self.plans
            self._sets[i].pop("plans_eval", None)

    def _eval_plans(self):
        """the plans of update_parameters(test=True) over the current input / geometry set (built on first use): every encoder pass
        with eval-mode BatchNorm, the backward passes with its eval-mode derivative (engine.plan_encoder_backward train=False)"""
        st = self._sets[self._set]
        if st.get("plans_eval") is None:
            keep = self.plans
            self._build_plans(train=False)
            st["plans_eval"], self.plans = self.plans, keep
        return st["plans_eval"]

    def _build_plans(self, train=True):
        d = self.dbuf
        enc, pol = self.enc, self.pol
        P = self.plans = {}
        P["p_fwd"] = engine.plan_encoder_forward(enc, self.slot_p, action=None, train=train)
        P["p_fwd"].extend(heads.plan_policy_forward(pol, self.hs_p, enc, self.slot_p, d["time_batch"]))
        # the backward pass's buffers are cleared at the END of the forward plan (the actor stream, long before the backward
        # starts) instead of at the head of the backward plan, one launch in front of its critical chain
        zp = [pol.flat.gacc, enc.flat.gacc, self.slot_p.bstats, self.slot_p.dF[0], self.slot_p.dF[1]]
        bw = Plan()
        (P["p_fwd"] if EARLY_ZERO else bw).zero_multi(zp)
        bw.extend(heads.plan_policy_backward(pol, self.hs_p, enc, self.slot_p, d["time_batch"]))
        bw.extend(engine.plan_encoder_backward(enc, self.slot_p, self.hs_p.g_feat, action=None, want_dw=True, dw_lane=2,
                                               zero_scatter=False, early_hook=self._early_hook(pol, enc, "a"), train=train))
        self._grad_tail(bw, pol, enc, "a", True)
        P["p_bwd"] = bw
        if not self.has_critic:
            return
        venc, cr = self.venc, self.cr
        c = engine.plan_encoder_forward(venc, self.slot_v, action=d["action_batch"], train=train)
        c.extend(heads.plan_critic_forward(cr, self.hs_c, venc, self.slot_v, d["time_batch"]))
        zc = [cr.flat.gacc, venc.flat.gacc, self.slot_v.bstats, self.slot_v.dF[0], self.slot_v.dF[1], self.clip_sumsq]
        if EARLY_ZERO:
            c.zero_multi(zc)
        t1 = engine.plan_encoder_forward(enc, self.slot_t, action=None, train=train)
        t1.extend(heads.plan_policy_forward(self.pol_t, self.hs_pt, enc, self.slot_t, d["time_m1"]))
        t1.call("gad_policy_outputs", self.hs_pt.out, self.B, self.pol_t.n_heads, self.action_scale, self.action_bias, self.pi_t, None)
        t2 = engine.plan_encoder_forward(venc, self.slot_t, action=self.a_next, update_running=not OVERLAP_PASSES, train=train)
        P["t2_run"] = engine.plan_running_update(venc, self.slot_t)
        t2.extend(heads.plan_critic_forward(self.cr_t, self.hs_ct, venc, self.slot_t, d["time_m1"]))
        P["c_fwd"], P["t1"], P["t2"] = c, t1, t2
        cb = Plan()                                 # (its buffers were cleared at the end of the value pass)
        if not EARLY_ZERO:
            cb.zero_multi(zc)
        cb.extend(heads.plan_critic_backward(cr, self.hs_c, venc, self.slot_v, d["time_batch"]))
        cb.extend(engine.plan_encoder_backward(venc, self.slot_v, self.hs_c.g_feat, action=d["action_batch"], want_dw=True,
                                               zero_scatter=False, early_hook=self._early_hook(cr, venc, "c"), train=train))
        self._grad_tail(cb, cr, venc, "c", True)
        P["c_bwd"] = cb
        # actor-critic term: Q(s, pi(s)) through the freshly updated critic, gradient back to pi
        v = engine.plan_encoder_forward(venc, self.slot_v, action=self.pi, train=train)
        v.extend(heads.plan_critic_forward(cr, self.hs_cpi, venc, self.slot_v, d["time_batch"]))
        P["v_fwd"] = v
        vb = Plan()
        vb.zero_multi([cr.flat.gacc, self.slot_v.bstats, self.slot_v.dF[0], self.slot_v.dF[1], self.slot_v.daction])
        vb.extend(heads.plan_critic_backward(cr, self.hs_cpi, venc, self.slot_v, d["time_batch"]))
        # the reference's backward also leaves the actor-loss gradient in critic.grad (logged as critic_grad): converted on the
        # weight-gradient lane, behind the head's dW GEMMs, off the dX chain (v_bwd's closing join covers it)
        vb.call("gad_grad_from_arena", cr.flat.gacc, cr.flat.m2p, cr.flat.n, cr.flat.grad, 1, side=1 if heads.CONCURRENT_DW_HEADS() else 0)
        vb.extend(engine.plan_encoder_backward(venc, self.slot_v, self.hs_cpi.g_feat, action=self.pi, want_dw=False,
                                               want_daction=True, zero_scatter=False, train=train))
        if heads.CONCURRENT_DW_HEADS():
            vb.join(1)                   # (no encoder weight gradients in this pass: nothing else joins the head's lane)
        P["v_bwd"] = vb

    # ------------------------------------------------------------------ host -> device
    def load_device_batch(self, dbatch, keys=None):
        """device-resident minibatch (dict of CUDA float32 tensors with the BATCH_KEYS layout): ONE launch copies every key into
        the static buffers (gad_copy_buffers; "time minus one" formed on the way), no host traffic.  `keys`: only these
        (staged upload)."""
        import ctypes as C
        ks = [k for k in (BATCH_KEYS if keys is None else keys) if k in dbatch]
        segs = []
        for k in ks:
            src, dst = dbatch[k], self.dbuf[k]
            if src.dtype != torch.float32 or not src.is_contiguous() or src.numel() != dst.numel() or src.device != dst.device:
                dst.copy_(src, non_blocking=True)            # (a layout gad_copy_buffers does not take: torch converts)
                src = dst
            else:
                segs.append((dst.data_ptr(), src.data_ptr(), 4 * dst.numel(), 0.0))
            if k == "time_batch":
                segs.append((self.dbuf["time_m1"].data_ptr(), src.data_ptr(), 4 * dst.numel(), -1.0))
        for i in range(0, len(segs), hip.COPY_MAX_SEGS):
            grp = segs[i:i + hip.COPY_MAX_SEGS]
            arr = (hip.CopySeg * len(grp))()
            for a, (dp_, sp_, nb, add) in zip(arr, grp):
                a.dst, a.src, a.bytes, a.add = dp_, sp_, nb, add
            hip.check(hip.lib().gad_copy_buffers(arr, len(grp), hip.stream()), "gad_copy_buffers")

    def upload(self, batch, keys=None):
        """minibatch -> the static device buffers.  `keys` restricts the call to some of BATCH_KEYS: ddpg_step uploads
        in stages so that the first kernels of the critical chain are enqueued before the remaining copies."""
        if batch is None:
            return
        if "replay_gather" in batch:             # DeviceReplay.sample_lazy(): one gather launch into the static buffers
            if int(batch["idx"].shape[0]) != self.B:
                raise RuntimeError("batch size changed: runtime was built for B=%d" % self.B)
            return batch["replay_gather"].gather_into(batch, self.dbuf)
        if torch.is_tensor(batch["point_state_batch"]):
            return self.load_device_batch(batch, keys)
        B = self.B
        for k in (BATCH_KEYS if keys is None else keys):
            if k not in batch or (not self.has_critic and k in ("next_point_state_batch",)):
                continue
            a = np.asarray(batch[k])
            if a.shape[0] != B:
                raise RuntimeError("batch size changed: runtime was built for B=%d, got %d" % (B, a.shape[0]))
            h = self.hbuf[k]
            np.copyto(h.numpy(), a.reshape(h.shape), casting="same_kind")
            self.dbuf[k].copy_(h, non_blocking=True)
            if k == "time_batch":
                h = self.hbuf["time_m1"]
                np.subtract(self.hbuf["time_batch"].numpy(), 1.0, out=h.numpy())
                self.dbuf["time_m1"].copy_(h, non_blocking=True)

    def _optim_job(self, flat, adam=True, arena=True, clip=None, target=None, sel=None, absmax_p=None, absmax_grad=None,
                   counter=None):
        j = hip.OptimJob()
        j.n = flat.n
        j.p, j.grad, j.exp_avg, j.exp_avg_sq = (hip.ptr(t) for t in (flat.master, flat.grad, flat.exp_avg, flat.exp_avg_sq))
        j.active, j.m2p, j.packed = hip.ptr(flat.active), hip.ptr(flat.m2p), hip.ptr(flat.packed)
        if arena:
            j.gacc = hip.ptr(flat.gacc)
        if adam:
            j.hyper = hip.ptr(flat.hyper)
        if clip is not None:
            j.clip_sumsq, j.clip_max = hip.ptr(clip), float(self.agent.clip_grad)
        if target is not None:
            j.target, j.target_m2p, j.target_packed = hip.ptr(target.master), hip.ptr(target.m2p), hip.ptr(target.packed)
            j.target_sel = hip.ptr(sel)
            j.tau = float(self.agent.tau)
        j.absmax_p, j.absmax_grad = hip.ptr(absmax_p), hip.ptr(absmax_grad)
        if counter is not None:
            j.counter, j.counter_n = hip.ptr(counter), int(counter.numel())
        return j

    def _optim_jobs(self):
        """the gad_optim_job arrays of the optimiser phases (cached; rebuilt when the gradient buffers move)"""
        ag = self.agent
        jobs = getattr(self, "_optim_jobs_cache", None)
        ar = self.dp is None          # data-parallel: the .grad buffers hold the converted, all-reduced gradients already
        if jobs is None or jobs["key"] != (id(self.pol.flat.grad), id(self.enc.flat.grad), bool(ag.train_feature), ar):
            sc = self.scal
            arr = lambda js: (hip.OptimJob * len(js))(*js)
            self._jobs_gen = getattr(self, "_jobs_gen", 0) + 1
            jobs = self._optim_jobs_cache = {
                "key": (id(self.pol.flat.grad), id(self.enc.flat.grad), bool(ag.train_feature), ar),
                # value encoder: arena -> grad + Adam; critic: Adam with the clip (its .grad is converted already) + target
                # update from the updated parameters (nothing reads critic_target before the next step) + max |parameter|
                "c": arr([self._optim_job(self.venc.flat, arena=ar),
                          self._optim_job(self.cr.flat, arena=False, clip=self.clip_sumsq, target=self.cr_t.flat, sel=self.critic_sel,
                                          absmax_p=engine._ptr(sc, 48))]),
                "a": arr([self._optim_job(self.pol.flat, arena=ar, target=self.pol_t.flat, absmax_p=engine._ptr(sc, 32)),
                          self._optim_job(self.enc.flat, arena=ar, adam=bool(ag.train_feature), counter=self.enc.batches_tracked)]),
                "end": arr([self._optim_job(self.cr.flat, adam=False, arena=False, absmax_grad=engine._ptr(sc, 40),
                                            counter=self.venc.batches_tracked)])}
            # the end-of-step bookkeeping (max |critic.grad| as the reference logs it after the actor backward, the value encoder's
            # BatchNorm counters) folded into a launch that runs anyway -- one dependent launch fewer at the step boundary:
            #   policy steps: a third job of the actor phase's launch (critic.grad is final once the actor-critic backward added to it);
            #   other steps : inside the critic job of the critic phase's launch (nothing touches critic.grad after it; the Adam
            #                 kernel's max |grad| is taken after its in-place clip scaling, as torch's clip_grad_norm_ leaves it)
            jobs["a+end"] = arr([jobs["a"][0], jobs["a"][1], jobs["end"][0]])
            ce = self._optim_job(self.cr.flat, arena=False, clip=self.clip_sumsq, target=self.cr_t.flat, sel=self.critic_sel,
                                 absmax_p=engine._ptr(sc, 48), absmax_grad=engine._ptr(sc, 40), counter=self.venc.batches_tracked)
            jobs["c+end"] = arr([jobs["c"][0], ce])
        return jobs

    def _optim_select(self, which, policy_step):
        """the job array of an optimiser phase with this step's scalars written into it (None: the phase has no launch of its own)"""
        ag = self.agent
        jobs = self._optim_jobs()
        fold = self.dp is None                # (data-parallel runs keep the separate launch: their phases end with exchanges)
        ev = getattr(self, "_eval", False)    # eval-mode BatchNorm (test=True): num_batches_tracked stays
        if which == "end":
            if fold:
                return None                   # (folded: see above)
            js = jobs["end"]
            js[0].counter_add = 0 if ev else (3 if policy_step else 2)
        elif which == "c":
            js = jobs["c+end"] if (fold and not policy_step) else jobs["c"]
            js[1].hard_enable = int(ag.update_step % ag.target_update_interval == 0)
            js[1].tau = float(ag.tau)
            js[1].counter_add = 0 if ev else 2
        else:
            js = jobs["a+end"] if (fold and policy_step) else jobs["a"]
            js[0].tau = float(ag.tau)
            js[1].counter_add = 0 if ev else 2
            if len(js) == 3:
                js[2].counter_add = 0 if ev else 3
        return js

    def _optim_phase(self, which, policy_step):
        """gad_optim_jobs for the critic phase ("c"), the actor phase ("a") or the end of the step ("end": what has to
        wait for both phases -- max |critic.grad| as the reference logs it after the actor backward, the value encoder's
        BatchNorm counters)"""
        ag = self.agent
        js = self._optim_select(which, policy_step)
        if js is None:
            return
        hip.check(hip.lib().gad_optim_jobs(js, len(js), hip.stream()), "gad_optim_jobs")
        if which == "c":                     # the encoders' split-bf16 weight mirrors follow their packed weights
            self.venc.flat.refresh_split()
        elif which == "a" and ag.train_feature:
            self.enc.flat.refresh_split()

    def _adam_host(self, flat, optim):
        g = optim.param_groups[0]
        flat.set_adam_hyper(g["lr"], g["betas"], g["eps"], g["weight_decay"], upload=False)

    def _adam(self, flat, optim, clip=None):
        self._adam_host(flat, optim)
        flat.hyper.copy_(flat.hyper_host, non_blocking=True)
        hip.call("gad_adam_step", flat.master, flat.grad, flat.exp_avg, flat.exp_avg_sq, flat.active, flat.m2p,
                 flat.packed, flat.n, flat.hyper, clip, float(self.agent.clip_grad) if clip is not None else 0.0)
        flat.refresh_split()

    def _reduce(self, flats, tag=None):
        if self.allreduce is None:
            return
        head, enc = flats
        if self.bucketed and tag is not None:                 # the early bucket is in flight: SA1's slice + wait for both
            lo, hi = enc.segment("0.0.")
            self.dp.reduce_finish(tag, [enc.grad[:hi]])
            return
        b = getattr(enc, "_grad_bucket", None)
        whole = b is not None and b[0] == (id(enc), id(head))
        self.allreduce([b[1]] if whole else [f.grad for f in flats])

    # ------------------------------------------------------------------ the update steps
    def _begin_step(self, slot=None, alternate=False):
        """pick the host staging set of this step (pinned result / noise / Adam-scalar / input blocks).  A set is reused
        every engine.HOST_RING steps: wait for the step that used it last -- that is what bounds the host's run-ahead."""
        R = engine.HOST_RING
        if slot is None:
            slot = self._nsteps % R
        self._nsteps += 1
        pend = self._pending[slot]
        if pend is not None:
            pend.wait()                      # its numbers leave the pinned block before the block is reused
            self._pending[slot] = None
        elif self._ev_done[slot] is not None:
            self._ev_done[slot].synchronize()
        self._slot = slot
        self._bind_set((self._set + 1) % len(self._sets) if (alternate and len(self._sets) > 1) else 0)
        if ROW_HINTS and self.has_critic:
            self._update_row_hints()
        self.scal_host = self._scal_ring[slot]
        self.noise_host = self._noise_ring[slot]
        if self._hbuf_ring[slot] is None:
            self._hbuf_ring[slot] = {k: torch.zeros(*shp, dtype=torch.float32).pin_memory() for k, shp in self._hshapes.items()}
        self.hbuf = self._hbuf_ring[slot]
        for k, net in enumerate(self._opt_nets):
            net.flat.hyper_host = self._hyper_ring[slot][k]
        return slot

    def _update_row_hints(self):
        """grid-size hints for the tile launches over de-duplicated rows: 1.25 x the largest live-row count among the last
        minibatches (whatever has arrived in the pinned counters: values of earlier steps, possibly of one still in flight --
        a hint only, the kernels' grid-stride loops are bounded by the device-side count)"""
        for st in self._sets:
            v = st["rows_pin"].numpy()
            for stage in (0, 1):
                for x in (int(v[stage]), int(v[2 + stage])):
                    if x > 0:
                        seen = self._rows_seen[stage]
                        seen.append(x)
                        del seen[:-32]
        for stage in (0, 1):
            if self._rows_seen[stage]:
                h = int(1.25 * max(self._rows_seen[stage])) + 256
                for st in self._sets:
                    st["geo"].rows_hint[stage] = h
                    if st["geo_next"] is not None:
                        st["geo_next"].rows_hint[stage] = h

    def _end_step(self, slot, sync):
        ev = self._ev_done[slot]
        if ev is None:
            ev = self._ev_done[slot] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._sets[self._set]["ev_free"] = ev        # every stream of the step has been joined into this one by now
        fused = self.fused_optim and self.has_critic
        if sync:
            ev.synchronize()
            return _fold_stats(self.scal_host.numpy(), fused)
        pend = self._pending[slot] = PendingStep(ev, self.scal_host.numpy(), fused)
        return pend

    def flush(self):
        """wait for every step enqueued so far"""
        for p in self._pending:
            if p is not None:
                p.wait()
        torch.cuda.current_stream().synchronize()

    def ddpg_step(self, batch, noise_u=None, sync=True, test=False):
        """one DDPG / TD3 update: eager multi-stream enqueue.
        sync=False: return a PendingStep right after the enqueue.  The host then prepares and enqueues the next step while
        this one runs (up to engine.HOST_RING - 1 steps ahead): the ~2 ms of launch calls, the uploads and the geometry of
        step N+1 no longer sit in front of its critical chain, they are queued behind step N on the GPU."""
        ag = self.agent
        policy_step = ag.update_step % ag.policy_update_gap == 0
        slot = self._begin_step(alternate=True)
        self._eval = bool(test)
        replay = (not test) and STEP_PLAN and OVERLAP_PASSES and self.fused_optim and self.has_critic
        if not replay:
            self._sets[self._set]["prefetched"] = None        # (the call-by-call path stages its inputs itself)
        if test:
            # test=True (reference core/agent.py:276-280): the same update with eval-mode BatchNorm in every pass -- its own plans,
            # enqueued call by call (no shipped configuration trains this way: not a replayed list, not a fast path)
            keep = self.plans
            self.plans = self._eval_plans()
            try:
                self._ddpg_enqueue(batch, noise_u, policy_step)
            finally:
                self.plans, self._eval = keep, False
        elif replay:
            self._ddpg_replay(batch, noise_u, policy_step)
        else:
            self._ddpg_enqueue(batch, noise_u, policy_step)
        nxt, self._next_batch = getattr(self, "_next_batch", None), None
        if nxt is not None and replay:
            pend = self._end_step(slot, sync=False)
            self._prefetch_now(nxt)               # (after the step's own launches, before the host waits for its result)
            if not sync:
                return pend
            v = pend.wait()
            self._pending[slot] = None
            return v
        return self._end_step(slot, sync)

    # ------------------------------------------------------------------ the step as ONE replayed launch list
    def _step_key(self):
        ag = self.agent
        self._optim_jobs()
        return (self._jobs_gen, self._plans_gen, id(self.dp), id(self.allreduce), None if self.inv_n is None else self.inv_n.data_ptr(),
                bool(ag.train_feature), float(ag.gamma), bool(ag.critic_aux), bool(ag.policy_aux), float(ag.clip_grad),
                self.bucketed, ROW_HINTS, EARLY_ZERO)

    def _step_plan(self, set_index, policy_step):
        """The whole update step over input / geometry set `set_index` as one engine.Plan (what _ddpg_enqueue issues call by
        call, in the same order on the same logical streams): replayed by gad_plan_run in one foreign call -- or a few, when a
        data-parallel run has host callbacks (collectives) inside the step.  What changes from step to step is patched into
        the list before the run (_ddpg_replay): the noise level, the mix ratio, the pinned ring slot of the Adam scalars and
        of the result block; the optimiser scalars live in the job arrays the list points to."""
        key = self._step_key()
        cache = self.__dict__.setdefault("_step_plans", {})
        if cache.get("key") != key:
            cache.clear()
            cache["key"] = key
            cache["refs"] = (self.dp, self.allreduce, self.inv_n)      # (alive while their ids are part of the key)
        ent = cache.get((set_index, policy_step))
        if ent is not None:
            return ent
        bound = self._set
        st = self._bind_set(set_index)
        try:
            ent = self._build_step_plan(st, policy_step)
        finally:
            self._bind_set(bound)
        cache[(set_index, policy_step)] = ent
        return ent

    def _build_step_plan(self, st, policy_step):
        ag, d, P = self.agent, self.dbuf, self.plans
        B = self.B
        EH = engine.EventHolder
        if "ev_up_h" not in st:
            st["ev_up_h"] = EH()
            st["ev_up"] = st["ev_up_h"].event
            st["geo_plan"] = self.geo.plan(d["point_state_batch"])
            st["geo_next_plan"] = self.geo_next.plan(d["next_point_state_batch"])
        H = {}                                  # items patched per step
        MAIN, S1, S2, SC, PRE, PRE2 = 0, 1, 2, 3, 20, 21
        # ---- prefetch lanes: [geometry of the next state] and [geometry of the current state]; the uploads were enqueued on
        # these streams by the prelude (_stage_inputs).  This part is a list of its own, ONE per input set (both kinds of step
        # start from its events): prefetch_inputs() replays it for the NEXT step's minibatch while the current step runs
        cache = self.__dict__["_step_plans"]
        pre = cache.get(("pre", self._set))
        if pre is None:
            ev_in, ev_gn, ev_g = EH(), EH(), EH()
            M = Plan()
            M.keep.append((ev_in, ev_gn, ev_g, st["ev_up_h"]))
            M.record(ev_in, on=PRE)
            M.extend(st["geo_next_plan"], on=PRE)
            M.record(ev_gn, on=PRE)
            if ROW_HINTS:
                M.memcpy(st["rows_pin"].data_ptr() + 8, self.geo_next.rows_n.data_ptr(), 8, on=PRE)
            M.wait_event(ev_in, on=PRE2)
            M.record(st["ev_up_h"], on=PRE2)                    # every input of the step has left the caller's buffers
            M.extend(st["geo_plan"], on=PRE2)
            M.record(ev_g, on=PRE2)
            if ROW_HINTS:
                M.memcpy(st["rows_pin"].data_ptr(), self.geo.rows_n.data_ptr(), 8, on=PRE2)
            pre = cache[("pre", self._set)] = dict(plan=M, ev_gn=ev_gn, ev_g=ev_g)
        Mpre, ev_gn, ev_g = pre["plan"], pre["ev_gn"], pre["ev_g"]
        M = Mstep = Plan()
        ev0, ev1, ev2, ev3, ev4, ev_run, ev_counts = (EH() for _ in range(7))
        M.keep.append((ev0, ev1, ev2, ev3, ev4, ev_run, ev_counts, pre))
        M = Mstep
        # ---- critic phase (the comments of _ddpg_enqueue apply line by line)
        M.record(ev0, on=MAIN)
        M.wait_event(ev_gn, on=MAIN)
        M.extend(P["t1"], on=MAIN)
        M.wait_event(ev0, on=SC)
        M.wait_event(ev_g, on=SC)
        M.zero(self.scal, on=SC)
        H["hyper"] = M.memcpy(self.hyper_all.data_ptr(), self._hyper_ring[0].data_ptr(), 4 * self.hyper_all.numel(), on=SC)
        if self.dp is not None:
            M.fn(lambda: self.dp.set_counts(self._cur_batch), on=SC)
        M.record(ev_counts, on=SC)
        M.wait_event(ev_counts, on=MAIN)
        M.wait_event(ev0, on=S1)
        M.wait_event(ev_g, on=S1)
        M.extend(P["c_fwd"], on=S1)
        M.record(ev2, on=MAIN)
        H["noise"] = M.call("gad_target_noise", self.pi_t, self.noise_u, B, 0.0, 0, self.a_next, on=MAIN)
        M.extend(P["t2"], on=MAIN)

        def actor_tail(g_pi, on):
            H["actor_loss"] = M.call("gad_actor_loss", self.hs_p.out, self.pi, d["expert_action_batch"], d["expert_flag_batch"],
                                     d["return_batch"], d["goal_batch"], B, self.pol.n_heads, 0.0, int(bool(ag.policy_aux)),
                                     self.action_scale, g_pi, self.inv_n_actor(), self.hs_p.g_out, engine._ptr(self.scal, 4), on=on)
            M.extend(P["p_bwd"], on=on)
            if self.allreduce is not None:
                M.fn(lambda: self._reduce([self.pol.flat, self.enc.flat], "a"), on=on)
            optim("a", on)

        def optim(which, on):
            js = self._optim_select(which, policy_step)
            if js is None:
                return
            M.call("gad_optim_jobs", js, len(js), on=on)
            fl = self.venc.flat if which == "c" else (self.enc.flat if (which == "a" and ag.train_feature) else None)
            if fl is not None and fl.split is not None:
                M.call("gad_split_weights", fl.packed, fl._split_layers, len(fl._split_layers), fl.split, on=on)

        M.wait_event(ev2, on=S2)
        M.wait_event(ev0, on=S2)
        M.wait_event(ev_g, on=S2)
        M.extend(P["p_fwd"], on=S2)
        nh = self.pol.n_heads
        M.call("gad_policy_outputs", self.hs_p.out, B, nh, self.action_scale, self.action_bias, self.pi, self.aux_pred if nh == 13 else None,
               on=S2)
        if not policy_step:
            actor_tail(None, S2)
        M.record(ev1, on=S1)
        M.wait_event(ev1, on=MAIN)
        M.record(ev4, on=MAIN)
        M.wait_event(ev4, on=S1)
        lane_run = S1 if EARLY_ZERO else MAIN
        M.extend(P["t2_run"], on=lane_run)
        M.record(ev_run, on=lane_run)
        M.call("gad_critic_loss", self.hs_c.out, self.hs_ct.out, d["reward_batch"], d["mask_batch"], d["perturb_flag_batch"],
               d["return_batch"], d["goal_batch"], B, float(ag.gamma), int(bool(ag.critic_aux)), self.inv_n_critic(), self.y,
               self.critic_aux_norm, self.hs_c.g_out, engine._ptr(self.scal, 0), on=MAIN)
        M.extend(P["c_bwd"], on=MAIN)
        M.wait_event(ev_run, on=MAIN)
        if self.allreduce is not None:
            M.fn(lambda: self._reduce([self.cr.flat, self.venc.flat], "c"), on=MAIN)
        if self.dp is not None:
            M.call("gad_sumsq", self.cr.flat.grad, self.cr.flat.n, self.clip_sumsq, on=MAIN)
        optim("c", MAIN)
        # ---- actor phase
        M.record(ev3, on=S2)
        M.wait_event(ev3, on=MAIN)
        if policy_step:
            M.extend(P["v_fwd"], on=MAIN)
            H["ac_loss"] = M.call("gad_actor_critic_loss", self.hs_cpi.out, d["expert_flag_batch"], d["return_batch"], B, 0.0,
                                  self.inv_n_actor_critic(), self.hs_cpi.g_out, engine._ptr(self.scal, 8), on=MAIN)
            M.extend(P["v_bwd"], on=MAIN)
            actor_tail(self.slot_v.daction, MAIN)
        optim("end", MAIN)
        if self.dp is not None:
            M.fn(lambda: self.dp.reduce_scalars(self.scal), on=MAIN)
        H["download"] = M.memcpy(self._scal_ring[0].data_ptr(), self.scal.data_ptr(), 4 * self.scal.numel(), on=MAIN)
        return dict(plan=M, pre=Mpre, items=H, last={})

    def _stage_inputs(self, batch, st, pre):
        """the inputs of a step into set `st` + its prefetch list `pre` (geometry of both cloud sets): stream waits on events
        owned by other steps / producers, the uploads (or the one gather / copy launch of a device-resident minibatch), then
        the replayed prefetch-lane launches.  The set must be the bound one (self.dbuf / self.geo*)."""
        main = torch.cuda.current_stream()
        spre, spre2 = engine.side_stream(which=20), engine.side_stream(which=21)
        ready = batch.get("ready_event") if batch is not None else None
        if ready is None and batch is not None and ("replay_gather" in batch or (
                torch.is_tensor(batch["point_state_batch"]) and batch["point_state_batch"].is_cuda)):
            self._ev_pre.record(main)               # device tensors of unknown origin: after everything enqueued so far
            ready = self._ev_pre
        for s_ in ((spre,) if spre2 is spre else (spre, spre2)):
            if st["ev_free"] is not None:
                s_.wait_event(st["ev_free"])
            if ready is not None:                   # (a producer's event says when its device tensors are complete)
                s_.wait_event(ready)
        whole = batch is not None and "replay_gather" in batch          # one gather launch fills every buffer
        first = ("next_point_state_batch", "time_batch")
        with torch.cuda.stream(spre):
            self.upload(batch, None if whole else first)
        if not whole:
            with torch.cuda.stream(spre2):
                self.upload(batch, tuple(k for k in BATCH_KEYS if k not in first))
        if isinstance(batch, dict) and "uploaded_event" in batch:   # asked for by the producer (PrefetchSampler): it may
            batch["uploaded_event"] = st["ev_up"]                   # reuse its staging buffers after this event
        pre.run()

    def prefetch_inputs(self, batch):
        """Stage the NEXT step's minibatch now: its upload (one gather / copy launch) and the geometry of both cloud sets go onto
        the prefetch lanes, into the input set the current step is not using, and run beside whatever is in flight.  The next
        ddpg_step(batch) -- the SAME batch object -- then starts from there.  This is what lets the reference-shaped loop
        (sample, update, read the losses, every iteration: core/train_test_offline.py:117-126) keep the GPU busy across its
        host synchronisation: core.train_test_offline.train_off_policy samples one minibatch ahead and calls this.
        Device-resident minibatches only (DeviceReplay.sample_lazy, CUDA tensors); -> False when nothing was staged."""
        if not (STEP_PLAN and OVERLAP_PASSES and self.fused_optim and self.has_critic) or engine.SERIAL or batch is None:
            return False
        dev_batch = "replay_gather" in batch or (torch.is_tensor(batch.get("point_state_batch")) and batch["point_state_batch"].is_cuda)
        if not dev_batch or len(self._sets) < 2:
            return False
        # DEFERRED to the end of the next ddpg_step's enqueue: the prefetch lanes share a hardware queue with the value pass and the
        # critic's weight-gradient lane (engine._PHYS), so launches enqueued NOW would sit in front of that step's value pass
        # (measured: 365 -> 350 steps/s); enqueued behind the step they run beside its tail, which is where the run-ahead loop has them
        self._next_batch = batch
        return True

    def _prefetch_now(self, batch):
        if self._sets[(self._set + 1) % len(self._sets)].get("prefetched") is not None:
            return False
        bound = self._set
        nxt = (self._set + 1) % len(self._sets)
        ent = self._step_plan(nxt, True)                     # (the prefetch list is the same for both step kinds)
        st = self._bind_set(nxt)
        try:
            self._stage_inputs(batch, st, ent["pre"])
            st["prefetched"] = batch
        finally:
            self._bind_set(bound)
        return True

    def _ddpg_replay(self, batch, noise_u, policy_step):
        """enqueue one update step through its replayed launch list (_step_plan).  Eager host work that remains: the staging of
        the inputs (unless prefetch_inputs did it), the TD3 noise draw from torch's generator, this step's scalars."""
        ag = self.agent
        st = self._sets[self._set]
        ent = self._step_plan(self._set, policy_step)
        M, H, last = ent["plan"], ent["items"], ent["last"]
        self._cur_batch = batch
        main = torch.cuda.current_stream()
        sc = engine.side_stream(which=3)
        engine.apply_lane_priorities(main)
        staged, st["prefetched"] = st.get("prefetched"), None
        if staged is None or staged is not batch:
            self._stage_inputs(batch, st, ent["pre"])
        elif isinstance(batch, dict) and "uploaded_event" in batch:
            batch["uploaded_event"] = st["ev_up"]
        # the TD3 noise: torch's device generator (or the injected draw), ordered after the previous step's reader
        self._ev[0].record(main)
        sc.wait_event(self._ev[0])
        normal_noise = getattr(ag, "noise_type", "uniform") != "uniform"     # core/utils.py:568-569
        with torch.cuda.stream(sc):
            if noise_u is None:
                if normal_noise:
                    self.noise_u.normal_()                              # torch.randn_like (core/utils.py:573)
                else:
                    self.noise_u.uniform_(0.0, 1.0)                     # torch.rand_like (core/utils.py:575)
            else:
                self.noise_u.copy_(torch.as_tensor(np.asarray(noise_u, dtype=np.float32)), non_blocking=True)
        # this step's scalars
        self._adam_host(self.venc.flat, ag.state_feat_val_encoder_optim)
        self._adam_host(self.cr.flat, ag.critic_optim)
        self._adam_host(self.pol.flat, ag.policy_optim)
        if ag.train_feature:
            self._adam_host(self.enc.flat, ag.state_feat_encoder_optim)
        for which in ("c", "a", "end"):
            self._optim_select(which, policy_step)
        idx = sum(1 for m in ag.mix_milestones if ag.update_step > m)
        level = float(ag.action_noise * ag.noise_ratio_list[min(len(ag.noise_ratio_list) - 1, idx)])
        ratio = float(ag.mix_policy_ratio)
        want = {("noise", 3): level, ("noise", 4): int(normal_noise), ("actor_loss", 8): 1.0 - ratio, ("ac_loss", 4): ratio,
                ("hyper", 1): self._hyper_ring[self._slot].data_ptr(), ("download", 0): self.scal_host.data_ptr()}
        for (name, index), v in want.items():
            if name in H and last.get((name, index)) != v:
                Plan.patch(H[name], index, v)
                last[(name, index)] = v
        M.run()
        self._cur_batch = None

    def _ddpg_enqueue(self, batch, noise_u, policy_step):
        """enqueue one update step on the current stream + the side streams (no host synchronisation inside); ends with the
        32-float result block on its way to the pinned host buffer"""
        ag, d, P = self.agent, self.dbuf, self.plans
        B = self.B
        ratio = float(ag.mix_policy_ratio)
        main = torch.cuda.current_stream()
        # uploads + geometry of both cloud sets on the prefetch stream, into this step's input / geometry set: ordered only
        # after the last step that used the set, i.e. they run beside the previous step when the host is ahead of the GPU
        st = self._sets[self._set]
        inline = (not OVERLAP_PASSES) or engine.SERIAL
        spre = main if inline else engine.side_stream(which=20)
        spre2 = main if inline else engine.side_stream(which=21)
        if not inline:
            ready = batch.get("ready_event") if batch is not None else None
            if ready is None and batch is not None and ("replay_gather" in batch or (
                    torch.is_tensor(batch["point_state_batch"]) and batch["point_state_batch"].is_cuda)):
                self._ev_pre.record(main)               # device tensors of unknown origin: after everything enqueued so far
                ready = self._ev_pre
            for s_ in (spre, spre2):
                if st["ev_free"] is not None:
                    s_.wait_event(st["ev_free"])
                if ready is not None:                   # (a producer's event says when its device tensors are complete)
                    s_.wait_event(ready)
        # two prefetch streams: [inputs of the target chain -> geometry of the next state] and [the rest -> geometry of the
        # current state] run side by side (each geometry is a chain of small latency-bound kernels)
        whole = batch is not None and "replay_gather" in batch          # one gather launch fills every buffer
        first = ("next_point_state_batch", "time_batch")
        with torch.cuda.stream(spre):
            self.upload(batch, None if whole else first)
            st["ev_in"].record(spre)
            self.geo_next.run(d["next_point_state_batch"])      # the target chain (the critical path) needs this one first
            st["ev_gn"].record(spre)
            if ROW_HINTS:
                st["rows_pin"][2:3].copy_(self.geo_next.rows[0]["n"], non_blocking=True)
                st["rows_pin"][3:4].copy_(self.geo_next.rows[1]["n"], non_blocking=True)
        if whole and not inline:
            spre2.wait_event(st["ev_in"])
        with torch.cuda.stream(spre2):
            if not whole:
                self.upload(batch, tuple(k for k in BATCH_KEYS if k not in first))
                if not inline:
                    spre2.wait_event(st["ev_in"])
            st["ev_up"].record(spre2)                           # every input of the step has left the caller's buffers
            self.geo.run(d["point_state_batch"])
            st["ev_g"].record(spre2)                            # ev_g: ALL inputs are in + the geometry of the current state
            if ROW_HINTS:
                st["rows_pin"][0:1].copy_(self.geo.rows[0]["n"], non_blocking=True)
                st["rows_pin"][1:2].copy_(self.geo.rows[1]["n"], non_blocking=True)
        if isinstance(batch, dict) and "uploaded_event" in batch:   # asked for by the producer (PrefetchSampler): it may
            batch["uploaded_event"] = st["ev_up"]                   # reuse its staging buffers after this event
        idx = sum(1 for m in ag.mix_milestones if ag.update_step > m)
        level = ag.action_noise * ag.noise_ratio_list[min(len(ag.noise_ratio_list) - 1, idx)]
        normal_noise = getattr(ag, "noise_type", "uniform") != "uniform"     # core/utils.py:568-569

        def small_inits():
            # not needed by the geometry / first passes: enqueued after them so that the GPU starts the step earlier
            if noise_u is None:
                if normal_noise:
                    self.noise_u.normal_()                              # torch.randn_like (core/utils.py:573)
                else:
                    self.noise_u.uniform_(0.0, 1.0)                     # torch.rand_like (core/utils.py:575)
            else:
                self.noise_u.copy_(torch.as_tensor(np.asarray(noise_u, dtype=np.float32)), non_blocking=True)
            self.scal.zero_()
            if self.fused_optim:                    # this step's Adam scalars of all four networks: one upload
                self._adam_host(self.venc.flat, ag.state_feat_val_encoder_optim)
                self._adam_host(self.cr.flat, ag.critic_optim)
                self._adam_host(self.pol.flat, ag.policy_optim)
                if ag.train_feature:
                    self._adam_host(self.enc.flat, ag.state_feat_encoder_optim)
                self.hyper_all.copy_(self._hyper_ring[self._slot], non_blocking=True)
        # ---- critic phase.  The TD target chain (encoder(next) -> target policy -> value encoder(next, a') -> target
        # critic) and the value pass on the current state are independent: the latter runs on a second stream.
        # Both go through value_encoder's BatchNorms; the reference runs the current-state pass first
        # (core/ddpg.py:145-152, then target_value() inside compute_critic_loss), so the target chain's value-encoder
        # pass only computes batch statistics and its running-statistics momentum update is applied after the join.
        def actor_tail(g_pi):
            hip.call("gad_actor_loss", self.hs_p.out, self.pi, d["expert_action_batch"], d["expert_flag_batch"],
                     d["return_batch"], d["goal_batch"], B, self.pol.n_heads, 1.0 - ratio, int(bool(ag.policy_aux)),
                     self.action_scale, g_pi,
                     self.inv_n_actor(), self.hs_p.g_out, engine._ptr(self.scal, 4))
            P["p_bwd"].run()
            if self.fused_optim:
                self._reduce([self.pol.flat, self.enc.flat], "a")          # (no-op outside data-parallel runs)
                self._optim_phase("a", policy_step)
                return
            self._reduce([self.pol.flat, self.enc.flat], "a")
            self._adam(self.pol.flat, ag.policy_optim)
            if ag.train_feature:
                self._adam(self.enc.flat, ag.state_feat_encoder_optim)

        if OVERLAP_PASSES:
            s1, s2, sc = engine.side_stream(which=1), engine.side_stream(which=2), engine.side_stream(which=3)
            engine.apply_lane_priorities(main)
            # the side streams share weights and activation scratch with the previous step: they start after it (ev[0] on the
            # main stream, which every stream of that step was joined into) and after their inputs on the prefetch stream
            self._ev[0].record(main)
            main.wait_event(st["ev_gn"])
            P["t1"].run()                                           # enqueued FIRST: the host needs ~0.1 ms for the launches below
            # small independent launches (noise draw, result-slot clear, and for data-parallel runs the global mask counts: a
            # 4-double all-reduce) would sit on the critical chain in front of t1: on their own stream they overlap the
            # geometry and t1; the main stream -- and through _ev[2] the actor stream -- waits for them after t1
            sc.wait_event(self._ev[0])
            sc.wait_event(st["ev_g"])
            with torch.cuda.stream(sc):
                small_inits()
                if self.dp is not None:
                    self.dp.set_counts(batch)
                self._ev_counts.record(sc)
            main.wait_event(self._ev_counts)
            s1.wait_event(self._ev[0])
            s1.wait_event(st["ev_g"])
            with torch.cuda.stream(s1):
                P["c_fwd"].run()
        else:
            if self.dp is not None:
                self.dp.set_counts(batch)
            P["c_fwd"].run()
            small_inits()
            P["t1"].run()
        if OVERLAP_PASSES:
            # The actor phase's policy forward needs nothing from the critic update: it starts as soon as t1 is done (its
            # encoder BatchNorms must come after t1's, as in the reference) and runs beside t2 and the critic backward.
            # On steps without the actor-critic term (update_step % policy_update_gap != 0) the WHOLE actor phase --
            # loss, backward, Adam of policy + encoder -- is independent of the critic phase (disjoint parameters,
            # gradient arenas, dW lanes and result slots) and runs there too.
            self._ev[2].record(main)
        # the rest of the target chain is enqueued BEFORE the actor pass: the host needs ~0.4 ms for the actor pass's
        # launches, and the main stream (the critical path) would sit idle behind them
        hip.call("gad_target_noise", self.pi_t, self.noise_u, B, float(level), int(normal_noise), self.a_next)
        P["t2"].run()
        if OVERLAP_PASSES:
            s2.wait_event(self._ev[2])
            s2.wait_event(self._ev[0])
            s2.wait_event(st["ev_g"])
            with torch.cuda.stream(s2):
                P["p_fwd"].run()
                self._policy_outputs()
                if not policy_step:
                    actor_tail(None)
        if OVERLAP_PASSES:
            self._ev[1].record(s1)
            main.wait_event(self._ev[1])
            # the target chain's deferred running-statistics update: on the value stream, behind the value pass's own
            # updates (the reference's order) and the target pass (event), off the main stream's chain; the main stream picks
            # it up again before the next writer of those statistics
            self._ev[4].record(main)
            s1.wait_event(self._ev[4])
            with torch.cuda.stream(s1 if EARLY_ZERO else main):
                if not getattr(self, "_eval", False):      # (eval-mode BatchNorm: the running statistics stay)
                    P["t2_run"].run()
                self._ev_run.record(s1 if EARLY_ZERO else main)
        hip.call("gad_critic_loss", self.hs_c.out, self.hs_ct.out, d["reward_batch"], d["mask_batch"],
                 d["perturb_flag_batch"], d["return_batch"], d["goal_batch"], B, float(ag.gamma), int(bool(ag.critic_aux)),
                 self.inv_n_critic(), self.y, self.critic_aux_norm, self.hs_c.g_out, engine._ptr(self.scal, 0))
        P["c_bwd"].run()
        if OVERLAP_PASSES:
            main.wait_event(self._ev_run)           # (long done; orders the running statistics before the next value pass)
        if self.fused_optim:
            self._reduce([self.cr.flat, self.venc.flat], "c")
            if self.dp is not None:       # (single process: the backward plan's conversion launch already formed the sum of squares)
                hip.call("gad_sumsq", self.cr.flat.grad, self.cr.flat.n, self.clip_sumsq)
            self._optim_phase("c", policy_step)
        else:
            self._reduce([self.cr.flat, self.venc.flat], "c")
            hip.call("gad_sumsq", self.cr.flat.grad, self.cr.flat.n, self.clip_sumsq)
            self._adam(self.venc.flat, ag.state_feat_val_encoder_optim)
            self._adam(self.cr.flat, ag.critic_optim, clip=self.clip_sumsq)
        # ---- actor phase
        if OVERLAP_PASSES:
            self._ev[3].record(s2)
            main.wait_event(self._ev[3])
        else:
            P["p_fwd"].run()
            self._policy_outputs()
        if policy_step:
            P["v_fwd"].run()
            hip.call("gad_actor_critic_loss", self.hs_cpi.out, d["expert_flag_batch"], d["return_batch"], B, ratio,
                     self.inv_n_actor_critic(), self.hs_cpi.g_out, engine._ptr(self.scal, 8))
            P["v_bwd"].run()
            actor_tail(self.slot_v.daction)
        elif not OVERLAP_PASSES:
            actor_tail(None)
        if self.fused_optim:
            self._optim_phase("end", policy_step)
            self._download(sync=False)
            return
        self._target_updates()
        # (measured and not kept: the bookkeeping below on the small-launch lane instead of the main stream: 287 vs 297
        # steps/s -- that lane also carries the next step's uploads and geometry)
        self._stats()
        if not getattr(self, "_eval", False):
            self.enc.bump_batches_tracked(2)
            self.venc.bump_batches_tracked(3 if policy_step else 2)
        self._download(sync=False)

    def bc_step(self, batch):
        ag, d, P = self.agent, self.dbuf, self.plans
        B = self.B
        self._begin_step()
        self.upload(batch)
        if self.dp is not None:
            self.dp.set_counts(batch)
        self.scal.zero_()
        self.geo.run(d["point_state_batch"])
        P["p_fwd"].run()
        self._policy_outputs()
        hip.call("gad_actor_loss", self.hs_p.out, self.pi, d["expert_action_batch"], d["expert_flag_batch"],
                 d["return_batch"], d["goal_batch"], B, self.pol.n_heads, 1.0, int(bool(ag.policy_aux)), self.action_scale, None,
                 self.inv_n_actor(), self.hs_p.g_out, engine._ptr(self.scal, 4))
        P["p_bwd"].run()
        self._reduce([self.pol.flat, self.enc.flat], "a")      # (bucketed: the early bucket left from the plan's hook)
        self._adam(self.pol.flat, ag.policy_optim)
        if ag.train_feature:
            self._adam(self.enc.flat, ag.state_feat_encoder_optim)
        self._target_updates()
        self._stats()
        self.enc.bump_batches_tracked(1)
        return self._download()

    def _policy_outputs(self):
        """pi = tanh(mean) * scale and the aux pose (unit quaternion + translation) when the head has one"""
        nh = self.pol.n_heads
        hip.call("gad_policy_outputs", self.hs_p.out, self.B, nh, self.action_scale, self.action_bias, self.pi,
                 self.aux_pred if nh == 13 else None)

    # data-parallel hooks: device pointers to globally reduced 1/count pairs (None = local counts)
    def inv_n_critic(self):
        return None if self.inv_n is None else engine._ptr(self.inv_n, 0)

    def inv_n_actor(self):
        return None if self.inv_n is None else engine._ptr(self.inv_n, 2)

    def inv_n_actor_critic(self):
        return None if self.inv_n is None else engine._ptr(self.inv_n, 4)

    def _target_updates(self):
        ag = self.agent
        pt, p = self.pol_t.flat, self.pol.flat
        hip.call("gad_polyak", pt.master, p.master, None, pt.m2p, pt.packed, p.n, float(ag.tau), 0)
        if self.has_critic:
            ct, c = self.cr_t.flat, self.cr.flat
            hard = int(ag.update_step % ag.target_update_interval == 0)
            hip.call("gad_polyak", ct.master, c.master, self.critic_sel, ct.m2p, ct.packed, c.n, float(ag.tau), hard)

    def _stats(self):
        hip.call("gad_absmax_segments", self.pol.flat.master, self.seg["pol"], 1, engine._ptr(self.scal, 10))
        if self.has_critic:
            hip.call("gad_absmax_segments", self.cr.flat.grad, self.seg["cr"], 1, engine._ptr(self.scal, 11))
            hip.call("gad_absmax_segments", self.cr.flat.master, self.seg["cr"], 1, engine._ptr(self.scal, 12))

    def _host_flags(self):
        return {k: self.dbuf[k].cpu().numpy() for k in ("return_batch", "expert_flag_batch", "perturb_flag_batch")}

    def _download(self, sync=True):
        if self.dp is not None:
            self.dp.reduce_scalars(self.scal)
        self.scal_host.copy_(self.scal, non_blocking=True)
        if sync:
            torch.cuda.current_stream().synchronize()
        return self.scal_host.numpy()


# ----------------------------------------------------------------------------------------------
# module-level forward helpers (inference / feature extraction through the same kernels, no autograd)
# ----------------------------------------------------------------------------------------------
def _module_runtime(mod, key, builder):
    rt = mod.__dict__.get("_gad_rt")
    if rt is None:
        rt = {}
        object.__setattr__(mod, "_gad_rt", rt)
    if key not in rt:
        rt[key] = builder()
    return rt[key]


def sync_module(mod):
    """mirror a module's (externally modified) parameters into its packed compute copy, if it has one"""
    rt = mod.__dict__.get("_gad_rt") or {}
    if "net" in rt:
        rt["net"].flat.sync_packed()
    for net in (rt.get("nets") or {}).values():
        net.flat.sync_packed()


def _resync_on_load(module, flats):
    """load_state_dict() copies into the flat master views; mirror them into the packed compute copies"""
    def hook(mod, incompatible_keys):
        for f in flats:
            f.sync_packed()
    module.register_load_state_dict_post_hook(hook)


def _head_net(module, kind, dev):
    def build():
        net = heads.CriticNet(module, dev) if kind == "critic" else heads.PolicyNet(module, dev)
        _resync_on_load(module, [net.flat])
        return net
    return _module_runtime(module, "net", build)


def _encoder_nets(fe, dev):
    """EncoderNet views of a PointNetFeature's two encoders (shared with the agent's fused runtime if
    the parameters were already re-homed into flat buffers)."""
    def build():
        nets = {False: engine.EncoderNet(fe.encoder, dev), True: engine.EncoderNet(fe.value_encoder, dev)}
        _resync_on_load(fe, [nets[False].flat, nets[True].flat])
        return nets
    return _module_runtime(fe, "nets", build)


def feature_forward(fe, pc, value=False):
    """PointNetFeature.forward: pc (B,C,NP) -> z (B,512).  BatchNorm uses batch statistics (and updates the
    running ones) when fe.training, the running statistics otherwise -- as torch modules do."""
    hip.require_cuda(pc)
    B, C, NP = pc.shape
    dev = pc.device
    N = NP - 6 if NP != 1024 else NP
    encs = _encoder_nets(fe, dev)

    def build_b():
        geo = engine.Geometry(B, N, engine.SAConfig(fe.pointnet_nclusters, fe.pointnet_radius, 64),
                              engine.SAConfig(32, 0.04, 128), dev)
        slot = engine.EncoderSlot(geo, encs[False], dev, with_backward=False)
        act = torch.zeros(B, 6, device=dev)
        plans = {(v, t): engine.plan_encoder_forward(encs[v], slot, action=act if v else None, train=t)
                 for v in (False, True) for t in (False, True)}
        return dict(geo=geo, slot=slot, action=act, out=torch.empty(B, 512, device=dev), plans=plans)
    rt = _module_runtime(fe, ("shape", B, NP), build_b)
    enc = encs[bool(value)]
    rt["geo"].run(pc[:, :4].contiguous())
    if value:
        rt["action"].copy_(pc[:, 4:10, 0])
    rt["plans"][(bool(value), bool(fe.training))].run()
    if fe.training:
        enc.bump_batches_tracked(1)
    fc2 = enc.fc_mats[1]
    o = enc.bn_off[fc2.bn_index]
    slot = rt["slot"]
    hip.call("gad_affine_act", slot.Zfc[1], 512, B, 512, engine._ptr(slot.scale, o), engine._ptr(slot.shift, o), 1,
             rt["out"], 512)
    return rt["out"].clone()


class _FeatureSource(object):
    """Stands in for an (encoder, slot) pair when a head is evaluated on a plain feature tensor: identity
    'BatchNorm' (scale 1, shift 0), no ReLU, over a raw buffer holding the features as given."""
    input_relu = 0

    def __init__(self, B, dev):
        f32 = dict(dtype=torch.float32, device=dev)
        self.B, self.tot = B, 512
        self.Zfc = [None, torch.zeros(B, 512, **f32)]
        self.scale, self.shift = torch.ones(512, **f32), torch.zeros(512, **f32)
        self.mean, self.istd = torch.zeros(512, **f32), torch.ones(512, **f32)
        self.bstats = torch.zeros(hip.STAT_REPLICAS * 2 * 512, dtype=torch.float64, device=dev)
        self.time = torch.zeros(B, **f32)

        class _M(object):
            n_out, bn_index = 512, 0
        self.fc_mats = [None, _M()]
        self.bn_off = [0]

    def load(self, state):
        """state (B,513) = [feature (512), remaining time]"""
        self.Zfc[1].copy_(state[:, :512])
        self.time.copy_(state[:, 512])


def _head_runtime(module, kind, state, full=False):
    """full (policy only): the head plan also evaluates log_std_linear (GaussianPolicy.forward / sample)"""
    hip.require_cuda(state)
    B, dev = state.shape[0], state.device
    if state.shape[1] != 513:
        raise RuntimeError("heads take (B,513) = [512-d feature, remaining time]")

    net = _head_net(module, kind, dev)

    def build_b():
        src = _FeatureSource(B, dev)
        if kind == "critic":
            hs = heads.HeadSlot(B, net.width, 9, dev)
            plan = heads.plan_critic_forward(net, hs, src, src, src.time)
        else:
            hs = heads.HeadSlot(B, net.hidden, net.n_heads + (6 if full else 0), dev)
            plan = heads.plan_policy_forward(net, hs, src, src, src.time, with_log_std=full)
        return dict(src=src, hs=hs, plan=plan)
    rt = _module_runtime(module, ("B", B, bool(full)), build_b)
    rt["src"].load(state)
    rt["plan"].run()
    return net, rt


def critic_forward(module, state):
    """QNetwork.forward(state) -> (q1 (B,1), q2 (B,1), aux (B,7) with a unit quaternion | None)"""
    net, rt = _head_runtime(module, "critic", state)
    out = rt["hs"].out
    B = out.shape[0]
    aux = None
    if net.aux:
        aux = torch.empty(B, 7, device=out.device)
        z = torch.zeros(B, device=out.device)
        goal = torch.zeros(B, 7, device=out.device)
        hip.call("gad_critic_loss", out, out, z, z, z, z, goal, B, 0.0, 0, None, torch.empty(B, device=out.device), aux,
                 torch.empty(B, 9, device=out.device), torch.empty(4, device=out.device))
    return out[:, 0:1].clone(), out[:, 1:2].clone(), aux


def policy_forward(module, state):
    """GaussianPolicy: squashed mean action (B,6) and the aux head (B,7 unit-quaternion pose, or raw)."""
    net, rt = _head_runtime(module, "policy", state)
    out = rt["hs"].out
    B, dev = out.shape[0], out.device
    pi = torch.empty(B, 6, device=dev)
    scale = _dev_f32(module.action_scale, dev)
    bias = _dev_f32(module.action_bias, dev).reshape(-1)
    bias = bias.expand(6).contiguous() if bias.numel() == 1 else bias
    if net.extra_dim == 7:
        aux = torch.empty(B, 7, device=dev)
        hip.call("gad_policy_outputs", out, B, net.n_heads, scale, bias, pi, aux)
    else:
        hip.call("gad_policy_outputs", out, B, net.n_heads, scale, bias, pi, None)
        aux = out[:, 6:6 + net.extra_dim].clone()
    return pi, aux


def policy_sample(module, state, eps=None, draw=True):
    """GaussianPolicy.forward + sample (reference core/networks.py:339-371) through the head GEMMs and gad_policy_sample.
    eps (B,6): the N(0,1) draw of the reparameterised sample (default: torch.randn on the device, as Normal.rsample
    draws it); draw=False evaluates at eps = 0.  -> dict(mean (raw), log_std (clamped), extra, mean_sq, log_prob (B,1),
    action)."""
    net, rt = _head_runtime(module, "policy", state, full=True)
    out = rt["hs"].out
    B, dev = out.shape[0], out.device
    f32 = dict(dtype=torch.float32, device=dev)
    if eps is None and draw:
        eps = torch.randn(B, 6, **f32)
    elif eps is not None:
        eps = torch.as_tensor(eps, dtype=torch.float32).to(dev).contiguous()
    squash = module.action_space is not None
    scale = _dev_f32(module.action_scale, dev).reshape(-1)
    bias = _dev_f32(module.action_bias, dev).reshape(-1)
    if scale.numel() == 1:
        scale, bias = scale.expand(6).contiguous(), bias.expand(6).contiguous()
    res = dict(mean_sq=torch.empty(B, 6, **f32), log_std=torch.empty(B, 6, **f32), log_prob=torch.empty(B, 1, **f32),
               action=torch.empty(B, 6, **f32), extra=torch.empty(B, net.extra_dim, **f32))
    hip.call("gad_policy_sample", out, B, out.shape[1], net.extra_dim, eps, scale, bias, int(squash), res["mean_sq"],
             res["log_std"], res["log_prob"], res["action"], res["extra"])
    res["mean"] = out[:, :6].clone()
    return res
