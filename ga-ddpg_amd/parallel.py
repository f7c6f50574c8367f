"""Data-parallel update step: one process per GPU, batch rows sharded across ranks, gradients summed
with ONE all-reduce per optimiser phase over the flat gradient buffers (RCCL over xGMI with backend
'nccl'; 'gloo' works for the CPU tests).

What the reference does: nn.DataParallel around the feature extractors only (core/utils.py:202):
replicas normalise their own chunk (per-replica BatchNorm statistics), heads and losses see the
whole batch, so every masked mean is a GLOBAL-batch mean.  To reproduce that with independent ranks:
  * BatchNorm statistics stay local to the rank (= a DataParallel replica);
  * each rank's loss kernels divide by the GLOBAL mask counts (all-reduced 5-float vector per
    step), so the SUM of the ranks' gradients is the global-mean gradient -- no post-division;
  * clip_grad_norm_ is evaluated after the all-reduce, identically on every rank;
  * optimiser / target-network state is replicated and advances identically (no exchange).
Buckets (BASELINE configs[4]: "overlapped sample-prefetch and all-reduce"): the backward pass finishes its weight gradients in
the order head -> FC -> SA3 -> SA2 -> SA1, and SA1 -- 13 k of the encoder's 1.4 M parameters -- is the longest stage.  So each
optimiser phase exchanges [head | encoder without SA1] (critic phase 0.59 M + 1.39 M floats = 7.9 MB, actor phase 0.20 M +
1.39 M = 6.4 MB) asynchronously as soon as the SA2 backward is done, under the SA1 backward, and the 52 KB SA1 slice at
the end (FusedRuntime.enable_bucketed_reduce; ring all-reduce of 8 MB over xGMI ~ 91 us per-link bound, the SA1 backward
~ 0.3 ms).  The result is bit-identical to one exchange of the whole buffer (tests/test_parallel_gloo.py).
Mask counts are formed on the device (gad_mask_counts) and exchanged stream-ordered: no host pass over the flags.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

# Two buckets per phase, the first all-reduced under the SA1 backward (above); "0": one exchange per phase after the whole
# backward pass.  Round 2 measured the bucketed exchange at -9 % at one rank because torch.distributed runs an asynchronous
# collective on a stream of its own -- a fifth active queue in a process that keeps four busy (engine._PHYS).  With the
# collectives issued through RCCL's C API on the weight-gradient lane itself (rccl.Communicator, DIRECT below) there is no
# extra stream.  MEASURED, round 3, one rank, same box (tools/ab_dp2.sh; GAD_BENCH_FORCE_DP=1): no hooks 320.0 steps/s, one
# exchange per phase 318.7 (direct) / 316.8 (torch.distributed), bucketed 319.4 (direct) / 288.8 (torch.distributed).  So
# buckets are the default wherever the direct path is available (backend 'nccl', world > 1); over gloo / torch.distributed
# the default stays one exchange per phase.
BUCKETED = {None: None, "0": False, "1": True}[os.environ.get("GAD_DP_BUCKETS")]        # None: decided in attach()
# gradient / count exchanges of CUDA tensors through rccl.Communicator on the caller's stream (backend 'nccl' only: gloo
# groups -- the CPU tests, two test ranks sharing one GPU -- keep torch.distributed)
DIRECT = os.environ.get("GAD_DP_DIRECT_RCCL", "1") != "0"
# GAD_DP_COMMS: "one" (default, round 6) = ONE RCCL communicator: its collectives are serialised in host-issue order on every rank,
# the only order RCCL guarantees deadlock-free; "lanes" = one communicator per issuing stream (see _make_comm): exchanges of
# different lanes become independent operations that may overlap, but concurrent collectives on different communicators of one
# process are only safe if every rank's GPU starts them in the same order -- which host-issue order on four streams does not
# imply.  N > 1 has never run on hardware here, so the first curve is taken with the safe setting;
# tools/first_multigpu_run.sh measures both.
PER_LANE_COMMS = os.environ.get("GAD_DP_COMMS", "one") == "lanes"
CORUN_EVERY = int(os.environ.get("GAD_DP_CORUN_EVERY", "2000"))      # steps between co-run self-checks of a multi-rank job (0: only at attach)
INIT_TIMEOUT_S = float(os.environ.get("GAD_DP_INIT_TIMEOUT", "180"))     # watchdog on ncclCommInitRank (a rank that failed leaves the others inside it)


class _SplitAggressor(object):
    """back-to-back split-bf16 GEMM launches over private buffers (the SA2 wide-tile forward and the SA1 streaming forward at
    their B = 256 shapes): what corun_check runs beside the exchanges.  Results are discarded."""

    def __init__(self, dev):
        from . import hip
        from .engine import _fwd_args, _ptr
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        self.calls, self.keep = [], []
        for rows, K, N in ((27240, 128, 128), (213034, 64, 64)):
            zin = torch.randn(rows, K, device=dev, generator=g)
            scale, shift = torch.rand(K, device=dev, generator=g) + 0.5, torch.randn(K, device=dev, generator=g) * 0.3
            W = torch.randn(N, K, device=dev, generator=g) * 0.05
            nrows = torch.tensor([rows], dtype=torch.int32, device=dev)
            zout = torch.empty(rows, N, device=dev)
            stats = torch.zeros(hip.STAT_REPLICAS * 2 * N, dtype=torch.float64, device=dev)
            plane = N * K
            mirror = torch.zeros(6 * plane, dtype=torch.int16, device=dev)
            lay = (hip.SplitLayer * 1)()
            lay[0].w_off, lay[0].n_out, lay[0].Kp, lay[0].Ks, lay[0].fwd_off, lay[0].t_off = 0, N, K, K, 0, 3 * plane
            hip.call("gad_split_weights", W, lay, 1, mirror)
            a = _fwd_args(n_rows_dev=_ptr(nrows), n_rows=rows, W=_ptr(W), Kp=K, n_out=[N], zout=_ptr(zout), zout_pitch=N,
                          stat_sum=_ptr(stats, 0, 8), stat_sq=_ptr(stats, N, 8), stat_stride=2 * N, mode=0, zin=_ptr(zin), zin_pitch=K,
                          c_in=K, scale=_ptr(scale), shift=_ptr(shift), relu=1, W_split=_ptr(mirror, 0, 2), W_split_pitch=K,
                          W_split_plane=plane)
            self.calls.append(a)
            self.keep.append((zin, scale, shift, W, nrows, zout, stats, mirror, lay))
        self.routes = []

    def launch(self, n):
        """n rounds of the launches on the CURRENT stream, with the split-bf16 form forced on for them"""
        import ctypes as C
        from . import hip
        was = hip.get_option("mfma_split")
        hip.set_option("mfma_split", 1)
        try:
            f, st = hip.lib().gad_gemm_fwd, hip.stream()
            for _ in range(n):
                for a in self.calls:
                    hip.check(f(C.byref(a), st), "gad_gemm_fwd")
                    if len(self.routes) < len(self.calls):
                        self.routes.append(hip.lib().gad_last_kernel().decode())
        finally:
            hip.set_option("mfma_split", was)

    def describe(self):
        return ", ".join(self.routes)


def mask_counts(batch):
    """local counts the masked means divide by: [kept, goal rows, expert rows, non-(expert&reward) rows]"""
    if "mask_counts" in batch:               # precomputed for a device-resident batch
        return np.asarray(batch["mask_counts"], dtype=np.float64)

    def host(x):
        return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
    batch = {k: host(batch[k]) for k in ("return_batch", "expert_flag_batch", "perturb_flag_batch")}
    ret = np.asarray(batch["return_batch"]).reshape(-1)
    exp = np.asarray(batch["expert_flag_batch"]).reshape(-1)
    per = np.asarray(batch["perturb_flag_batch"]).reshape(-1)
    reward = ret > 0
    expert = exp >= 1
    return np.array([(per < 1).sum(), reward.sum(), expert.sum(), (~(reward & expert)).sum()], dtype=np.float64)


def inverse_counts(c):
    """[1/kept, 1/(6*goal), 1/(6*expert), 1/(6*goal), 1/rows_ac, 0] -- layout read by the loss kernels:
    critic at +0, actor at +2, actor-critic at +4 (runtime.inv_n_*).  Division by zero gives inf/NaN
    on purpose: the reference's mean over an empty mask is NaN as well (core/loss.py:23,31)."""
    with np.errstate(divide="ignore"):
        return np.array([1.0 / c[0], 1.0 / (6.0 * c[1]), 1.0 / (6.0 * c[2]), 1.0 / (6.0 * c[1]), 1.0 / c[3], 0.0, 0.0, 0.0])


class DataParallelContext(object):
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised before DataParallelContext")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.rt = None
        self._comm = None
        self._comms, self._lane_comm = [], {}
        self._direct = DIRECT and dist.get_backend(group) == "nccl" and torch.cuda.is_available()

    @property
    def comm(self):
        """rccl.Communicator of this group (None: collectives go through torch.distributed).  Built by attach() /
        broadcast_parameters() -- outside any stream context -- or at first use."""
        if self._direct and self._comm is None:
            self._make_comm()
        return self._comm

    def _agree(self, ok):
        """MIN over the ranks of a 0 / 1 flag, over the bootstrap group (every rank calls this at the same point)"""
        flag = torch.tensor([int(ok)], dtype=torch.int32, device="cuda" if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return int(flag.item()) == 1

    def _init_comm_watched(self, rccl, uid):
        """ncclCommInitRank under a watchdog: the call blocks until EVERY rank has entered it, so a rank whose own init failed
        (or never got here) would leave the others inside it for ever, in front of the agreement that is meant to catch exactly
        that.  The init runs on a helper thread; after INIT_TIMEOUT_S this rank gives up on the direct transport
        (the helper stays blocked, a daemon) -- the peers time out the same way and the agreement sends everybody to
        torch.distributed."""
        import threading
        box = {}
        dev = torch.cuda.current_device() if torch.cuda.is_available() else None

        def work():
            try:
                if dev is not None:
                    torch.cuda.set_device(dev)
                box["comm"] = rccl.Communicator(self.group, uid=uid)
            except Exception as exc:            # noqa: BLE001
                box["exc"] = exc
        th = threading.Thread(target=work, name="gad-rccl-init", daemon=True)
        th.start()
        th.join(INIT_TIMEOUT_S)
        if th.is_alive():
            raise RuntimeError("ncclCommInitRank did not return within %.0f s (a peer failed to enter it?)" % INIT_TIMEOUT_S)
        if "exc" in box:
            raise box["exc"]
        return box["comm"]

    def _make_comm(self):
        """Build the direct-RCCL communicators -- ONE PER ISSUING LANE (main stream, value / weight-gradient lane A, actor lane
        B, small-launch / weight-gradient lane C: engine._PHYS), so that exchanges issued from different streams are independent
        RCCL operations and can overlap (one communicator serialises its operations in host-issue order, whatever streams
        they ride on: VERDICT r04 item 5; GAD_DP_COMMS=one keeps a single communicator) -- PROVE each (a known-answer
        all-reduce on its stream) and agree on the outcome over the bootstrap group IN PHASES, so that no rank can skip a
        collective another rank is waiting in: (1) library + streams, (2) the unique ids are broadcast by every
        rank unconditionally, then ncclCommInitRank, (3) every self-test runs on every rank before the verdict is reduced.
        If any rank fails anywhere, EVERY rank logs once and falls back to torch.distributed for this context (never a mix of
        transports inside one job).  GAD_DP_DIRECT_RCCL=0 skips the attempt."""
        import sys
        from . import rccl
        why, comms, streams = "", [], []

        def fail(exc):
            return "%s: %s" % (type(exc).__name__, exc)
        # ---- phase 1: librccl loads and every stream of the step has launched something
        ok = 1
        try:
            rccl.lib()
            streams = self._warm_streams()
        except Exception as exc:                # noqa: BLE001 -- whatever went wrong, the job continues on torch.distributed
            ok, why = 0, fail(exc)
        own = ok
        agreed = self._agree(ok)
        # ---- phase 2: unique ids (rank 0) -> every rank; one ncclCommInitRank per lane
        if agreed:
            n = len(streams) if PER_LANE_COMMS else 1
            box = [None]
            if self.rank == 0:
                try:
                    box = [[rccl.unique_id() for _ in range(n)] if hasattr(rccl, "unique_id") else None]
                except Exception as exc:        # noqa: BLE001
                    ok, why = 0, fail(exc)
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.broadcast_object_list(box, src=src, group=self.group)          # unconditionally: nobody waits alone
            try:
                if box[0] is None:
                    raise RuntimeError("rank 0 could not draw the ncclUniqueIds")
                for uid in box[0]:
                    comms.append(self._init_comm_watched(rccl, uid))
                    if comms[-1].count() != self.world:
                        raise RuntimeError("ncclCommCount = %d, group has %d ranks" % (comms[-1].count(), self.world))
            except Exception as exc:            # noqa: BLE001
                ok, why = 0, fail(exc)
            own = ok
            agreed = self._agree(ok)
        # ---- phase 3: a known-answer all-reduce of every communicator on its lane's stream, ALL of them on every rank
        if agreed:
            for i, st in enumerate(streams):
                comm = comms[i if len(comms) > 1 else 0]
                try:
                    with torch.cuda.stream(st):
                        if not comm.self_test():
                            ok, why = 0, "known-answer all-reduce returned a wrong sum (lane %d)" % i
                except Exception as exc:        # noqa: BLE001
                    ok, why = 0, fail(exc)
            own = ok
            agreed = self._agree(ok)
        self._make_comm_outcome = (own, why)                # (this rank's own attempt, before the agreement: tests / logs)
        if agreed:
            self._comm = comms[0]                           # the main stream's (and, with GAD_DP_COMMS=one, everybody's)
            self._comms = comms
            self._lane_comm = {}
            for i, st in enumerate(streams):
                h = getattr(st, "cuda_stream", None)
                if h is not None:
                    self._lane_comm[int(h)] = comms[i if len(comms) > 1 else 0]
            import atexit
            atexit.register(self.close)
            return
        for c in comms:
            try:
                c.destroy()
            except Exception:                   # noqa: BLE001
                pass
        self._direct = False
        self._comm = None
        self._comms, self._lane_comm = [], {}
        print("ga_ddpg_amd.parallel: rank %d: direct RCCL path unavailable (%s): every collective of this context goes "
              "through torch.distributed" % (self.rank, why or "another rank failed"), file=sys.stderr, flush=True)

    def _lane(self):
        """the communicator of the CURRENT stream's lane (the main stream's for a stream the step does not know)"""
        if self.comm is None:
            return None
        return self._lane_comm.get(int(torch.cuda.current_stream().cuda_stream), self._comm)

    @staticmethod
    def _warm_streams():
        """HIP binds a stream to one of the process's four hardware queues at the stream's FIRST LAUNCH, and ncclCommInitRank
        launches on streams of its own: created before the step's streams have run anything, the communicator takes queues
        and the step's chains end up sharing one (measured: 318 -> 260 steps/s at one rank).  So every stream of the step
        launches something first; returns those streams (the known-answer all-reduce then runs on each of them)."""
        from . import engine
        dev = torch.cuda.current_device()
        streams = [torch.cuda.current_stream(dev)] + [engine.side_stream(dev, w) for w in (1, 2, 3)]
        for st in streams:
            with torch.cuda.stream(st):
                torch.zeros(64, device="cuda").add_(1.0)
        torch.cuda.synchronize(dev)
        return streams

    def close(self):
        """destroy the RCCL communicators (before the process group goes away; registered with atexit)"""
        cs, self._comm, self._comms, self._lane_comm = list(self._comms), None, [], {}
        for c in cs:
            try:
                c.destroy()
            except Exception:                   # noqa: BLE001
                pass

    def transport(self):
        """what carries this context's CUDA collectives, and how many ranks IT reports (bench.py: config.rccl_nranks)"""
        if self._comm is not None:
            info = {"transport": "rccl-direct", "rccl_nranks": self._comm.count(), "rccl_comms": len(self._comms)}
        else:
            info = {"transport": "torch.distributed/%s" % dist.get_backend(self.group), "rccl_nranks": dist.get_world_size(self.group)
                    if dist.get_backend(self.group) == "nccl" else None}
        info.update(getattr(self, "_corun", None) or {"allreduce_corun_checked": False})
        return info

    def replicas_agree(self, flats):
        """True iff every rank holds bit-identical parameters: the all-reduced MAX and MIN of a per-rank checksum (sum of
        the parameters' int32 bit patterns, exact in int64) coincide.  bench.py puts the flag on its JSON line."""
        dev = flats[0].master.device
        cs = torch.zeros(1, dtype=torch.int64, device=dev)
        for f in flats:
            cs += f.master.detach().view(torch.int32).to(torch.int64).sum()
        hi, lo = cs.clone(), cs.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        return bool((hi == lo).item())

    def _sum(self, t):
        """in-place SUM over the ranks, ordered on the CURRENT stream"""
        c = self._lane() if t.is_cuda else None
        if c is not None:
            c.all_reduce_(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def attach(self, rt):
        self.rt = rt
        rt.world_size = self.world
        rt.dp = self
        rt.allreduce = self.allreduce_grads
        rt.inv_n = torch.zeros(8, dtype=torch.float32, device=rt.dev)
        self._counts = torch.zeros(4, dtype=torch.float64, device=rt.dev)
        from . import engine
        self._counts_host = torch.zeros(engine.HOST_RING, 4, dtype=torch.float64)      # one row per in-flight step
        if torch.device(rt.dev).type == "cuda":
            self._counts_host = self._counts_host.pin_memory()
        # inv_n[0:5] = numer / counts[pick]   (layout: inverse_counts)
        self._numer = torch.tensor([1.0, 1.0 / 6.0, 1.0 / 6.0, 1.0 / 6.0, 1.0], dtype=torch.float64, device=rt.dev)
        self._pick = torch.tensor([0, 1, 2, 1, 3], dtype=torch.int64, device=rt.dev)
        self._inflight = {}
        if self._direct and self._comm is None:
            self._make_comm()                       # eagerly, outside any stream context; may fall back (self._direct False)
        # the first exchanges of a multi-rank job run beside split-bf16 GEMMs and are compared with their closed form (corun_check);
        # GAD_DP_CORUN_CHECK=0 skips, =1 forces it for a one-rank group too (tests)
        want_check = os.environ.get("GAD_DP_CORUN_CHECK", "auto")
        self._corun, self._steps_seen = None, 0
        if torch.device(rt.dev).type == "cuda" and (want_check == "1" or (want_check == "auto" and self.world > 1)):
            self.corun_check()
        bucketed = BUCKETED if BUCKETED is not None else (self._direct and self.world > 1)
        if bucketed and hasattr(rt, "enable_bucketed_reduce"):
            rt.enable_bucketed_reduce()
        elif hasattr(rt, "_build_all_plans"):
            rt._build_all_plans()                   # the backward plans end differently once an exchange follows them

    # ------------------------------------------------------------------ co-run self-check
    def corun_check(self, n_elems=None, rounds=2):
        """Known-answer exchanges WHILE split-bf16 GEMM launches run on the other lanes.  Round 5 found that, beside wavefronts
        issuing v_mfma_f32_32x32x16_bf16, a packed-f32 instruction consuming LDS-fresh registers can compute with stale lanes
        (DESIGN.md section 5); libgaddpg is built without packed-f32 instructions, RCCL's reduction kernels and torch's
        elementwise kernels are not -- and a corrupted ring-reduce leaves every rank with the SAME wrong sum, which the replicas'
        bit-identity check cannot see.  Here, on every lane that carries exchanges, a gradient-bucket-sized buffer of small integers
        (rank-dependent, exactly representable: any summation order gives the same bits) is all-reduced through the transport the
        step uses while a second stream issues split wide-tile and streaming GEMM launches back to back, and compared bit for bit
        with the closed-form sum; the torch kernels that share the step's lanes (the fill and the uniform draw of the TD3 noise)
        are checked the same way against their results alone.  Collective: every rank must call it.  -> dict for the bench line
        (allreduce_corun_checked, corun_mismatches, ...)."""
        from . import engine
        dev = torch.cuda.current_device()
        rt = getattr(self, "rt", None)
        if n_elems is None:
            n_elems = int(rt.bucket_c.numel()) if (rt is not None and getattr(rt, "bucket_c", None) is not None) else (1 << 21)
        agg = _SplitAggressor(torch.device("cuda", dev))
        main = torch.cuda.current_stream(dev)
        lanes = [main] + [engine.side_stream(dev, w) for w in (1, 2, 3)]
        i = torch.arange(n_elems, device="cuda:%d" % dev, dtype=torch.float32)
        base = torch.remainder(i, 251.0)
        want = (base * float(self.world) + float(self.world * (self.world - 1) // 2)).clone()
        bad, n_checks = 0, 0
        torch.cuda.synchronize(dev)
        for rnd in range(rounds):
            for li, lane in enumerate(lanes):
                other = lanes[(li + 1) % len(lanes)]
                bufs = [(base + float(self.rank)).clone() for _ in range(3)]
                torch.cuda.synchronize(dev)
                with torch.cuda.stream(other):
                    agg.launch(40)
                with torch.cuda.stream(lane):
                    for b in bufs:
                        self._sum(b)
                torch.cuda.synchronize(dev)
                for b in bufs:
                    n_checks += 1
                    bad += int(not torch.equal(b, want))
        # torch's own kernels that share the lanes: a fill and the generator's uniform draw, beside the same launches vs alone
        gen = torch.Generator(device="cuda:%d" % dev)
        gen.manual_seed(1234)
        ref_u = torch.empty(256, 6, device="cuda:%d" % dev).uniform_(0.0, 1.0, generator=gen)
        torch.cuda.synchronize(dev)
        t_bad = 0
        for rnd in range(rounds):
            with torch.cuda.stream(lanes[1]):
                agg.launch(40)
            with torch.cuda.stream(lanes[3]):
                for _ in range(8):
                    gen.manual_seed(1234)
                    u = torch.empty(256, 6, device="cuda:%d" % dev).uniform_(0.0, 1.0, generator=gen)
                    z = torch.full((1 << 16,), 3.0, device="cuda:%d" % dev)
                    z.zero_()
                    t_bad += int(not torch.equal(u, ref_u)) + int(bool((z != 0).any().item()))
            torch.cuda.synchronize(dev)
        # every rank reports the job-wide verdict
        v = torch.tensor([float(bad), float(t_bad)], dtype=torch.float64, device="cuda:%d" % dev if self._counts_is_cuda() else "cpu")
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
        bad_all, t_bad_all = int(v[0].item()), int(v[1].item())
        self._corun = {"allreduce_corun_checked": bad_all == 0 and t_bad_all == 0, "corun_exchanges_checked": n_checks,
                       "corun_exchange_mismatches": bad_all, "corun_torch_kernel_mismatches": t_bad_all,
                       "corun_elems": int(n_elems), "corun_aggressor": agg.describe()}
        if bad_all or t_bad_all:
            import sys
            print("ga_ddpg_amd.parallel: rank %d: CO-RUN CHECK FAILED: %d of %d known-answer all-reduces and %d torch launches "
                  "beside split-bf16 GEMMs differ from their closed form (job-wide)" % (self.rank, bad_all, n_checks * self.world, t_bad_all),
                  file=sys.stderr, flush=True)
        return dict(self._corun)

    def step_hook(self):
        """called by the agent before every update step of a data-parallel run: every GAD_DP_CORUN_EVERY steps (default 2000,
        0 = never) the co-run check is repeated -- between two steps, with the GPU drained (a collective: every rank counts the
        same steps)"""
        self._steps_seen = getattr(self, "_steps_seen", 0) + 1
        if CORUN_EVERY > 0 and self.world > 1 and self._steps_seen % CORUN_EVERY == 0 and getattr(self, "rt", None) is not None:
            if hasattr(self.rt, "flush"):
                self.rt.flush()
            self.corun_check(rounds=1)

    def _counts_is_cuda(self):
        try:
            return dist.get_backend(self.group) == "nccl"
        except Exception:                       # noqa: BLE001
            return False

    def reduce_early(self, tag, tensors):
        """start the exchange of a bucket whose gradients are complete, ordered after the launches enqueued so far on the
        current stream (the weight-gradient lane that produced the bucket).  Direct RCCL: the collective is a kernel of
        that very stream and an event marks its end; torch.distributed: asynchronous on the backend's own stream."""
        self._inflight.setdefault(tag, [])
        for t in tensors:
            c = self._lane() if t.is_cuda else None
            if c is not None:
                c.all_reduce_(t)
                ev = torch.cuda.Event()
                ev.record()
                self._inflight[tag].append(ev)
            else:
                self._inflight[tag].append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def reduce_finish(self, tag, tensors):
        """exchange the rest of the phase's gradients and make the current stream wait for every bucket of the phase"""
        works = self._inflight.pop(tag, [])
        for t in tensors:
            c = self._lane() if t.is_cuda else None
            if c is not None:
                c.all_reduce_(t)
            else:
                works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            if isinstance(w, torch.cuda.Event):
                torch.cuda.current_stream().wait_event(w)
            else:
                w.wait()

    def allreduce_grads(self, tensors):
        for t in tensors:
            self._sum(t)

    def set_counts(self, batch):
        """global mask counts -> rt.inv_n: one 4-double all-reduce per step, stream-ordered (no host sync: counts and
        reciprocals are formed on the device; an empty mask gives inf -> NaN losses, like the reference)"""
        d = getattr(self.rt, "dbuf", None)
        if d is not None and self._counts.is_cuda:
            from . import hip
            hip.call("gad_mask_counts", d["return_batch"], d["expert_flag_batch"], d["perturb_flag_batch"], self.rt.B,
                     self._counts)
        else:
            h = self._counts_host[getattr(self.rt, "_slot", 0)]
            h.numpy()[:] = mask_counts(batch)
            self._counts.copy_(h, non_blocking=True)
        self._sum(self._counts)
        self.rt.inv_n[0:5] = (self._numer / self._counts.index_select(0, self._pick)).float()

    def global_counts(self, local, device):
        t = torch.as_tensor(local, dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def reduce_scalars(self, scal):
        """losses / counts are partial sums per rank (already divided by the global counts)"""
        self._sum(scal[0:10])

    def broadcast_parameters(self, flats):
        for f in flats:
            c = self._lane() if f.master.is_cuda else None
            if c is not None:
                c.broadcast_(f.master, root=0)
            else:
                dist.broadcast(f.master, src=0, group=self.group)
            f.sync_packed()
