"""GPU-resident mirror of the replay buffer ("the step before the path").

The reference samples a minibatch on the host (core/replay_memory.py:166-176: fancy-index every array, two
(B,4,1030) float64 gathers = 17 MB per step) and ships 22 arrays to the device (core/agent.py:211-240).  At > 200
update steps/s that costs more than the update itself.  `DeviceReplay` keeps the arrays the update step reads in
HBM as float32 and performs the gather there; the INDEX arithmetic stays on the host and is the reference's,
so a minibatch drawn here equals `BaseMemory.sample` with the same `batch_idx` (tests/test_gpu_modules.py:
test_device_replay_matches_host_sampling).

    mem  = BaseMemory(...); mem.load(...) / filled by the environment loop
    dmem = DeviceReplay(mem)                     # one upload (cap x 4 x 1030 float32 for the clouds)
    batch = dmem.sample(256, rng)                # dict of CUDA tensors + host-side mask counts
    agent.update_parameters(batch, agent.update_step, k)

`refresh(lo, hi)` re-uploads a slice after the host buffer was written to (online training).

Hindsight relabelling (`memory.self_supervision` on a non-expert buffer, reference core/replay_memory.py:233-249,271-272): the
relabelled goals are 4x4 pose algebra on B rows -- host work, done by the SAME BaseMemory.onpolicy_goals the host path
uses -- and travel with the index vectors as a (B, 8) block [goal (7) | relabel flag]; the gather's goal rows are
overwritten where the flag is set, so both paths train on the same goals.
"""
import numpy as np
import torch

from .. import hip

_ROW_KEYS = (("action", "action_batch"), ("expert_action", "expert_action_batch"), ("reward", "reward_batch"),
             ("returns", "return_batch"), ("terminal", "mask_batch"), ("goal", "goal_batch"),
             ("expert_flags", "expert_flag_batch"), ("perturb_flags", "perturb_flag_batch"))


class _Shape(object):
    """placeholder that only answers `.shape` (the agent sizes its runtime from point_state_batch.shape)"""

    def __init__(self, shape):
        self.shape = tuple(shape)


class DeviceReplay(object):
    def __init__(self, memory, device=None):
        self.memory = memory
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceReplay needs a GPU (the update path has no CPU fallback)")
        cap = memory.point_state.shape[0]
        self.cap = cap
        f32 = dict(dtype=torch.float32, device=self.device)
        self.point_state = torch.empty(tuple(memory.point_state.shape), **f32)
        self.rows = {src: torch.empty(tuple(getattr(memory, src).shape), **f32) for src, _ in _ROW_KEYS}
        self.timestep = torch.empty(cap, **f32)
        self._stage = {}
        self._copy_stream = None
        self._ev_refresh = None
        self.refresh()

    # ------------------------------------------------------------------ host -> device
    def refresh(self, lo=0, hi=None):
        """(re-)upload transitions [lo, hi) of the host buffer"""
        m = self.memory
        hi = self.cap if hi is None else hi
        if hi <= lo:
            return
        sl = slice(lo, hi)
        # float64 -> float32 on the host in bounded chunks (the cloud array is 33 KB per transition)
        step = 4096
        for a in range(lo, hi, step):
            b = min(a + step, hi)
            self.point_state[a:b].copy_(torch.from_numpy(np.ascontiguousarray(m.point_state[a:b], dtype=np.float32)))
        for src, _ in _ROW_KEYS:
            self.rows[src][sl].copy_(torch.from_numpy(np.ascontiguousarray(getattr(m, src)[sl], dtype=np.float32)))
        self.timestep[sl].copy_(torch.from_numpy(np.ascontiguousarray(m.timestep[sl], dtype=np.float32)))
        if self._ev_refresh is None:
            self._ev_refresh = torch.cuda.Event()
        self._ev_refresh.record(torch.cuda.current_stream())      # later handles become ready after this upload

    RING = 8        # index staging sets (a set is reused only after the gather that read it has run: `used` event)

    def _stage_set(self, n):
        """the next handle's staging set: dict(host = pinned (3,n) int64, dev = device (3,n) int64, copied = event of the
        host -> device copy, used = event recorded after the LAST gather that read `dev`, or None).  The copies run on a
        stream of their own and the handle carries `copied`: a runtime that enqueues steps ahead of the GPU
        (update_parameters(sync=False)) starts the gather as soon as the indices are on the device, not after everything
        queued on the caller's stream.  Reuse is guarded by BOTH events: the pinned block by the copy that read it, the
        device block by the gather that read it -- however far ahead handles are drawn (PrefetchSampler depth, host ring),
        a set never changes under a gather that is still queued; a set whose handle has not been gathered at all yet
        (`pending`) is skipped, and the ring grows if every set is held that way."""
        key = ("sets", n)
        sets = self._stage.get(key)
        if sets is None:
            sets = self._stage[key] = {"next": 0, "items": [None] * self.RING}
        items = sets["items"]
        j = sets["next"]
        for _ in range(len(items)):              # a set whose handle has not been gathered yet is never handed out again
            if items[j] is None or not items[j]["pending"]:
                break
            j = (j + 1) % len(items)
        else:                                    # every set is held by a handle drawn ahead of its use: grow the ring
            if len(items) >= 64:
                raise RuntimeError("DeviceReplay: 64 sample_lazy() handles are outstanding (drawn but never passed to an "
                                   "update step); drop-and-redraw loops should use sample() instead")
            items.append(None)
            j = len(items) - 1
        sets["next"] = (j + 1) % len(items)
        it = items[j]
        if it is None:
            it = items[j] = {"host": torch.empty(3, n, dtype=torch.int64).pin_memory(),
                             "dev": torch.empty(3, n, dtype=torch.int64, device=self.device),
                             "ghost": torch.zeros(n, 8, dtype=torch.float32).pin_memory(),       # [relabelled goal | flag]
                             "gdev": torch.zeros(n, 8, dtype=torch.float32, device=self.device),
                             "copied": torch.cuda.Event(), "used": None, "pending": False}
        else:
            it["copied"].synchronize()           # the copy that last read this pinned block
            if it["used"] is not None:
                it["used"].synchronize()         # the gather(s) that read the device block
        it["pending"] = True
        return it

    def _relabels(self):
        m = self.memory
        return bool(getattr(m, "self_supervision", False)) and getattr(m, "name", "") != "expert"

    def _indices3(self, idx, nxt, end):
        it = self._stage_set(idx.shape[0])
        host, dev, ev = it["host"], it["dev"], it["copied"]
        h = host.numpy()
        h[0], h[1], h[2] = idx, nxt, end
        relabel = self._relabels()
        if relabel:                              # hindsight goals of the on-policy rows (BaseMemory.post_process_batch)
            mask, goal, _ = self.memory.onpolicy_goals(idx)
            g = it["ghost"].numpy()
            g[:, :7] = goal
            g[:, 7] = np.asarray(mask, dtype=np.float32).reshape(-1)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        if self._ev_refresh is not None:
            self._copy_stream.wait_event(self._ev_refresh)
        with torch.cuda.stream(self._copy_stream):
            dev.copy_(host, non_blocking=True)
            if relabel:
                it["gdev"].copy_(it["ghost"], non_blocking=True)
            ev.record(self._copy_stream)
        it["relabel"] = relabel
        return dev, ev, it

    @staticmethod
    def _apply_relabel(it, goal):
        """goal rows <- the staged hindsight goals where their flag is set (same stream as the gather that wrote `goal`)"""
        g = it["gdev"]
        goal.copy_(torch.where(g[:, 7:8] > 0, g[:, :7], goal))

    # ------------------------------------------------------------------ sampling
    def sample_lazy(self, batch_size, rng=None, batch_idx=None):
        """like sample(), but nothing is gathered yet: the returned dict carries the device index vectors and
        `FusedRuntime.upload` fills its static input buffers with ONE gad_replay_gather launch (no intermediate
        tensors, no device-to-device copies).  `point_state_batch` is a shape-only placeholder."""
        m = self.memory
        if batch_idx is None:
            batch_idx = m.draw_indices(batch_size, rng)
        batch_idx = np.asarray(batch_idx, dtype=np.int64)
        nxt = m.next_indices(batch_idx)
        end = np.asarray(m.episode_map[batch_idx], dtype=np.int64)
        B = batch_idx.shape[0]
        dev, ev, it = self._indices3(batch_idx, nxt, end)
        return {"replay_gather": self, "idx": dev[0], "nxt": dev[1], "end": dev[2], "ready_event": ev, "_stage_set": it,
                "batch_idx": np.uint8(batch_idx),
                "point_state_batch": _Shape((B,) + tuple(self.point_state.shape[1:])),
                "mask_counts": self._mask_counts(batch_idx)}

    def gather_into(self, lazy, dbuf):
        """fill the runtime's static batch buffers (dict of CUDA float32 tensors) from a sample_lazy() handle.  Ordered on
        the CURRENT stream after the handle's index upload (which ran on the copy stream), whoever the caller is."""
        cur = torch.cuda.current_stream()
        if lazy.get("ready_event") is not None:
            cur.wait_event(lazy["ready_event"])
        a = hip.ReplayGatherArgs()
        a.B = int(lazy["idx"].shape[0])
        a.cloud_elems = int(self.point_state.shape[1] * self.point_state.shape[2])
        a.idx, a.nxt, a.end = hip.ptr(lazy["idx"]), hip.ptr(lazy["nxt"]), hip.ptr(lazy["end"])
        a.point_state = hip.ptr(self.point_state)
        for src in ("action", "expert_action", "goal", "reward", "returns", "terminal", "expert_flags", "perturb_flags"):
            setattr(a, src, hip.ptr(self.rows[src]))
        a.timestep = hip.ptr(self.timestep)
        a.out_point, a.out_next_point = hip.ptr(dbuf["point_state_batch"]), hip.ptr(dbuf["next_point_state_batch"])
        for dst, key in (("out_action", "action_batch"), ("out_expert_action", "expert_action_batch"), ("out_goal", "goal_batch"),
                         ("out_reward", "reward_batch"), ("out_return", "return_batch"), ("out_mask", "mask_batch"),
                         ("out_time", "time_batch"), ("out_time_m1", "time_m1"), ("out_expert_flag", "expert_flag_batch"),
                         ("out_perturb_flag", "perturb_flag_batch")):
            setattr(a, dst, hip.ptr(dbuf[key]))
        hip.call_struct("gad_replay_gather", a)
        it = lazy.get("_stage_set")
        if it is not None and it.get("relabel"):
            self._apply_relabel(it, dbuf["goal_batch"])
        if it is not None:                       # the staging set is not rewritten before this gather has run
            if it["used"] is None:
                it["used"] = torch.cuda.Event()
            it["used"].record(cur)
            it["pending"] = False

    def release(self, lazy):
        """give back the staging set of a sample_lazy() handle that will never be passed to an update step (a prefetcher
        closing with handles in flight, a batch dropped by a validity check): without this the set stays `pending` and
        is never handed out again"""
        it = lazy.get("_stage_set") if isinstance(lazy, dict) else None
        if it is not None:
            it["pending"] = False

    def _mask_counts(self, batch_idx):
        m = self.memory
        ret = np.asarray(m.returns[batch_idx]).reshape(-1)
        exp = np.asarray(m.expert_flags[batch_idx]).reshape(-1)
        per = np.asarray(m.perturb_flags[batch_idx]).reshape(-1)
        reward, expert = ret > 0, exp >= 1
        return np.array([(per < 1).sum(), reward.sum(), expert.sum(), (~(reward & expert)).sum()], dtype=np.float64)

    def sample(self, batch_size, rng=None, batch_idx=None):
        """the update step's 11 arrays as CUDA float32 tensors (runtime.BATCH_KEYS layout) + `batch_idx` and the
        host-side `mask_counts` the data-parallel path all-reduces.  Index semantics: BaseMemory.draw_indices /
        next_indices / post_process_batch (reference core/replay_memory.py:166-176,251-272)."""
        m = self.memory
        if batch_idx is None:
            batch_idx = m.draw_indices(batch_size, rng)
        batch_idx = np.asarray(batch_idx, dtype=np.int64)
        nxt = m.next_indices(batch_idx)
        end = np.asarray(m.episode_map[batch_idx], dtype=np.int64)
        dev, ev, it = self._indices3(batch_idx, nxt, end)
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        d_idx, d_nxt, d_end = dev[0], dev[1], dev[2]
        out = {"point_state_batch": self.point_state.index_select(0, d_idx),
               "next_point_state_batch": self.point_state.index_select(0, d_nxt)}
        for src, dst in _ROW_KEYS:
            out[dst] = self.rows[src].index_select(0, d_idx)
        # remaining steps to the end of the episode (post_process_batch)
        out["time_batch"] = self.timestep.index_select(0, d_end) + 1.0 - self.timestep.index_select(0, d_idx)
        if it.get("relabel"):
            self._apply_relabel(it, out["goal_batch"])
        if it["used"] is None:
            it["used"] = torch.cuda.Event()
        it["used"].record(cur)                   # the index_selects above read the staging set's device block
        it["pending"] = False
        out["batch_idx"] = np.uint8(batch_idx)
        out["mask_counts"] = self._mask_counts(batch_idx)
        return out
