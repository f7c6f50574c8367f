"""Sample prefetch for the update loop (BASELINE configs[4]: "overlapped sample-prefetch and all-reduce").

The reference's loop samples synchronously between two updates (core/train_test_offline.py:119): 17 MB of float64 fancy
indexing per minibatch (core/replay_memory.py:109-127) while the GPU idles.  `PrefetchSampler` moves that to a background
thread: it keeps `depth` minibatches ready in PINNED float32 staging sets, so `update_parameters` only issues the
host-to-device copies (FusedRuntime.upload takes dicts of pinned tensors as they are), and the gather of batch k+1 overlaps
the update on batch k.  Batches come out in exactly the order -- and with exactly the contents -- a synchronous
`memory.sample(batch_size)` loop on the same random stream would give (one producer, FIFO queue):
tests/test_prefetch.py.

    sampler = PrefetchSampler(memory, batch_size=512, rng=np.random.default_rng(seed))
    for i in range(n):
        agent.update_parameters(sampler.next(), agent.update_step, i)
    sampler.close()

With a DeviceReplay (GPU-resident buffer) the handles of `sample_lazy` are prefetched instead (index vectors only).
"""
import queue
import threading

import numpy as np
import torch

KEYS = ("point_state_batch", "next_point_state_batch", "action_batch", "expert_action_batch", "reward_batch",
        "return_batch", "mask_batch", "time_batch", "goal_batch", "expert_flag_batch", "perturb_flag_batch")


class PrefetchSampler(object):
    def __init__(self, memory, batch_size, depth=2, rng=None, pin=None, sample=None, threads=4):
        """memory: BaseMemory (host sampling into pinned staging sets) or DeviceReplay (lazy gather handles).
        sample: optional callable(batch_size) -> batch dict replacing memory.sample (e.g. a validity-checking sampler); if it
        accepts `clouds_out` / `pool` keywords (BaseMemory.sample's, synth_data.sample_valid_batch's) the two cloud gathers --
        97 % of the bytes -- are written straight into the pinned staging set by `threads` pool threads (np.take releases the
        GIL) instead of being fancy-indexed into a temporary on this one thread and copied again."""
        self.memory, self.batch_size, self.depth = memory, int(batch_size), int(depth)
        self.rng = rng
        self.lazy = hasattr(memory, "sample_lazy")
        self.pin = torch.cuda.is_available() if pin is None else bool(pin)
        self._sample = sample
        self._pool = None
        self._direct = False
        if not self.lazy and hasattr(memory, "gather_clouds"):
            import inspect
            fn = sample if sample is not None else memory.sample
            try:
                self._direct = "clouds_out" in inspect.signature(fn).parameters
            except (TypeError, ValueError):
                self._direct = False
            if self._direct and threads > 1:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=int(threads), thread_name_prefix="gad-gather")
        self._free = queue.Queue()
        self._ready = queue.Queue(maxsize=self.depth)
        self._sets = []
        self._held = self._held_item = None
        self._stop = threading.Event()
        self._error = None
        self._thread = threading.Thread(target=self._run, name="gad-prefetch", daemon=True)
        self._thread.start()

    # ------------------------------------------------------------------ producer
    def _draw(self, st=None):
        kw = {}
        if st is not None:                       # direct mode: the clouds land in the staging set's own buffers
            kw = dict(clouds_out=(st["point_state_batch"].numpy(), st["next_point_state_batch"].numpy()), pool=self._pool)
        if self._sample is not None:
            return self._sample(self.batch_size, **kw)
        if self.lazy:
            return self.memory.sample_lazy(self.batch_size, self.rng)
        try:
            return self.memory.sample(self.batch_size, rng=self.rng, **kw)
        except TypeError:
            return self.memory.sample(batch_size=self.batch_size)

    def _staging(self, batch):
        """one pinned float32 staging set shaped like `batch` (allocated on first use: depth + 1 sets circulate)"""
        st = {}
        for k in KEYS:
            if k in batch:
                t = torch.empty(tuple(np.asarray(batch[k]).shape), dtype=torch.float32)
                st[k] = t.pin_memory() if self.pin else t
        return st

    def _cloud_staging(self):
        """direct mode: the set's two cloud buffers exist BEFORE the draw (their shape is the buffer's); the small arrays
        are added by the first batch that passes through the set"""
        shape = (self.batch_size,) + tuple(self.memory.point_state.shape[1:])
        st = {}
        for k in ("point_state_batch", "next_point_state_batch"):
            t = torch.empty(shape, dtype=torch.float32)
            st[k] = t.pin_memory() if self.pin else t
        return st

    def _acquire(self, make):
        """a free staging set: one that came back from the consumer (after its upload event), a new one while fewer than
        depth + 1 exist, else wait for one; None when the sampler is closing"""
        try:
            return self._release(self._free.get_nowait())
        except queue.Empty:
            pass
        if len(self._sets) < self.depth + 1:
            st = make()
            self._sets.append(st)
            return st
        while not self._stop.is_set():
            try:
                return self._release(self._free.get(timeout=0.05))
            except queue.Empty:
                pass
        return None

    @staticmethod
    def _release(entry):
        """a staging set comes back with the event after which its host-to-device copies are done (the runtime hangs it on
        the batch dict as `uploaded_event`; a run-ahead consumer may return the set before those copies have run)"""
        st, ev = entry
        if ev is not None:
            ev.synchronize()
        return st

    def _run(self):
        try:
            while not self._stop.is_set():
                if self.lazy:
                    item = self._draw()
                else:
                    if self._direct:
                        st = self._acquire(self._cloud_staging)
                        if st is None:
                            return
                        batch = self._draw(st)
                    else:
                        batch = self._draw()
                        st = self._acquire(lambda: self._staging(batch))
                        if st is None:
                            return
                    for k in KEYS:
                        if k not in batch:
                            continue
                        if k not in st:                             # (direct mode: the small arrays, first pass through the set)
                            t = torch.empty(tuple(np.asarray(batch[k]).shape), dtype=torch.float32)
                            st[k] = t.pin_memory() if self.pin else t
                        dst = st[k].numpy()
                        src = np.asarray(batch[k])
                        if not np.shares_memory(dst, src):          # (direct mode: the clouds are already in place)
                            np.copyto(dst, src.reshape(dst.shape), casting="same_kind")
                    item = dict(st)
                    for k, v in batch.items():                      # bookkeeping fields (indices, counts) ride along
                        if k not in item and k not in ("image_state_batch", "next_image_state_batch"):
                            item[k] = v
                    item["_staging"] = st
                    item["uploaded_event"] = None                   # the runtime puts the upload's completion event here
                while not self._stop.is_set():
                    try:
                        self._ready.put(item, timeout=0.05)
                        item = None
                        break
                    except queue.Full:
                        pass
                if item is not None:
                    self._drop(item)                                # closing with a drawn batch in hand
        except BaseException as e:                                  # surfaced by next()
            self._error = e
            self._ready.put(None)

    def _drop(self, item):
        """a prefetched batch that will never be consumed: a DeviceReplay handle gives its staging set back"""
        if self.lazy and isinstance(item, dict) and hasattr(self.memory, "release"):
            self.memory.release(item)

    # ------------------------------------------------------------------ consumer
    def next(self):
        """the next minibatch; the staging set handed out by the PREVIOUS call returns to the pool together with the event
        that marks its host-to-device copies done (the producer waits for it before overwriting the set)"""
        if self._held is not None:
            self._free.put((self._held, self._held_item.get("uploaded_event")))
            self._held = self._held_item = None
        item = self._ready.get()
        if item is None:
            raise RuntimeError("prefetch thread failed") from self._error
        self._held = item.pop("_staging", None) if isinstance(item, dict) else None
        self._held_item = item if self._held is not None else None
        return item

    __call__ = lambda self, batch_size=None: self.next()           # drop-in for memory.sample(batch_size=...)

    def close(self):
        self._stop.set()
        for _ in range(2):                       # before and after the producer has let go of what it was holding
            try:
                while True:
                    item = self._ready.get_nowait()
                    if item is not None:
                        self._drop(item)
            except queue.Empty:
                pass
            self._thread.join(timeout=2.0)
        if self._pool is not None:
            self._pool.shutdown(wait=False)
            self._pool = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
