"""RCCL through its C API (ctypes), for collectives that must run ON A GIVEN HIP STREAM.

torch.distributed's 'nccl' backend (= RCCL on ROCm) issues every collective on a stream of its own and fences it against
the caller's stream with events.  This process already keeps its four hardware queues busy (engine._PHYS); a fifth active
stream halves the step rate (measured, profiles/README.md), which is what the asynchronous bucketed gradient exchange cost
in round 2 (-9 % at one rank).  ncclAllReduce takes the stream as an argument: issued on the weight-gradient lane the
bucket was produced on, the exchange is just one more kernel in that lane's queue.

The communicator is bootstrapped over an initialised torch.distributed group (any backend): rank 0 draws the
ncclUniqueId, broadcast_object_list hands it round.  One process per GPU; the device is the caller's current one.
Reference counterpart: nn.DataParallel's replicate / gather over NCCL inside one process (core/utils.py:202).
"""
import ctypes as C
import os

import torch

_lib = None
NCCL_SUM = 0
_DTYPES = {torch.float32: 7, torch.float64: 8, torch.int32: 2, torch.int64: 4}


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * 128)]


def lib():
    """librccl.so: the copy torch itself loads when it is there (one RCCL per process), else ROCm's"""
    global _lib
    if _lib is None:
        cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so", "librccl.so"]
        err = None
        for p in cands:
            try:
                _lib = C.CDLL(p)
                break
            except OSError as e:
                err = e
        if _lib is None:
            raise OSError("librccl.so not found (%s)" % err)
        L = _lib
        L.ncclGetErrorString.restype = C.c_char_p
        L.ncclGetErrorString.argtypes = [C.c_int]
        L.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclBroadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        L.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, lib().ncclGetErrorString(rc).decode()))


def unique_id():
    """a fresh ncclUniqueId as 128 bytes (drawn by ONE rank and handed to the others over the bootstrap group)"""
    uid = _UniqueId()
    _check(lib().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    return bytes(bytearray(uid.internal))


class Communicator(object):
    """one RCCL communicator over the ranks of `group` (default: the world group).  uid: the 128 bytes of an ncclUniqueId every
    rank already holds (parallel.DataParallelContext exchanges them itself, so that no rank can skip the exchange); None: rank 0
    draws one and broadcasts it here."""

    def __init__(self, group=None, uid=None):
        import torch.distributed as dist
        L = lib()
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if uid is None:
            box = [unique_id() if self.rank == 0 else None]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            uid = box[0]
        raw, uid = uid, _UniqueId()
        C.memmove(C.byref(uid), raw, 128)
        self.device = torch.cuda.current_device()
        self._comm = C.c_void_p()
        _check(L.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")

    def all_reduce_(self, t, stream=None):
        """in-place SUM of a contiguous CUDA tensor over the ranks, enqueued on `stream` (default: the current stream);
        returns at once -- ordered like any kernel of that stream"""
        assert t.is_cuda and t.is_contiguous() and t.dtype in _DTYPES, "rccl.all_reduce_: contiguous CUDA f32/f64/i32/i64 tensor"
        s = torch.cuda.current_stream(t.device) if stream is None else stream
        _check(lib().ncclAllReduce(C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), t.numel(), _DTYPES[t.dtype], NCCL_SUM,
                                   self._comm, C.c_void_p(s.cuda_stream)), "ncclAllReduce")

    def broadcast_(self, t, root=0, stream=None):
        assert t.is_cuda and t.is_contiguous() and t.dtype in _DTYPES
        s = torch.cuda.current_stream(t.device) if stream is None else stream
        _check(lib().ncclBroadcast(C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), t.numel(), _DTYPES[t.dtype], root,
                                   self._comm, C.c_void_p(s.cuda_stream)), "ncclBroadcast")

    def count(self):
        """ranks RCCL itself reports for this communicator (ncclCommCount): what bench.py prints as config.rccl_nranks"""
        n = C.c_int(-1)
        _check(lib().ncclCommCount(self._comm, C.byref(n)), "ncclCommCount")
        return int(n.value)

    def self_test(self):
        """one 64-float all-reduce on the current stream whose answer is known (sum over ranks of rank + 1); True if this
        rank saw it.  Run once right after the communicator is built: a transport that initialises but cannot move data is
        found here, not inside the first update step."""
        t = torch.full((64,), float(self.rank + 1), dtype=torch.float32, device="cuda:%d" % self.device)
        self.all_reduce_(t)
        torch.cuda.synchronize(self.device)
        want = self.world * (self.world + 1) / 2.0
        return bool((t == want).all().item())

    def destroy(self):
        if self._comm:
            try:
                torch.cuda.synchronize(self.device)
                lib().ncclCommDestroy(self._comm)
            finally:
                self._comm = C.c_void_p()
