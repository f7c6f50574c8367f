"""CPU-side check of the RCCL ctypes binding (ga_ddpg_amd/rccl.py): librccl loads and exports the entry points the
data-parallel step calls on its own streams.  No communicator is created here (that needs a GPU: tests/test_gpu_dp.py)."""


def test_librccl_loads_and_exports_the_entry_points():
    from ga_ddpg_amd import rccl
    L = rccl.lib()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclBroadcast", "ncclCommDestroy", "ncclGetErrorString"):
        assert hasattr(L, name), name
    assert rccl.lib() is L                              # one library instance per process
    assert rccl._DTYPES[__import__("torch").float32] == 7 and rccl.NCCL_SUM == 0      # ncclFloat32 / ncclSum of nccl.h
