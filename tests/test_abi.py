"""CPU-side checks of the C-ABI boundary: the library loads and exports every symbol that
include/gaddpg.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gaddpg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gad_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from ga_ddpg_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    L = hip.lib()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "libgaddpg.so does not export " + n
    assert set(names) == set(hip.EXPORTS), set(names) ^ set(hip.EXPORTS)
    assert L.gad_abi_version() == 10


def test_missing_library_fails_loudly(monkeypatch):
    from ga_ddpg_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/libgaddpg.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip.lib()


def test_struct_sizes_match_header():
    """ctypes mirrors must have the C layout: compile a tiny probe with the real header."""
    import ctypes, subprocess, tempfile
    from ga_ddpg_amd import hip
    src = '#include "gaddpg.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(gad_gemm_fwd_args), ' \
          'sizeof(gad_dz_src), sizeof(gad_gemm_dx_args), sizeof(gad_gemm_dw_args), sizeof(gad_replay_gather_args), ' \
          'sizeof(gad_optim_job), sizeof(gad_split_layer));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "p.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "p")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(hip.GemmFwdArgs), ctypes.sizeof(hip.DzSrc), ctypes.sizeof(hip.GemmDxArgs),
                     ctypes.sizeof(hip.GemmDwArgs), ctypes.sizeof(hip.ReplayGatherArgs), ctypes.sizeof(hip.OptimJob),
                     ctypes.sizeof(hip.SplitLayer)]


def test_argument_errors_are_status_codes_not_crashes():
    """error behaviour at the boundary (int return code, no exceptions across the ABI, message through
    gad_last_error): bad arguments are rejected by the host-side checks before anything is launched -- so this runs
    without a GPU -- and the Python binding turns the status into a RuntimeError carrying the message, as the
    upstream extension's TORCH_CHECKs do."""
    import ctypes as C
    from ga_ddpg_amd import hip
    L = hip.lib()
    null = C.c_void_p(None)
    assert L.gad_furthest_point_sampling(null, 1, 8, 4, null, null, null) < 0
    assert b"null" in L.gad_last_error().lower()
    assert L.gad_gemm_fwd(null, null) < 0
    a = hip.GemmFwdArgs()
    a.W, a.zout, a.n_groups, a.Kp = 1, 1, 7, 8                 # non-null dummies, bad group count: rejected before any use
    rc = L.gad_gemm_fwd(C.byref(a), null)
    assert rc < 0 and b"n_groups" in L.gad_last_error()
    a.n_groups, a.Kp = 1, 12                                   # Kp must be a multiple of 8
    assert L.gad_gemm_fwd(C.byref(a), null) < 0 and b"Kp" in L.gad_last_error()
    assert L.gad_set_option(b"no_such_option", 1) < 0 and b"unknown option" in L.gad_last_error()
    assert L.gad_set_option(b"fwd_stream", 1) == 0
    g = hip.ReplayGatherArgs()
    assert L.gad_replay_gather(C.byref(g), null) < 0
    with pytest.raises(RuntimeError, match="gad_gemm_dx failed"):
        hip.check(L.gad_gemm_dx(C.byref(hip.GemmDxArgs()), null), "gad_gemm_dx")
    # per-stream wavefront priority (ABI 10): a table entry, no launch -- range-checked, removable; the timing slot must be 8-byte
    # aligned because the priority rides in the low bits of the pointer the kernels receive
    st = C.c_void_p(0x1230)
    assert L.gad_stream_priority(st, 7) < 0 and b"priority" in L.gad_last_error()
    assert L.gad_stream_priority(st, 2) == 0 and L.gad_stream_priority(st, 0) == 0
    assert L.gad_timing_slot(C.c_void_p(0x1004)) < 0 and b"aligned" in L.gad_last_error()
    assert L.gad_timing_slot(null) == 0
    assert L.gad_set_option(b"skinny_nw", 4) == 0 and L.gad_set_option(b"skinny_nw", 8) == 0
