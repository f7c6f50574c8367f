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
    assert L.gad_abi_version() == 11


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
    src = '#include "gaddpg.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(gad_gemm_fwd_args), ' \
          'sizeof(gad_dz_src), sizeof(gad_gemm_dx_args), sizeof(gad_gemm_dw_args), sizeof(gad_replay_gather_args), ' \
          'sizeof(gad_optim_job), sizeof(gad_split_layer), sizeof(gad_copy_seg));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "p.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "p")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(hip.GemmFwdArgs), ctypes.sizeof(hip.DzSrc), ctypes.sizeof(hip.GemmDxArgs),
                     ctypes.sizeof(hip.GemmDwArgs), ctypes.sizeof(hip.ReplayGatherArgs), ctypes.sizeof(hip.OptimJob),
                     ctypes.sizeof(hip.SplitLayer), ctypes.sizeof(hip.CopySeg)]


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
    # furthest point sampling keeps a cloud (and its picks) in one workgroup's LDS: shapes that cannot fit are refused up front
    p = C.c_void_p(0x1000)
    assert L.gad_furthest_point_sampling(p, 1, 64, 65, p, p, null) < 0 and b"picks from" in L.gad_last_error()
    assert L.gad_furthest_point_sampling(p, 1, 16384, 16384, p, p, null) < 0 and b"bytes of LDS" in L.gad_last_error()
    # the library's own default arithmetic is the f32 MFMA (the Python package opts into the split-bf16 form when it loads it)
    assert hip.get_option_default("mfma_split") in (0, 1)


def test_plan_items_are_checked_against_the_entry_points_signature():
    """step replay (include/gaddpg.h section H): a launch item's argument words are checked against the real signature of
    the entry point when the item is added -- count and kind -- so a replay cannot mis-call the ABI.  No GPU needed: nothing
    is enqueued before gad_plan_run."""
    import ctypes as C
    from ga_ddpg_amd import hip, engine
    L = hip.lib()
    names = {L.gad_plan_entry_name(i).decode() for i in range(L.gad_plan_entry_count())}
    # every declared entry point whose last parameter is the stream can be replayed
    src = open(os.path.join(ROOT, "include", "gaddpg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    with_stream = set(re.findall(r"\bint\s+(gad_[a-z0-9_]+)\s*\([^;]*?void\*\s*stream\s*\)\s*;", src))
    assert len(with_stream) >= 40 and with_stream == names, with_stream ^ names
    h = C.c_void_p()
    assert L.gad_plan_create(C.byref(h)) == 0
    # gad_target_noise(pi, u, B, level, normal, out): ptr, ptr, int, float, int, ptr
    argv = hip._args(hip.Ptr(0x1000), hip.Ptr(0x2000), 4, 0.5, 0, hip.Ptr(0x3000))
    words, kinds = engine._pack_words(argv)
    assert kinds == [1, 1, 0, 2, 0, 1]
    W, K = (C.c_uint64 * 6)(*words), (C.c_uint8 * 6)(*kinds)
    assert L.gad_plan_add_call(h, b"gad_target_noise", W, K, 6, 0) == 0
    assert L.gad_plan_add_call(h, b"gad_target_noise", W, K, 5, 0) < 0 and b"takes 6 arguments" in L.gad_last_error()
    K[3] = 3                                                   # a double where the entry point takes a float
    assert L.gad_plan_add_call(h, b"gad_target_noise", W, K, 6, 0) < 0 and b"argument 3" in L.gad_last_error()
    assert L.gad_plan_add_call(h, b"gad_set_option", W, K, 2, 0) < 0 and b"not an entry point" in L.gad_last_error()
    assert L.gad_plan_add_call(h, b"gad_target_noise", W, K, 6, 99) < 0 and b"lane" in L.gad_last_error()
    assert L.gad_plan_add_memcpy(h, C.c_void_p(0x10), C.c_void_p(0x20), C.c_longlong(64), 2) == 1
    assert L.gad_plan_size(h) == 2
    assert L.gad_plan_patch(h, 0, 3, C.c_uint64(engine._float_word(0.25))) == 0
    assert L.gad_plan_patch(h, 0, 6, C.c_uint64(0)) < 0 and b"word 6" in L.gad_last_error()
    assert L.gad_plan_patch(h, 5, 0, C.c_uint64(0)) < 0
    streams = (C.c_void_p * 1)(None)
    assert L.gad_plan_run(h, streams, 1, 0, -1) < 0 and b"lane 2" in L.gad_last_error()      # (refused before anything is enqueued)
    assert L.gad_plan_arm_timing(h, 1, None) < 0 and b"not a launch" in L.gad_last_error()
    assert L.gad_plan_destroy(h) == 0
