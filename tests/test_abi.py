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
    assert L.gad_abi_version() == 2


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
    src = '#include "gaddpg.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu\\n", sizeof(gad_gemm_fwd_args), ' \
          'sizeof(gad_dz_src), sizeof(gad_gemm_dx_args), sizeof(gad_gemm_dw_args));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "p.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "p")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(hip.GemmFwdArgs), ctypes.sizeof(hip.DzSrc), ctypes.sizeof(hip.GemmDxArgs),
                     ctypes.sizeof(hip.GemmDwArgs)]
