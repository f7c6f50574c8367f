"""cycles per part of an FPS arg-max round (wavefront 0 of cloud 0), from the -DGAD_FPS_PHASES build of geometry.hip:
    GAD_LIB_PATH=tools/ubench/libgaddpg_fpsphases.so python tools/diag_fps_phases.py
(build: see tools/README.md).  Parts: 0 read pick's coordinates + distance update, 1 per-thread arg-max scan, 2 wavefront
float max (DPP), 3 ballot / key, 4 LDS write + barrier, 5 read the wavefronts' candidates + 64-bit max + outputs."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_ddpg_amd import hip

L = hip.lib()
L.gad_fps_phase_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for B, N, M in ((128, 4096, 512), (256, 1024, 32)):
    g = torch.Generator(device="cuda").manual_seed(1)
    xyz = torch.rand(B, N, 3, device="cuda", generator=g)
    idx = torch.empty(B, M, dtype=torch.int32, device="cuda")
    nx = torch.empty(B, M, 3, device="cuda")
    hip.call("gad_furthest_point_sampling", xyz, B, N, M, idx, nx)
    torch.cuda.synchronize()
    L.gad_fps_phase_read(None, 1)
    hip.call("gad_furthest_point_sampling", xyz, B, N, M, idx, nx)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    L.gad_fps_phase_read(out, 0)
    v = [int(x) for x in out]
    print("B=%d N=%d M=%d: cycles per round (s_memtime, 100 MHz-domain or shader clock as the part reports it):" % (B, N, M))
    for i, nm in enumerate(("coords + distance update", "thread arg-max scan", "wavefront fmax (DPP)", "ballot / key", "LDS write + barrier",
                            "candidates read + max + outputs")):
        print("   %-34s %8.1f" % (nm, v[i] / (M - 1)))
    print("   %-34s %8.1f" % ("sum", sum(v[:6]) / (M - 1)))
