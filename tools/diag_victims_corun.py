"""victims (tools/ubench/victims.hip) beside GEMM launches of one family: python tools/diag_victims_corun.py [split|f32] [family]
family: fwd_stream (default) | fwd | dx | dw"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_ddpg_amd import hip
from tests import split_cases as sc

other = sys.argv[1] if len(sys.argv) > 1 else "split"
fam = sys.argv[2] if len(sys.argv) > 2 else "fwd_stream"
V = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", os.environ.get("VICTIMS_LIB", "libvictims.so")))
V.victim_mix_launch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
case = {"fwd_stream": lambda: sc.FwdStream(213034, 64, 64, "act"), "fwd": lambda: sc.FwdWide(27240, 128, 128, "act"),
        "dx": lambda: sc.DxWide(27240, 128, 128, "act"), "dw": lambda: sc.DwWide(27240, 128, 128, "act")}[fam]()
hip.set_option("mfma_split", 0 if other == "f32" else 1)
a = case.args()
f = getattr(hip.lib(), case.entry)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for red, nm in ((0, "DPP ladder with row_bcast"), (1, "no DPP"), (2, "DPP row_shr + 4 readlanes")):
    flags = torch.zeros(1, dtype=torch.int32, device="cuda")
    detail = torch.zeros(4, dtype=torch.int32, device="cuda")
    for rep in range(6):
        with torch.cuda.stream(s2):
            for _ in range(300):
                hip.check(f(C.byref(a), hip.stream()), "gemm")
        with torch.cuda.stream(s1):
            for _ in range(200):
                V.victim_mix_launch(32, 400, flags.data_ptr(), detail.data_ptr(), hip.stream(), red)
        torch.cuda.synchronize()
    print("victim_mix (%s) beside %s %s: flags %s (1 wave max, 2 bystander registers, 4 packed != scalar, 8 ballot/readlane, 16 LDS word, 32 packed path != closed form, 64 scalar path != closed form), threads in error by 16-lane row %s"
          % (nm, other, fam, bin(int(flags.item())), detail.tolist()))
sys.exit(0)
V.victim_pk_launch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
counts = torch.zeros(24, dtype=torch.int32, device="cuda")
sample = torch.zeros(8, device="cuda")
for rep in range(6):
    with torch.cuda.stream(s2):
        for _ in range(300):
            hip.check(f(C.byref(a), hip.stream()), "gemm")
    with torch.cuda.stream(s1):
        for _ in range(200):
            V.victim_pk_launch(32, 400, counts.data_ptr(), sample.data_ptr(), hip.stream())
    torch.cuda.synchronize()
c = counts.view(3, 2, 4).tolist()
print("victim_pk beside %s %s: mismatches vs the scalar instruction [lo half rows 0-3 | hi half rows 0-3]: add %s mul %s fma %s" % (other, fam, c[0], c[1], c[2]))
print("   first add mismatch: packed %r scalar %r of %r + %r; first mul mismatch: packed %r scalar %r of %r * %r" % tuple(sample.tolist()))

V.victim_ldsuse_launch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
counts = torch.zeros(20, dtype=torch.int32, device="cuda")
for rep in range(6):
    with torch.cuda.stream(s2):
        for _ in range(300):
            hip.check(f(C.byref(a), hip.stream()), "gemm")
    with torch.cuda.stream(s1):
        for _ in range(200):
            V.victim_ldsuse_launch(32, 400, counts.data_ptr(), hip.stream())
    torch.cuda.synchronize()
c = counts.view(5, 4).tolist()
print("victim_ldsuse beside %s %s, wrong results by 16-lane row: uniform->v_add %s | per-lane->v_add %s | uniform, s_nop 7, v_add %s | read2->v_pk_add op_sel %s | uniform->v_mov->v_add %s"
      % (other, fam, c[0], c[1], c[2], c[3], c[4]))

V.victim_ldsuse2_launch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
counts = torch.zeros(20, dtype=torch.int32, device="cuda")
for rep in range(6):
    with torch.cuda.stream(s2):
        for _ in range(300):
            hip.check(f(C.byref(a), hip.stream()), "gemm")
    with torch.cuda.stream(s1):
        for _ in range(200):
            V.victim_ldsuse2_launch(32, 400, counts.data_ptr(), hip.stream())
    torch.cuda.synchronize()
c = counts.view(5, 4).tolist()
print("victim_ldsuse2 beside %s %s, wrong by 16-lane row: dest=addr -> pk_add %s | separate addr %s | + 8 VALU before the wait %s | + s_nop 3 after it %s | dest=addr -> v_sub %s"
      % (other, fam, c[0], c[1], c[2], c[3], c[4]))
