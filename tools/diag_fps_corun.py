"""Is a geometry kernel's result independent of what else runs on the GPU?  FPS (and ball query) launched repeatedly on one
stream while another stream runs wide-tile GEMM launches (split-bf16 or FP32-MFMA form); every output compared with the result
of the same launch alone.   python tools/diag_fps_corun.py [iters] [split|f32|none]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_ddpg_amd import hip
from tests import split_cases as sc

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
other = sys.argv[2] if len(sys.argv) > 2 else "split"
B, N, M = 32, 1024, 128
g = torch.Generator(device="cuda").manual_seed(1)
xyz = torch.rand(B, N, 3, device="cuda", generator=g)
ref = torch.empty(B, M, dtype=torch.int32, device="cuda")
refx = torch.empty(B, M, 3, device="cuda")
hip.call("gad_furthest_point_sampling", xyz, B, N, M, ref, refx)
torch.cuda.synchronize()
R = 256
outs = torch.zeros(R, B, M, dtype=torch.int32, device="cuda")
outx = torch.zeros(R, B, M, 3, device="cuda")
case = sc.FwdWide(27240, 128, 128, "act")
dxc = sc.DxWide(27240, 128, 128, "act")
hip.set_option("mfma_split", 0 if other == "f32" else 1)
if os.environ.get("FPS_CFG"):
    hip.set_option("fps_cfg", int(os.environ["FPS_CFG"]))
a, ad = case.args(), dxc.args()
f, fd = hip.lib().gad_gemm_fwd, hip.lib().gad_gemm_dx
EXTRA = {"dw": lambda: sc.DwWide(27240, 128, 128, "act"), "bwd_stream": lambda: sc.BwdStream(213034, 64, "act"),
         "fwd_stream": lambda: sc.FwdStream(213034, 64, 64, "act"), "fwd_pool": lambda: sc.FwdWide(27240, 128, 256, "pool"),
         "dx_pool": lambda: sc.DxWide(27240, 256, 128, "pool")}
if os.environ.get("CORUN") in EXTRA:
    EXTRA_CASE = EXTRA[os.environ["CORUN"]]()
    if isinstance(EXTRA_CASE, sc.BwdStream):
        KEEP = (EXTRA_CASE.dx.args(), EXTRA_CASE.dw_args())
    else:
        KEEP = (EXTRA_CASE.args(),)
    EXTRA_ARGS = tuple(C.byref(x) for x in KEEP)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad = 0
hist_lane, hist_s, hist_wave = [0] * 64, [0] * 8, [0] * 16
for it in range(0, iters, R):
    with torch.cuda.stream(s2):
        if other != "none":
            for _ in range(R * 2):
                if os.environ.get("CORUN", "both") in EXTRA:
                    hip.check(getattr(hip.lib(), EXTRA_CASE.entry)(*EXTRA_ARGS, hip.stream()), "extra")
                if os.environ.get("CORUN", "both") in ("both", "fwd"):
                    hip.check(f(C.byref(a), hip.stream()), "fwd")
                if os.environ.get("CORUN", "both") in ("both", "dx"):
                    hip.check(fd(C.byref(ad), hip.stream()), "dx")
    with torch.cuda.stream(s1):
        for i in range(R):
            hip.call("gad_furthest_point_sampling", xyz, B, N, M, outs[i], outx[i])
    torch.cuda.synchronize()
    wrong = (outs != ref[None]).flatten(1).any(1)
    wx = (outx != refx[None]).flatten(1).any(1)
    bad += int(wrong.sum()) + int((wx & ~wrong).sum())
    if int(wrong.sum()):
        print("   round at %d: wrong launch offsets %s" % (it, torch.nonzero(wrong).flatten().tolist()[:24]))
        for i in torch.nonzero(wrong).flatten().tolist():
            for c in torch.nonzero((outs[i] != ref).any(1)).flatten().tolist():
                j0 = int(torch.nonzero(outs[i][c] != ref[c])[0])
                k = int(outs[i][c][j0])
                hist_lane[k & 63] += 1; hist_s[k >> 8] += 1; hist_wave[(k & 255) >> 6] += 1
        for i in torch.nonzero(wrong).flatten().tolist()[:0]:
            cl = torch.nonzero((outs[i] != ref).any(1)).flatten().tolist()
            msg = []
            for c in cl[:4]:
                j0 = int(torch.nonzero(outs[i][c] != ref[c])[0])
                gath = xyz[c][outs[i][c].long()]                       # coordinates of the picks this launch reported
                lds_ok = bool((gath == outx[i][c]).all())
                nbad = int((gath != outx[i][c]).any(1).sum())
                msg.append("cloud %d from pick %d (%d -> %d), new_xyz == xyz[idx]: %s (%d rows differ)" % (c, j0, int(ref[c][j0]), int(outs[i][c][j0]), lds_ok, nbad))
            print("      launch +%d: %s" % (i, "; ".join(msg)))
        i = int(torch.nonzero(wrong)[0])
        d = torch.nonzero(outs[i] != ref)
        nc = int((outs[i] != ref).any(1).sum())
        print("   launch %d: %d clouds wrong;" % (it + i, nc), end="")
        print("   launch %d: %d picks differ, first at cloud %d pick %d: %d vs %d" % (it + i, len(d), int(d[0, 0]), int(d[0, 1]),
                                                                                  int(outs[i][d[0, 0], d[0, 1]]), int(ref[d[0, 0], d[0, 1]])))
print("first wrong pick k = s * 256 + tid: lanes", hist_lane, "s", hist_s[:4], "wave", hist_wave[:4])
print("FPS beside %s wide GEMMs: %d of %d launches differ from the launch alone" % (other, bad, iters))
