"""Diagnostic: gad_gemm_fwd on the mid-size layer shapes under a kernel-selection option (A/B), with the correctness check
of tools/diag_gemm.py.   python tools/diag_gemm_ab.py fwd_wide"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ga_ddpg_amd import hip
from tools.diag_gemm import bench

opt = sys.argv[1] if len(sys.argv) > 1 else "fwd_wide"
for shape in ((29248, 128, 128), (29248, 128, 256), (8192, 256, 256), (8192, 256, 512), (14000, 128, 256), (4096, 256, 512)):
    for v in (0, 1):
        hip.set_option(opt, v)
        print(opt, v, end="  ")
        bench(*shape, stats=True, bias=False)
