"""configs[3] fused SA stack (bench.sa_kernel_mfma) alone, for a kernel trace: python tools/diag_sa_op.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

if __name__ == "__main__":
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    r = bench.sa_kernel_mfma(iters=it)
    print({k: r[k] for k in ("ms_fwd_bwd", "achieved", "frac", "live_rows")})
