"""Micro-benchmark of gad_gemm_bwd on one mid-size layer shape (synthetic operands), fused wide kernel vs the separate
dX + dW kernels.  (Round-4 ablation of the first fused version on the SA2 layer-2 shape, profiles/r04_bwd_wide_ablation.txt: 32.1 us as built,
24.8 / 24.4 without the dX / dW MFMAs, 18.0 without both, 12.6 us for the empty shell -- the MFMAs overlapped with nothing: the
loop carried 7 vector instructions per MFMA.)

    python tools/ubench_bwd_wide.py [rows N K pooled]      default: 27240 128 128 0   (SA2 layer 2 at B = 256)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ga_ddpg_amd import hip
from ga_ddpg_amd.engine import _dz, _fwd_args, _ptr


def main():
    rows, N, K, pooled = (int(x) for x in (sys.argv[1:5] + ["27240", "128", "128", "0"][len(sys.argv) - 1:]))
    dev = torch.device("cuda:0")
    cap = rows + 4096
    g = torch.Generator(device="cuda").manual_seed(1)
    z = torch.randn(cap, N, device=dev, generator=g)
    G = torch.randn(cap, N, device=dev, generator=g)
    zprev = torch.randn(cap, K, device=dev, generator=g)
    gout = torch.empty(cap, K, device=dev)
    W = torch.randn(N, K, device=dev, generator=g) * 0.05
    vecN = [torch.rand(N, device=dev, generator=g) + 0.5 for _ in range(5)]
    vecK = [torch.rand(K, device=dev, generator=g) + 0.5 for _ in range(4)]
    row_w = torch.ones(cap, device=dev)
    nrows = torch.tensor([rows], dtype=torch.int32, device=dev)
    bst = torch.zeros(hip.STAT_REPLICAS * 2 * K, dtype=torch.float64, device=dev)
    gacc = torch.zeros(N * K, dtype=torch.float64, device=dev)
    ws = torch.empty(9 * 1024 * 1024, device=dev)
    n_grp = rows // 4
    row_grp = (torch.arange(cap, device=dev, dtype=torch.int32) // 4).clamp_(max=n_grp - 1)
    argmax = (torch.arange(n_grp, device=dev, dtype=torch.int32) * 4)[:, None].repeat(1, N).contiguous()
    dout = torch.randn(n_grp, N, device=dev, generator=g)
    dkw = dict(z=_ptr(z), z_pitch=N, scale=_ptr(vecN[0]), shift=_ptr(vecN[1]), relu=1, premasked=1, row_w=_ptr(row_w), c=N,
               coefP=_ptr(vecN[2]), coefQ=_ptr(vecN[3]), coefS=_ptr(vecN[4]))
    if pooled:
        dkw.update(gmode=1, argmax=_ptr(argmax), dout=_ptr(dout), row_grp=_ptr(row_grp))
    else:
        dkw.update(gmode=0, G=_ptr(G), g_pitch=N)
    ax = hip.GemmDxArgs()
    ax.n_rows_dev, ax.n_rows, ax.dz, ax.n_groups = _ptr(nrows), cap, _dz(**dkw), 1
    ax.n_out[0] = N
    ax.W, ax.Kp, ax.k_valid, ax.grp_per_sample, ax.epilogue = _ptr(W), K, K, 1, 0
    ax.gout, ax.gout_pitch, ax.zprev, ax.zprev_pitch = _ptr(gout), K, _ptr(zprev), K
    ax.prev_scale, ax.prev_shift, ax.prev_mean, ax.prev_istd = (_ptr(v) for v in vecK)
    ax.prev_dbeta, ax.prev_dgamma, ax.stat_stride, ax.store_masked = _ptr(bst, 0, 8), _ptr(bst, K, 8), 2 * K, 1
    aw = hip.GemmDwArgs()
    aw.inp = _fwd_args(mode=0, zin=_ptr(zprev), zin_pitch=K, c_in=K, scale=_ptr(vecK[0]), shift=_ptr(vecK[1]), relu=1,
                       n_rows_dev=_ptr(nrows), n_rows=cap, row_w=_ptr(row_w), Kp=K, n_out=[N], w_off=[0])
    aw.dz, aw.gacc, aw.partial, aw.partial_elems = ax.dz, _ptr(gacc), _ptr(ws), ws.numel()
    L = hip.lib()

    def run(iters=50):
        st = hip.stream()
        for _ in range(5):
            hip.check(L.gad_gemm_bwd(C.byref(ax), C.byref(aw), st), "bwd")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            hip.check(L.gad_gemm_bwd(C.byref(ax), C.byref(aw), st), "bwd")
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    flops = 4.0 * rows * N * K
    for name, opts in (("separate dX + dW (+ reduce)", {"bwd_wide": 0}), ("fused (+ reduce)", {"bwd_wide": 1})):
        for k, v in opts.items():
            hip.set_option(k, v)
        us = run()
        print("%-32s %7.1f us  %5.1f TFLOP/s" % (name, us, flops / us * 1e-6))
    aw.row_splits = -2
    print("%-32s %7.1f us" % ("fused, reduce left to the caller", run()))


if __name__ == "__main__":
    main()
