"""Diagnostic (not a test): micro-benchmark of gad_gemm_fwd (ACT input mode) on the step's layer shapes.
    python -m tools.diag_gemm"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ga_ddpg_amd import hip
from ga_ddpg_amd.engine import _fwd_args, _ptr


def bench(rows, K, n, stats=True, iters=50, bias=False, check=True):
    dev = torch.device("cuda:0")
    Kp = (K + (1 if bias else 0) + 7) // 8 * 8
    zin = torch.randn(rows, K, device=dev)
    W = torch.randn(n, Kp, device=dev) * 0.05
    scale = torch.rand(K, device=dev) + 0.5
    shift = torch.randn(K, device=dev) * 0.1
    zout = torch.empty(rows, n, device=dev)
    st = torch.zeros(hip.STAT_REPLICAS * 2 * n, dtype=torch.float64, device=dev)
    a = _fwd_args(mode=0, zin=_ptr(zin), zin_pitch=K, c_in=K, scale=_ptr(scale), shift=_ptr(shift), relu=1,
                  n_rows=rows, W=_ptr(W), Kp=Kp, n_out=[n], zout=_ptr(zout), zout_pitch=n, ones_col=K if bias else -1,
                  stat_sum=_ptr(st, 0, 8) if stats else None, stat_sq=_ptr(st, n, 8) if stats else None, stat_stride=2 * n)
    for _ in range(5):
        hip.call_struct("gad_gemm_fwd", a)
    if check:
        st.zero_()
        zout.fill_(-7.0)
        hip.call_struct("gad_gemm_fwd", a)
        xin = torch.relu(zin.double() * scale.double() + shift.double())
        ref = xin @ W[:, :K].double().t()
        if bias:
            ref = ref + W[:, K].double()
        err = (zout.double() - ref).abs()
        bad = (err.max(1).values > 1e-3).nonzero().flatten()
        if bad.numel():
            r = int(bad[0])
            cols = (err[r] > 1e-3).nonzero().flatten().tolist()
            print("   bad rows %d: %s ; row %d bad cols %s" % (bad.numel(), bad[:12].tolist(), r, cols[:8] + ["..."] + cols[-3:]))
            # does the row hold another row's values?
            other = ((ref - zout[r].double()).abs().max(1).values < 1e-4).nonzero().flatten().tolist()
            print("   row %d holds the values of rows %s" % (r, other[:5]))
        ssum = st.view(hip.STAT_REPLICAS, 2, n).sum(0)
        print("   max |z - ref| %.3e (max |ref| %.3e) at row %d; stat sum err %.3e sq err %.3e" % (
            err.max(), ref.abs().max(), int(err.max(1).values.argmax()), (ssum[0] - ref.sum(0)).abs().max() / ref.sum(0).abs().max(),
            (ssum[1] - (ref * ref).sum(0)).abs().max() / (ref * ref).sum(0).max()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        hip.call_struct("gad_gemm_fwd", a)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    gf = 2.0 * rows * Kp * n / us * 1e-3
    gb = 4.0 * rows * (K + n) / us * 1e-3
    print("rows %7d K %4d n %4d stats %d : %7.1f us  %6.1f GFLOP/s-k  %6.1f GB/s" % (rows, K, n, stats, us, gf, gb))


if __name__ == "__main__":
    for shape in ((224000, 64, 64), (224000, 64, 128), (223990, 64, 128), (29248, 128, 128), (29248, 128, 256), (8192, 256, 256),
                  (8192, 256, 512), (256, 512, 1024), (256, 1024, 512), (256, 512, 256), (256, 256, 256)):
        bench(*shape, stats=True, bias=shape[0] == 256)
