"""Do two independent launch chains overlap on the GPU?  N back-to-back forward GEMM launches of one layer shape on ONE stream
against the same N launches on each of TWO streams (separate buffers).  ratio = t(two streams) / t(one stream): 1.0 = the
second chain rides for free, 2.0 = the two chains take turns (every launch fills the CUs' register file / LDS, so a launch
of the other chain cannot become resident before workgroups retire).

    python tools/ubench_overlap.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ga_ddpg_amd import hip
from ga_ddpg_amd.engine import _fwd_args, _ptr


def layer(rows, K, n, dev):
    zin = torch.randn(rows, K, device=dev)
    W = torch.randn(n, K, device=dev) * 0.05
    scale = torch.rand(K, device=dev) + 0.5
    shift = torch.randn(K, device=dev) * 0.1
    zout = torch.empty(rows, n, device=dev)
    st = torch.zeros(hip.STAT_REPLICAS * 2 * n, dtype=torch.float64, device=dev)
    a = _fwd_args(mode=0, zin=_ptr(zin), zin_pitch=K, c_in=K, scale=_ptr(scale), shift=_ptr(shift), relu=1, n_rows=rows, W=_ptr(W),
                  Kp=K, n_out=[n], zout=_ptr(zout), zout_pitch=n, ones_col=-1, stat_sum=_ptr(st, 0, 8), stat_sq=_ptr(st, n, 8),
                  stat_stride=2 * n)
    a._keep = (zin, W, scale, shift, zout, st)
    return a


def main():
    import ctypes as C
    dev = torch.device("cuda:0")
    L = hip.lib()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    n = 60
    for name, (rows, K, N) in (("SA2 layer 2 (wide)", (27240, 128, 128)), ("SA2 layer 3 shape, no pool (wide)", (27240, 128, 256)),
                               ("SA3 layer 2 (wide)", (8192, 256, 256)), ("SA1 layer 2 (streaming)", (213034, 64, 64)),
                               ("SA1 layer 3 shape, no pool (streaming)", (213034, 64, 128)), ("FC 2 (skinny)", (256, 1024, 512))):
        a, b = layer(rows, K, N, dev), layer(rows, K, N, dev)

        def run(pairs):
            for _ in range(3):
                for arg, st in pairs:
                    hip.check(L.gad_gemm_fwd(C.byref(arg), C.c_void_p(st.cuda_stream)), "fwd")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            for _, st in pairs:
                st.wait_event(e0)
            for _ in range(n):
                for arg, st in pairs:
                    hip.check(L.gad_gemm_fwd(C.byref(arg), C.c_void_p(st.cuda_stream)), "fwd")
            for _, st in pairs:
                torch.cuda.current_stream().wait_stream(st)
            e1.record(torch.cuda.current_stream())
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        one = run([(a, s1)])
        two = run([(a, s1), (b, s2)])
        print("%-42s one chain %6.1f us per launch | two chains %6.1f us per pair | ratio %.2f  (%s)" % (
            name, one, two, two / one, L.gad_last_kernel().decode()))


if __name__ == "__main__":
    main()
