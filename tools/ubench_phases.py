"""Where a wavefront of the streaming forward kernel spends its cycles (diagnostic build only: the library compiled with
-DGAD_X_PHASES=1 writes, per wavefront, the shader-clock cycles it spent (a) from the end of a slab's epilogue to the first
MFMA of the next -- prefetch issue + the wait for the slab's own loads, (b) in the MFMA loop, (c) in the epilogue -- stores,
statistics, pool -- into the launch's timing slot).

    GAD_LIB_PATH=tools/_ab/lib_phases.so python tools/ubench_phases.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ga_ddpg_amd import hip
from ga_ddpg_amd.engine import _ptr
from tools.ubench_overlap import layer


def main():
    dev = torch.device("cuda:0")
    L = hip.lib()
    for name, (rows, K, N) in (("SA1 layer 2 (64 -> 64)", (213034, 64, 64)), ("SA1 layer 3 shape, no pool (64 -> 128)", (213034, 64, 128))):
        a = layer(rows, K, N, dev)
        if os.environ.get("PHASES_IN_BN") == "1":            # the consumer finalises its input layer's BatchNorm in its prologue
            R = hip.STAT_REPLICAS
            st = torch.zeros(R * 2 * K, dtype=torch.float64, device=dev)
            st.view(R, 2, K)[:, 0] = 0.1 * rows / R
            st.view(R, 2, K)[:, 1] = 1.0 * rows / R
            g, b = torch.ones(K, device=dev), torch.zeros(K, device=dev)
            mean, istd = torch.empty(K, device=dev), torch.empty(K, device=dev)
            a.in_stat_sum, a.in_stat_sq, a.in_stat_stride = _ptr(st, 0, 8), _ptr(st, K, 8), 2 * K
            a.in_count, a.in_gamma, a.in_beta, a.in_eps, a.in_momentum = float(rows), _ptr(g), _ptr(b), 1e-5, 0.1
            a.in_mean, a.in_istd = _ptr(mean), _ptr(istd)
            a._keep2 = (st, g, b, mean, istd)
        for _ in range(3):
            hip.check(L.gad_gemm_fwd(C.byref(a), C.c_void_p(0)), "fwd")
        slots = torch.zeros(16384, 2, dtype=torch.int64, device=dev)
        L.gad_timing_slot(C.c_void_p(slots.data_ptr()))
        hip.check(L.gad_gemm_fwd(C.byref(a), C.c_void_p(0)), "fwd")
        torch.cuda.synchronize()
        s = slots[:2048].cpu()
        mfma, epi = (s[:, 0] >> 32) & 0xffffffff, s[:, 0] & 0xffffffff
        top, life, n = (s[:, 1] >> 32) & 0xffffffff, s[:, 1] & 0xffffff00, s[:, 1] & 0xff
        live = n > 0
        if os.environ.get("PHASES_STARTUP") == "1":          # -DGAD_X_PHASES=2 build: the prologue's stamps (units of 4 cycles)
            q = s[:, 0]
            parts = [((q >> sh) & 0xffff)[live].float().mean().item() * 4 for sh in (0, 16, 32, 48)]
            print("%s: cycles from wavefront start to: live row count arrived %.0f | first slab + W loads issued %.0f | arrived %.0f | "
                  "barrier passed %.0f" % ((name,) + tuple(parts)))
            continue
        f = lambda t: "%7.0f" % float(t[live].float().mean())
        per = lambda t: "%6.0f" % float((t[live].float() / n[live].float()).mean())
        print("%s: %d wavefronts, slabs per wavefront %.2f (max %d)" % (name, int(live.sum()), float(n[live].float().mean()), int(n.max())))
        print("   cycles per wavefront: launch -> end of last slab %s | wait+prefetch %s | MFMA loop %s | epilogue %s" % (
            f(life), f(top), f(mfma), f(epi)))
        print("   cycles per slab:      wait+prefetch %s | MFMA loop %s (MFMA pipe alone: %d) | epilogue %s" % (
            per(top), per(mfma), 64 * 4 * 8 * (N // 32), per(epi)))
        startup = (life - top - mfma - epi)[live].float().mean()
        print("   launch -> first slab's prefetch issued: %.0f cycles" % float(startup))


if __name__ == "__main__":
    main()
