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
from tools.ubench_overlap import layer


def main():
    dev = torch.device("cuda:0")
    L = hip.lib()
    for name, (rows, K, N) in (("SA1 layer 2 (64 -> 64)", (213034, 64, 64)), ("SA1 layer 3 shape, no pool (64 -> 128)", (213034, 64, 128))):
        a = layer(rows, K, N, dev)
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
