"""Per-wavefront start / end stamps of ONE launch (gad_timing_slot, 100 MHz wall clock): how long is the launch from its first
wavefront's start to its last one's end, how long does a wavefront live, and how far apart do the wavefronts start (dispatch
ramp)?  For the latency-bound M = 256 layers (skinny kernels) and two mid-size shapes.

    python tools/ubench_wave_life.py
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
    khz = L.gad_wall_clock_khz()
    for name, (rows, K, N) in (("FC 1 (256 x 512 -> 1024)", (256, 512, 1024)), ("FC 2 (256 x 1024 -> 512)", (256, 1024, 512)),
                               ("head layer (256 x 256 -> 256)", (256, 256, 256)), ("SA3 layer 2 (8192 x 256 -> 256, wide)", (8192, 256, 256)),
                               ("SA2 layer 2 (27240 x 128 -> 128, wide)", (27240, 128, 128))):
        a = layer(rows, K, N, dev)
        for _ in range(5):
            hip.check(L.gad_gemm_fwd(C.byref(a), C.c_void_p(0)), "fwd")
        torch.cuda.synchronize()
        res = []
        for _ in range(5):
            slots = torch.zeros(16384, 2, dtype=torch.int64, device=dev)
            slots[:, 0] = torch.iinfo(torch.int64).max
            L.gad_timing_slot(C.c_void_p(slots.data_ptr()))
            hip.check(L.gad_gemm_fwd(C.byref(a), C.c_void_p(0)), "fwd")
            torch.cuda.synchronize()
            s = slots.cpu()
            live = s[:, 1] > 0
            st, en = s[live, 0].double(), s[live, 1].double()
            us = 1e3 / khz
            res.append(((en.max() - st.min()) * us, (en - st).mean() * us, (st.max() - st.min()) * us, int(live.sum())))
        r = sorted(res)[len(res) // 2]
        print("%-42s launch %5.1f us first start -> last end | wavefront life %5.1f us (mean) | starts spread over %4.1f us | %d wavefronts (%s)" % (
            name, float(r[0]), float(r[1]), float(r[2]), r[3], L.gad_last_kernel().decode()))


if __name__ == "__main__":
    main()
