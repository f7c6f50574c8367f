"""Where a workgroup of the wide-tile forward kernel spends its life (diagnostic build only: the library compiled with
-DGAD_W_PHASES writes, per workgroup, the 100 MHz wall clock at the end of each part of its first row tile into the upper
half of the launch's timing slot): prologue (BatchNorm vectors + first loads issued + first barrier) | first K-tile staged in LDS
(= the global-load latency) | K loop | stores + statistics issued | column atomics + stores complete.

    make -C ga-ddpg_amd/csrc BUILD=build_wph OUT=../../tools/_ab/lib_wph.so EXTRA=-DGAD_W_PHASES
    GAD_LIB_PATH=tools/_ab/lib_wph.so python tools/ubench_wphases.py [substring of the case names]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ga_ddpg_amd import hip
from tests.test_gpu_split_families import _cases


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else "fwd_wide"
    L = hip.lib()
    for name, make in _cases().items():
        if pat not in name:
            continue
        case = make()
        for split in (True, False):
            hip.set_option("mfma_split", case.family if split else 0)
            a = case.args()
            f = getattr(L, case.entry)
            for _ in range(5):
                hip.check(f(C.byref(a), C.c_void_p(0)), case.entry)
            slots = torch.zeros(16384, 2, dtype=torch.int64, device="cuda")
            slots[:8192, 0] = torch.iinfo(torch.int64).max
            L.gad_timing_slot(C.c_void_p(slots.data_ptr()))
            hip.check(f(C.byref(a), C.c_void_p(0)), case.entry)
            torch.cuda.synchronize()
            s = slots.cpu().numpy()
            wav = s[:8192]
            livew = wav[:, 1] > 0
            t0, t1 = wav[livew, 0].min(), wav[livew, 1].max()
            ph = s[8192:12288]
            live = ph[:, 0] > 0
            start = (ph[live, 0] - t0) / 100.0                       # us after the first wavefront of the launch
            q = ph[live, 1].astype(np.uint64)
            c = [((q >> np.uint64(sh)) & np.uint64(0xffff)).astype(np.float64) / 100.0 for sh in (0, 16, 32, 48)]
            c5 = s[12288:16384][live, 0].astype(np.float64) / 100.0
            print("%-16s %-5s launch %.1f us, %d workgroups; workgroup start after launch begin: mean %.2f max %.2f us" % (
                name, "split" if split else "f32", (t1 - t0) / 100.0, int(live.sum()), start.mean(), start.max()))
            print("      mean us from workgroup start to: prologue done %.2f | first tile staged %.2f | K loop done %.2f | stores+stats issued %.2f | "
                  "atomics + stores complete %.2f   (max end %.2f)" % (c[0].mean(), c[1].mean(), c[2].mean(), c[3].mean(), c5.mean(),
                                                                       (start + c5).max()))
            kl = s[4096:8192]
            nw = 4
            nblk = int(live.sum())
            if split and nblk * nw <= 4096:
                e = kl[:8 * nblk].reshape(nblk, 8, 2)[:, :nw]
                mf = (e[:, :, 0].astype(np.uint64) >> np.uint64(32)).astype(np.float64)
                stg = (e[:, :, 0].astype(np.uint64) & np.uint64(0xffffffff)).astype(np.float64)
                bar = e[:, :, 1].astype(np.float64)
                nk = case.K // 32
                print("      K loop, shader cycles per K-tile and wavefront (%d wavefronts per workgroup, %d K-tiles): fragments + MFMAs %.0f | stage next tile + "
                      "issue loads %.0f | barrier wait %.0f" % (nw, nk, mf.mean() / nk, stg.mean() / nk, bar.mean() / nk))
        hip.set_option("mfma_split", hip.get_option_default("mfma_split"))


if __name__ == "__main__":
    main()
