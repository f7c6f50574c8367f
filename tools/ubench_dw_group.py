"""The weight gradients of a stage as one grouped launch (gad_gemm_dw_group) against its three single launches: time alone.
    python tools/ubench_dw_group.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ga_ddpg_amd import hip
from ga_ddpg_amd.engine import _ptr
from tests import split_cases as sc


def main():
    for stage, (rows, n2, n3) in (("sa2", (27240, 128, 256)), ("sa3", (8192, 256, 512))):
        cases = [sc.DwWide(rows, n3, n2, "pool", seed=5), sc.DwWide(rows, n2, n2, "act", seed=6), sc.DwWide(rows, n2, n2, "gather", seed=7)]
        jobs = []
        for c in cases:
            a = c.args()
            a.inp.n_rows_dev, a.inp.n_rows = _ptr(cases[0].dx.nrows), cases[0].dx.cap
            a.partial, a.partial_elems = _ptr(cases[0].ws), cases[0].ws.numel()
            jobs.append(a)
        arr = (C.c_void_p * 3)(*[C.addressof(a) for a in jobs])
        L, st = hip.lib(), hip.stream()
        hip.set_option("mfma_split", 1)
        for grp in (1, 0, 1, 0):
            hip.set_option("dw_group", grp)
            t = sc.time_call(lambda: hip.check(L.gad_gemm_dw_group(arr, 3, st), "grp"), 40)
            print("%s dw_group=%d: %.1f us for the stage's three weight gradients (%s)" % (stage, grp, t, L.gad_last_kernel().decode()))
        hip.set_option("dw_group", 1)


if __name__ == "__main__":
    main()
