import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_kernel_families import _run
for B in (32, 96):
    a = _run(B, True, {"fwd_stream": 1})
    b = _run(B, True, {"fwd_stream": 0})
    print("B", B, "rows", a["rows"])
    for k in a:
        if k == "rows": continue
        x, y = a[k].double(), b[k].double()
        sc = float(y.abs().max()) + 1e-30
        d = (x - y).abs()
        print("%-14s max %.3e median %.3e (scale %.3e) nonzero-frac %.3f" % (k, float(d.max()) / sc, float(d.median()) / sc, sc, float((d > 0).double().mean())))
