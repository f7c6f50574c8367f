"""Per-family A/B of the split-bf16 GEMM forms (library option "mfma_split") at the bench shapes: launch time alone on the
GPU and error against float64, FP32-MFMA path beside it (the tables of profiles/r05_split_families.txt).
    python tools/ubench_split.py [substring of the case names]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_split_families import _cases, check_case


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, make in _cases().items():
        if pat not in name:
            continue
        case = make()
        rep = []
        bad = check_case(case, name, rep)
        t32, tsp = case.time_mode(False), case.time_mode(True)
        fl = case.flops()
        print("%-18s f32-MFMA %6.1f us (%5.1f TFLOP/s) | split %6.1f us (%5.1f TFLOP/s f32-equivalent) | x%.2f  %s"
              % (name, t32, fl / t32 * 1e-6, tsp, fl / tsp * 1e-6, t32 / tsp, "OK" if not bad else "GATE FAILS"))
        for line in rep + bad:
            print("    " + line)


if __name__ == "__main__":
    main()
