"""Digest of a tools/prof_pmc.py summary (SQ MFMA / VALU counters): per kernel the MFMA-busy fraction, executed FP32-MFMA
TFLOP/s and VALU instructions per MFMA.   python tools/prof_pmc_digest.py <summary.txt>"""
import re
import sys

rows = {}
for l in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+(SQ_\w+)\s+n=\s*(\d+) avg=\s*([\d.]+) total=\s*([\d.]+)\s+avg_dur_us=\s*([\d.]+)", l)
    if m:
        rows.setdefault(m.group(1), {})[m.group(2)] = (int(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6)))
print("# counters are reported per shader-engine instance (32 samples per dispatch); kernels are serialised under --pmc")
print("# mfma_busy = MFMA-busy SIMD cycles / (SQ_BUSY_CYCLES x 1024 SIMDs); TFLOP/s = MFMA_MOPS_F32 x 512 / duration (executed FP32 MFMA flops)")
print("%-62s %8s %8s %10s %9s %10s %8s" % ("kernel", "launches", "dur_us", "mfma_busy", "TFLOP/s", "VALU/MFMA", "GFLOP"))
out = []
for k, c in rows.items():
    if "SQ_INSTS_MFMA" not in c or c["SQ_INSTS_MFMA"][2] == 0 or "SQ_BUSY_CYCLES" not in c:
        continue
    n, dur = c["SQ_BUSY_CYCLES"][0] // 32, c["SQ_BUSY_CYCLES"][3]
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (c["SQ_BUSY_CYCLES"][1] * 32.0)
    flop = c["SQ_INSTS_VALU_MFMA_MOPS_F32"][1] * 32 * 512
    valu = (c["SQ_INSTS_VALU"][2] - c["SQ_INSTS_MFMA"][2]) / c["SQ_INSTS_MFMA"][2]
    out.append((n * dur, "%-62s %8d %8.2f %9.1f%% %9.1f %10.2f %8.2f" % (k[:62], n, dur, 100 * busy, flop / dur / 1e6, valu, flop / 1e9)))
for _, l in sorted(out, reverse=True):
    print(l)
