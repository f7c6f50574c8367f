"""Derived per-kernel table from the two SQ counter summaries tools/collect_profiles.sh writes (prof_pmc.py output):
   python tools/prof_derived.py gpurun_out/r03_rocprofv3_pmc_SQ_mfma.txt gpurun_out/r03_rocprofv3_pmc_SQ_waits.txt
Counter rows are per dispatch and XCD (8 per dispatch; 32 CUs each).  Columns:
  mfma_busy   SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES): share of CU-cycles with the matrix pipe busy
  valu/mfma   SQ_INSTS_VALU / SQ_INSTS_MFMA (SQ_INSTS_VALU counts the MFMAs too)
  wait_any    SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of wavefront-cycles spent waiting on anything (s_waitcnt, barrier, ...)
  wait_inst   SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: ... waiting for an instruction issue slot / dependency
  wait_lds    SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
  active      SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES: share of wavefront-cycles issuing
  lds_confl   SQ_LDS_BANK_CONFLICT / SQ_BUSY_CYCLES / 32: LDS bank-conflict cycles per CU-cycle"""
import re
import sys


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+(SQ_\w+)\s+n=\s*(\d+) avg=\s*([\d.]+) total=\s*([\d.]+)\s+avg_dur_us=\s*([\d.]+)", line)
        if m:
            out.setdefault(m.group(1), {})[m.group(2)] = (float(m.group(4)), float(m.group(6)), int(m.group(3)))
    return out


def short(name):
    m = re.match(r"([a-z_]+?)(I[\w]*?E)?(v|E|\d|P).*", name)
    base = re.match(r"[a-z_0-9]+?_kernel", name)
    targs = re.findall(r"L[ib](\d+)E", name.split("Ev")[0]) if "kernelI" in name else []
    return (base.group(0) if base else name[:40]) + ("<" + ",".join(targs) + ">" if targs else "")


def main(f_mfma, f_wait):
    a, b = parse(f_mfma), parse(f_wait)
    rows = []
    for k in a:
        c = a[k]
        if "SQ_BUSY_CYCLES" not in c or "SQ_INSTS_MFMA" not in c or c["SQ_INSTS_MFMA"][0] <= 0:
            continue
        busy = c["SQ_BUSY_CYCLES"][0]
        w = b.get(k, {})
        wc = w.get("SQ_WAVE_CYCLES", (0, 0, 0))[0]
        wb = w.get("SQ_BUSY_CYCLES", (busy, 0, 0))[0]

        def share(name):
            return w[name][0] / wc if name in w and wc else float("nan")
        rows.append((c["SQ_BUSY_CYCLES"][1] * c["SQ_BUSY_CYCLES"][2] / 8, short(k), c["SQ_BUSY_CYCLES"][2] // 8, c["SQ_BUSY_CYCLES"][1],
                     c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (32 * busy), c["SQ_INSTS_VALU"][0] / c["SQ_INSTS_MFMA"][0],
                     share("SQ_WAIT_ANY"), share("SQ_WAIT_INST_ANY"), share("SQ_WAIT_INST_LDS"),
                     share("SQ_ACTIVE_INST_ANY"), (w["SQ_LDS_BANK_CONFLICT"][0] / (32 * wb)) if "SQ_LDS_BANK_CONFLICT" in w else float("nan")))
    print(__doc__.split("Counter rows")[0].strip().splitlines()[0])
    print("%-36s %7s %8s %9s %9s %8s %9s %8s %7s %9s" % ("kernel (PMC pass: kernels serialised)", "calls", "avg us", "mfma_busy", "valu/mfma",
                                                             "wait_any", "wait_inst", "wait_lds", "active", "lds_confl"))
    for r in sorted(rows, reverse=True):
        print("%-36s %7d %8.1f %9.3f %9.1f %8.3f %9.3f %8.3f %7.3f %9.4f" % r[1:])
    print(__doc__.split("Columns:")[1].rstrip())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
