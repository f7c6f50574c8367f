"""VGPR / AGPR / scratch / occupancy / LDS of the kernels in a hipcc -Rpass-analysis=kernel-resource-usage log.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c gemm.hip -o /tmp/x.o 2> /tmp/ru.txt; python tools/kernel_resources.py /tmp/ru.txt [substring]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)
for b in blocks[1:]:
    name = b.split("\n")[0].strip()
    try:
        name = subprocess.check_output(["c++filt", name]).decode().strip()
    except Exception:
        pass
    if pat not in name:
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    print("%-110s VGPR %s AGPR %s scratch %s occ %s LDS %s" % (name.split("(")[0][:110], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                                                                g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
