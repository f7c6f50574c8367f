"""Instruction mix of a kernel's largest loop from hipcc's assembly (hipcc ... -S --cuda-device-only x.hip -o x.s):
    python tools/isa_mix.py x.s <mangled-name prefix> [...]"""
import collections
import re
import sys


def body(txt, name):
    i = txt.find("\n" + name)
    if i < 0:
        return None
    j = txt.find("s_endpgm", i)
    return txt[i:j].split("\n")


def biggest_loop(lines):
    best = None
    for h, l in enumerate(lines):
        if "Loop Header" in l and l.startswith(".LBB"):
            lab = l.split(":")[0].strip()
            ends = [i for i in range(h + 1, len(lines)) if re.search(r"s_cbranch\w+\s+%s\b" % re.escape(lab), lines[i])]
            if ends and (best is None or ends[-1] - h > len(best)):
                best = lines[h:ends[-1] + 1]
    return best or lines


def main():
    txt = open(sys.argv[1]).read()
    for nm in sys.argv[2:]:
        ls = body(txt, nm)
        if ls is None:
            print(nm, "not found")
            continue
        c = collections.Counter()
        for l in biggest_loop(ls):
            l = l.strip()
            if not l or l[0] in ";.":
                continue
            op = l.split()[0]
            kind = "mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else \
                "vmem" if op.startswith(("buffer_", "global_", "scratch_")) else "salu" if op.startswith("s_") else "other"
            c[kind] += 1
            c[kind + ":" + op] += 1
        print(nm[:60], {k: v for k, v in c.items() if ":" not in k})
        for kind in ("valu", "lds", "vmem"):
            print("   %s:" % kind, sorted([(v, k.split(":")[1]) for k, v in c.items() if k.startswith(kind + ":")], reverse=True)[:12])


if __name__ == "__main__":
    main()
