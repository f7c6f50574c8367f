"""Critical-path view of a rocprofv3 --kernel-trace CSV of bench.py: per hardware queue timelines of the timed steps,
how much of the wall time has 0 / 1 / 2 / 3+ kernels in flight, time per kernel symbol, and the gaps between consecutive
dispatches of one queue.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -o r -- python bench.py --steps 24 --warmup 10 ...
    python tools/trace_timeline.py /tmp/ktrace/.../r_kernel_trace.csv [first_frac last_frac]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?", name)
    base = m.group(1) if m else name
    targs = (m.group(2) or "") if m else ""
    targs = re.sub(r"\s+", "", targs)
    return (base + targs)[:60]


def main():
    path = sys.argv[1]
    lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.55
    hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]),
                         int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
    rows.sort()
    t0, t1 = rows[0][0], rows[-1][1]
    a, b = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
    win = [r for r in rows if r[0] >= a and r[1] <= b]
    span = (win[-1][1] - win[0][0]) * 1e-3
    print("window: %d dispatches over %.1f us" % (len(win), span))
    # concurrency histogram
    ev = []
    for s, e, q, n, g in win:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    cur, last, hist = 0, ev[0][0], defaultdict(float)
    for t, d in ev:
        hist[min(cur, 4)] += (t - last) * 1e-3
        cur += d; last = t
    tot = sum(hist.values())
    print("kernels in flight:  " + "  ".join("%d%s: %.1f%%" % (k, "+" if k == 4 else "", 100 * v / tot) for k, v in sorted(hist.items())))
    # per symbol
    by = defaultdict(lambda: [0, 0.0])
    for s, e, q, n, g in win:
        by[n][0] += 1; by[n][1] += (e - s) * 1e-3
    busy = sum(v[1] for v in by.values())
    print("sum of kernel durations %.1f us = %.2f x the window" % (busy, busy / span))
    for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:45]:
        print("  %-62s x%5d  %9.1f us  avg %6.1f  %5.1f%% of window" % (n, c, t, t / c, 100 * t / span))
    # per queue: busy time and gap statistics
    perq = defaultdict(list)
    for r in win:
        perq[r[2]].append(r)
    for q, rs in sorted(perq.items()):
        rs.sort()
        qbusy = sum(e - s for s, e, _, _, _ in rs) * 1e-3
        gaps = [(rs[i + 1][0] - rs[i][1]) * 1e-3 for i in range(len(rs) - 1)]
        small = [g for g in gaps if 0 <= g < 30]
        print("queue %d: %5d dispatches, busy %.1f us (%.0f%%), back-to-back gaps (<30us): n=%d median %.1f us mean %.1f us" % (
            q, len(rs), qbusy, 100 * qbusy / span, len(small), sorted(small)[len(small) // 2] if small else 0, sum(small) / max(1, len(small))))
    if "--dump" in sys.argv:
        base = win[0][0]
        for s, e, q, n, g in win[:700]:
            print("%9.1f %9.1f %7.1f q%d %5d %s" % ((s - base) * 1e-3, (e - base) * 1e-3, (e - s) * 1e-3, q, g, n))


if __name__ == "__main__":
    main()
