"""Summarise a rocprofv3 --kernel-trace rocpd database: per-kernel count / total / avg (profiles/*.txt)."""
import glob
import re
import sqlite3
import sys


def main(path, steps):
    db = glob.glob(path + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
                     "max(d.end-d.start) from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
                     % (kd, ks)).fetchall()
    tot = sum(r[2] for r in rows)
    t0, t1 = c.execute("select min(start), max(end) from %s" % kd).fetchone()
    print("kernels: %d dispatches, busy %.3f ms over a %.3f ms span; %d steps -> %.3f ms kernel time / step"
          % (sum(r[1] for r in rows), tot / 1e6, (t1 - t0) / 1e6, steps, tot / 1e6 / steps))
    print("%-86s %8s %11s %10s %9s %9s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%"))
    for r in rows[:45]:
        name = re.sub(r"\(.*", "", r[0])
        name = re.sub(r"^_Z\d+", "", name)[:86]
        print("%-86s %8d %11.3f %10.2f %9.2f %9.2f %6.1f" % (name, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3,
                                                            100.0 * r[2] / tot))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))


def streams(path, steps):
    """per-queue busy time (which HIP stream is the long pole of a step)"""
    db = glob.glob(path + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    key = "stream_id" if "stream_id" in cols else "queue_id"
    t0, t1 = c.execute("select min(start), max(end) from %s" % kd).fetchone()
    print("span %.3f ms, %d steps -> %.3f ms / step; busy per %s:" % ((t1 - t0) / 1e6, steps, (t1 - t0) / 1e6 / steps, key))
    for q, n, busy in c.execute("select %s, count(*), sum(end-start) from %s group by %s order by 3 desc" % (key, kd, key)):
        print("   %s %-6s launches %6d  busy %.3f ms / step" % (key, q, n, busy / 1e6 / steps))


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "streams":
    streams(sys.argv[1], int(sys.argv[2]))


def gaps(path, steps):
    """idle time between consecutive kernels of the busiest stream, split into launch gaps (< 10 us) and waits"""
    db = glob.glob(path + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    key = "stream_id" if "stream_id" in cols else "queue_id"
    q = c.execute("select %s from %s group by %s order by sum(end-start) desc limit 1" % (key, kd, key)).fetchone()[0]
    rows = c.execute("select start, end from %s where %s=? order by start" % (kd, key), (q,)).fetchall()
    small = big = 0.0
    nsmall = nbig = 0
    hist = {}
    for (s0, e0), (s1, e1) in zip(rows[:-1], rows[1:]):
        g = (s1 - e0) / 1e3
        if g < 0:
            continue
        b = 1 if g < 1 else 2 if g < 2 else 4 if g < 4 else 10 if g < 10 else 50 if g < 50 else 1000
        hist[b] = hist.get(b, 0) + 1
        if g < 10:
            small += g; nsmall += 1
        else:
            big += g; nbig += 1
    print("stream %s: %d kernels; launch gaps (<10 us): %d, %.3f ms / step; waits (>=10 us): %d, %.3f ms / step" % (
        q, len(rows), nsmall, small / 1e3 / steps, nbig, big / 1e3 / steps))
    print("gap histogram (upper bound us -> count per step):", {k: round(v / steps, 1) for k, v in sorted(hist.items())})


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "gaps":
    gaps(sys.argv[1], int(sys.argv[2]))


def by_stream(path, steps, which):
    """per-kernel totals of ONE stream (default: the busiest = the main stream)"""
    db = glob.glob(path + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    key = "stream_id" if "stream_id" in cols else "queue_id"
    order = [r[0] for r in c.execute("select %s from %s group by %s order by sum(end-start) desc" % (key, kd, key))]
    q = order[which]
    rows = c.execute("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from %s d join %s s on d.kernel_id=s.id "
                     "where d.%s=? group by s.kernel_name order by 3 desc" % (kd, ks, key), (q,)).fetchall()
    print("stream %s: %.3f ms / step" % (q, sum(r[2] for r in rows) / 1e6 / steps))
    for r in rows[:30]:
        name = re.sub(r"\(.*", "", r[0])
        name = re.sub(r"^_Z\d+", "", name)[:70]
        print("%-72s %6.1f calls/step %8.1f us avg %8.3f ms/step" % (name, r[1] / steps, r[3] / 1e3, r[2] / 1e6 / steps))


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "stream":
    by_stream(sys.argv[1], int(sys.argv[2]), int(sys.argv[4]) if len(sys.argv) > 4 else 0)


def timeline(path, step_no, min_us=6.0):
    """kernels of ONE steady-state step in start order: offset, duration, stream, name (kernels shorter than min_us are
    folded into a count) -- for reading the critical path across streams.  Steps are delimited by actor_loss/adam pairs:
    the step starts at the first prep_points_kernel after the previous step's last adam_kernel."""
    db = glob.glob(path + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    key = "stream_id" if "stream_id" in cols else "queue_id"
    rows = c.execute("select d.start, d.end, d.%s, s.kernel_name from %s d join %s s on d.kernel_id=s.id order by d.start"
                     % (key, kd, ks)).fetchall()
    starts = [i for i, r in enumerate(rows) if "prep_points" in r[3]]
    starts = starts[::2]                                           # two per step (current / next clouds)
    a, b = starts[step_no], starts[step_no + 1]
    t0 = rows[a][0]
    sid = {}
    small = 0
    print("step %d: %.3f ms, %d kernels" % (step_no, (rows[b][0] - t0) / 1e6, b - a))
    last_end = {}
    for s, e, q, name in rows[a:b]:
        q = sid.setdefault(q, len(sid))
        name = re.sub(r"\(.*", "", name)
        name = re.sub(r"^_Z\d+", "", name)[:46]
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        if (e - s) / 1e3 < min_us and gap < 10:
            small += 1
            continue
        print("%8.1f us  +%6.1f  s%d %s%-46s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, "   " * q, name,
                                                   ("(idle %.0f us before)" % gap) if gap >= 10 else ""))
    print("(%d kernels under %.0f us not shown)" % (small, min_us))


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "timeline":
    timeline(sys.argv[1], int(sys.argv[2]), float(sys.argv[4]) if len(sys.argv) > 4 else 6.0)
