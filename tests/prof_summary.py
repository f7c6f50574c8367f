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
