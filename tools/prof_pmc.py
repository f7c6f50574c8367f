"""Summarise a rocprofv3 --pmc run (rocpd database): average counter value per dispatch of each kernel."""
import glob
import re
import sqlite3
import sys


def main(path):
    db = glob.glob(path + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
    print("tables:", [t.split("_0000")[0] for t in tabs])
    pmc = [t for t in tabs if "pmc_event" in t]
    if not pmc:
        print("no pmc table")
        return
    pmc = pmc[0]
    print(pmc, [r[1] for r in c.execute("pragma table_info(%s)" % pmc)])
    info = [t for t in tabs if "info_pmc" in t][0]
    print(info, [r[1] for r in c.execute("pragma table_info(%s)" % info)])
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = ("select s.kernel_name, i.name, count(*), avg(p.value), sum(p.value), avg(d.end-d.start) from %s p join %s i on p.pmc_id=i.id "
         "join %s d on p.event_id=d.event_id join %s s on d.kernel_id=s.id group by s.kernel_name, i.name order by 5 desc"
         % (pmc, info, kd, ks))
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    for r in c.execute(q).fetchall()[:limit]:
        name = re.sub(r"\(.*", "", r[0])
        name = re.sub(r"^_Z\d+", "", name)[:70]
        print("%-72s %-28s n=%6d avg=%14.1f total=%16.1f  avg_dur_us=%8.2f" % (name, r[1], r[2], r[3], r[4], r[5] / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
