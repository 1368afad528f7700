#!/usr/bin/env python
"""Summarise a rocprofv3 ``--kernel-trace`` result database (rocpd SQLite) into a per-kernel
stats table (the text committed under profiles/).  Usage: rocpd_stats.py results.db [top_n]"""
import sqlite3
import sys


def main(path, top=40):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in tabs if "kernel_dispatch" in x][0]
    ks = [x for x in tabs if "kernel_symbol" in x][0]
    q = (f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
         f"max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    print(f"# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':<72} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>10} {'pct':>6}")
    for r in rows[:top]:
        name = r[0].split("(")[0][:72]
        print(f"{name:<72} {r[1]:>7d} {r[2] / 1e6:>10.3f} {r[3] / 1e3:>10.2f} {r[4] / 1e3:>9.2f} {r[5] / 1e3:>10.2f} "
              f"{100 * r[2] / tot:>5.1f}%")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
