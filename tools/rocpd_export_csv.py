#!/usr/bin/env python
"""Export a rocprofv3 result database (rocpd SQLite, what every run leaves under -d even when the profiler's own tear-down
hangs and has to be killed) to the two CSV layouts the tools of this directory read:

  rocpd_export_csv.py results.db kernel_trace out.csv   -> Kernel_Name, Start_Timestamp, End_Timestamp   (trace_gaps.py)
  rocpd_export_csv.py results.db counters out.csv       -> Dispatch_Id, Kernel_Name, Counter_Name, Counter_Value
                                                            (pmc_to_traffic.py, pmc_summary.py)
"""
import csv
import sqlite3
import sys


def main(db, what, out):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table', 'view')")]
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        if what == "kernel_trace":
            kd = [x for x in tabs if x.startswith("rocpd_kernel_dispatch")][0]
            ks = [x for x in tabs if x.startswith("rocpd_info_kernel_symbol")][0]
            w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
            for r in cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"):
                w.writerow(r)
        elif what == "counters":
            w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            for r in cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
                w.writerow(r)
        else:
            raise SystemExit(__doc__)


if __name__ == "__main__":
    main(*sys.argv[1:4])
