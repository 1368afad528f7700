#!/usr/bin/env python
"""Format rocprofv3 `--kernel-trace --stats --output-format csv` (<prefix>_kernel_stats.csv) as the text
table committed under profiles/.  usage: kernel_stats_txt.py p_kernel_stats.csv [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# rocprofv3 --kernel-trace --stats summary; total kernel time {tot / 1e6:.3f} ms over "
      f"{sum(int(r['Calls']) for r in rows)} dispatches")
print(f"{'kernel':<64} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>10} {'pct':>6}")
for r in rows[:top]:
    print(f"{r['Name'].split('(')[0][:64]:<64} {int(r['Calls']):>7d} {float(r['TotalDurationNs']) / 1e6:>10.3f} "
          f"{float(r['AverageNs']) / 1e3:>10.2f} {float(r['MinNs']) / 1e3:>9.2f} {float(r['MaxNs']) / 1e3:>10.2f} "
          f"{float(r['Percentage']):>5.1f}%")
