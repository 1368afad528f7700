#!/usr/bin/env python
"""GPU idle-gap analysis of a rocprofv3 ``--kernel-trace --output-format csv`` run (<prefix>_kernel_trace.csv).

Kernels are sorted by start time; the device is "idle" wherever no kernel of ANY stream is running.  Every gap is
attributed to the kernel that ended last before it and the kernel that starts after it: the table says which host-side
decision points (device->host fetch + control flow + next launch) cost the most wall time -- the counter that matters
for a host-driven driver, where no roofline applies.
usage: trace_gaps.py p_kernel_trace.csv [top] [t_skip_frac]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as fh:
    rd = csv.DictReader(fh)
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + skip * (t1 - t0)
rows = [r for r in rows if r[0] >= lo]
busy = 0
gaps = defaultdict(lambda: [0, 0])
after = defaultdict(lambda: [0, 0])
cur_end, cur_name = rows[0][1], rows[0][2]
busy_start = rows[0][0]
hist = [0] * 8
for s, e, name in rows[1:]:
    if s > cur_end:
        g = s - cur_end
        busy += cur_end - busy_start
        busy_start = s
        k = gaps[(cur_name, name)]
        k[0] += g
        k[1] += 1
        a = after[cur_name]
        a[0] += g
        a[1] += 1
        b = 0
        while b < 7 and g >= 2000 * 4 ** b:
            b += 1
        hist[b] += g
    if e > cur_end:
        cur_end, cur_name = e, name
busy += cur_end - busy_start
span = cur_end - rows[0][0]
idle = span - busy
print(f"# span {span / 1e6:.1f} ms, device busy {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %), idle {idle / 1e6:.1f} ms "
      f"over {sum(v[1] for v in gaps.values())} gaps; {len(rows)} dispatches")
print("# idle time by gap length: " + ", ".join(
    f"{'<' if i == 0 else ''}{2 * 4 ** i if i < 7 else 2 * 4 ** 6}us{'+' if i == 7 else ''}: {h / 1e6:.1f} ms"
    for i, h in enumerate(hist)))
print(f"{'idle after kernel':<50} {'gaps':>7} {'idle_ms':>9} {'avg_us':>8}")
for name, (g, n) in sorted(after.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{name:<50} {n:>7d} {g / 1e6:>9.2f} {g / n / 1e3:>8.1f}")
print(f"\n{'gap between (previous -> next)':<100} {'gaps':>7} {'idle_ms':>9} {'avg_us':>8}")
for (a, b), (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{(a + ' -> ' + b):<100} {n:>7d} {g / 1e6:>9.2f} {g / n / 1e3:>8.1f}")
