#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name (sums over dispatches)."""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1]))
agg = collections.OrderedDict()
for r in rows:
    k = (r['Dispatch_Id'], r['Kernel_Name'].split('(')[0][:48])
    agg.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for k, v in agg.items():
    for a, b in v.items():
        tot[k[1]][a] += b
    tot[k[1]]['n'] += 1
key = sys.argv[2] if len(sys.argv) > 2 else 'SQ_BUSY_CYCLES'
for name, v in sorted(tot.items(), key=lambda kv: -kv[1].get(key, 0))[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(f"{name:<50} n={int(v['n']):6d} " + " ".join(f"{a}={b:.4g}" for a, b in sorted(v.items()) if a != 'n'))
