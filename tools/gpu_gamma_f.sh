#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/kt -o kt --output-format csv -- python $R/tools/gemm_real_bench.py 264859 503 gramscan > /tmp/kt.log 2>&1
python - <<'PY'
import csv
rows = list(csv.DictReader(open("/tmp/kt/kt_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:40]
    if "zgemm" not in name:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{name:42s} grid={r.get('Grid_Size_X', r.get('Grid_Size','?')):>8s} wg={r.get('Workgroup_Size_X','?'):>4s} dur={(e - s) / 1e3:10.1f} us  gap={(s - t0) / 1e3 if t0 else 0:8.1f}")
    t0 = e
PY
