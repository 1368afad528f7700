# MFMA utilisation of the zgemm kernels (rocprofv3 --pmc, counters only): matrix-pipe busy cycles per GPU-active
# cycle for every kernel of tools/gemm_bench.py, normalised by the same ratio of the pure-MFMA calibration
# kernel k_mfma_peak (4 waves/SIMD, no memory traffic) that the script runs first.  Run through gpurun.
# TAG=r05 BENCH="tools/gemm_real_bench.py 264859 503 struct" bash tools/pmc_mfma_util.sh  -> gpurun_out/${TAG}_pmc_mfma_util.txt
# (the calibration kernel comes from tools/gemm_bench.py, which always runs first; BENCH adds the kernels of another harness)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r01}
rm -rf /tmp/pm /tmp/pm2
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d /tmp/pm -o pm --output-format csv -- python $R/tools/gemm_bench.py > /tmp/gemm_bench_out.txt 2>&1
if [ -n "$BENCH" ]; then
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d /tmp/pm2 -o pm --output-format csv -- python $R/$BENCH >> /tmp/gemm_bench_out.txt 2>&1
  python - <<'PY2'
import csv
a = list(csv.DictReader(open('/tmp/pm/pm_counter_collection.csv')))
b = list(csv.DictReader(open('/tmp/pm2/pm_counter_collection.csv')))
off = max(int(r['Dispatch_Id']) for r in a) + 1
for r in b:
    r['Dispatch_Id'] = str(int(r['Dispatch_Id']) + off)
w = csv.DictWriter(open('/tmp/pm/pm_counter_collection.csv', 'w', newline=''), fieldnames=list(a[0].keys()))
w.writeheader(); w.writerows(a + b)
PY2
fi
python - <<'PY' > $R/gpurun_out/${TAG}_pmc_mfma_util.txt
import collections, csv
rows = csv.DictReader(open('/tmp/pm/pm_counter_collection.csv'))
disp = collections.OrderedDict()
for r in rows:
    k = (int(r['Dispatch_Id']), r['Kernel_Name'].split('(')[0][:44])
    disp.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])
agg = collections.OrderedDict()
for (d, name), v in disp.items():
    a = agg.setdefault(name, collections.defaultdict(float))
    for c, x in v.items():
        a[c] += x
    a['n'] += 1
def ratio(v):
    return v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(v['GRBM_GUI_ACTIVE'], 1.0)
cal = max((ratio(v) for n, v in agg.items() if 'k_mfma_peak' in n), default=None)
print('# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA -- python tools/gemm_bench.py')
print('# util = (MFMA_BUSY / GUI_ACTIVE) / same ratio of k_mfma_peak (pure MFMA stream, calibration = 1.00)')
print(f"{'kernel':<42}{'launches':>9}{'MFMA_BUSY':>14}{'GUI_ACTIVE':>14}{'INSTS_MFMA':>14}{'util':>7}")
for n, v in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_VALU_MFMA_BUSY_CYCLES']):
    if v['SQ_VALU_MFMA_BUSY_CYCLES'] <= 0:
        continue
    u = ratio(v) / cal if cal else float('nan')
    print(f"{n:<42}{int(v['n']):>9}{v['SQ_VALU_MFMA_BUSY_CYCLES']:>14.4g}{v['GRBM_GUI_ACTIVE']:>14.4g}{v['SQ_INSTS_MFMA']:>14.4g}{u:>7.2f}")
PY
grep -E "TFLOP|TF/s" /tmp/gemm_bench_out.txt | grep -v "^ *plan" >> $R/gpurun_out/${TAG}_pmc_mfma_util.txt
cat $R/gpurun_out/${TAG}_pmc_mfma_util.txt
