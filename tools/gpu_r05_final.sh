# Round-5 measurement pass on the committed tree: the driver's three steps (pytest -m gpu, smoke, bench), the C consumer,
# PMC traffic, kernel trace + idle gaps, secondary configs, late-step table, kernel micro-benchmarks.
# Everything lands in gpurun_out/r05_final/; what should be judged is copied into profiles/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_final
mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 2400 python -m pytest tests -m gpu -q --tb=short --durations=8 -p no:cacheprovider 2>&1 \
  | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path' > $O/r05_pytest_gpu.log
tail -14 $O/r05_pytest_gpu.log
fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/r05_pytest_gpu.log
./tools/bin/abi_c_check 1 2>&1 | tail -3 | tee -a $O/r05_pytest_gpu.log
# PMC traffic first: the bench lines below then quote it (same build, same workload)
TAG=r05 timeout 1500 bash $R/tools/pmc_traffic_bench.sh > $O/pmc.log 2>&1
cp $R/gpurun_out/r05_pmc_traffic.json $R/gpurun_out/r05_pmc_fetch_size.txt $R/gpurun_out/r05_pmc_write_size.txt $O/ 2>/dev/null
mkdir -p $R/profiles && cp $R/gpurun_out/r05_pmc_traffic.json $R/profiles/ 2>/dev/null
tail -3 $O/pmc.log | cut -c1-300
unset DFTK_MI_HEEV_PARTIAL PMC_NOTE
cd $R
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $O/r05_bench_cfg5_driver_args.json 2> $O/bench_driver.err ) 2> $O/bench_driver.time
( time timeout 1500 python bench.py --no-cpu-baseline > $O/r05_bench_cfg5.json 2> $O/bench_cfg5.err ) 2> $O/bench_cfg5.time
python - <<'PY'
import json
for f in ("r05_bench_cfg5_driver_args", "r05_bench_cfg5"):
    d = json.loads(open(f"gpurun_out/r05_final/{f}.json").read().strip().splitlines()[-1])
    r, c = d["roofline"], d["config"]
    print(f, round(d["value"], 3), d["steps"], c["scf_wall_s"], c["converged"], round(r["frac"], 3), r["traffic"], c.get("complex_iteration_value"), (d.get("cpu_baseline") or {}).get("value"), c.get("parity_pass"), c.get("late_step_ms"), c.get("step_roofline_frac"), (d.get("amdahl") or {}).get("predicted_speedup"), (d.get("amdahl") or {}).get("measured_speedup"))
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o p -- python $R/bench.py --no-cpu-baseline --no-complex-leg --no-parity --no-amdahl-probe > /tmp/bench_kt.json 2>/tmp/bench_kt.err
python $R/tools/kernel_stats_txt.py /tmp/kt/p_kernel_stats.csv 44 > $O/r05_kernel_trace_cfg5.txt
tail -1 /tmp/bench_kt.json >> $O/r05_kernel_trace_cfg5.txt
python $R/tools/trace_gaps.py /tmp/kt/p_kernel_trace.csv 24 0.5 > $O/r05_trace_gaps_cfg5.txt
head -14 $O/r05_kernel_trace_cfg5.txt; head -6 $O/r05_trace_gaps_cfg5.txt
rm -rf /tmp/kt
cd $R
timeout 600 python bench.py --supercell 4 --no-cpu-baseline --no-complex-leg > $O/r05_bench_cfg2.json 2> $O/bench_cfg2.err
timeout 900 python bench.py --mode kpoints --system al --no-cpu-baseline > $O/r05_bench_kpoints_al.json 2> $O/bench_kpoints_al.err
for S in si graphene; do
  timeout 600 python bench.py --mode kpoints --system $S --no-cpu-baseline --no-amdahl-probe > $O/r05_bench_kpoints_$S.json 2> $O/bench_kpoints_$S.err
done
python - <<'PY'
import json
for f in ("r05_bench_cfg2", "r05_bench_kpoints_al", "r05_bench_kpoints_si", "r05_bench_kpoints_graphene"):
    try:
        d = json.loads(open(f"gpurun_out/r05_final/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 2), d["steps"], d["config"]["scf_wall_s"], d["config"]["E_total"], (d.get("amdahl") or {}).get("predicted_speedup"), (d.get("amdahl") or {}).get("measured_speedup"), (d["config"].get("parity") or {}).get("pass"))
    except Exception as e:
        print(f, "failed", e)
PY
DFTK_MI_GEMM_SHAPES=1 timeout 300 python tools/late_step_profile.py 5 8 6 > $O/r05_late_step_cfg5.txt 2> $O/late.err
cat $O/r05_late_step_cfg5.txt | head -24
timeout 300 python tools/heev_bench.py lowest 1006 1509 777 > $O/r05_heev_lowest_bench.txt 2>&1
timeout 300 python tools/heev_bench.py lowest complex 1006 1509 >> $O/r05_heev_lowest_bench.txt 2>&1
cat $O/r05_heev_lowest_bench.txt
timeout 300 python tools/gemm_real_bench.py 264859 503 struct 2>&1 | grep -v "^ *plan" > $O/r05_gemm_struct_bench.txt
timeout 300 python tools/fft_bench.py > $O/r05_fft_bench_192.txt 2>&1
timeout 300 python tools/ew_bench.py > $O/r05_ew_bench.txt 2>&1
tail -8 $O/r05_fft_bench_192.txt
