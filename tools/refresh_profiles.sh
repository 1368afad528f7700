# Regenerates every artefact committed under profiles/ for the current build (run through gpurun).
R=$GRAFT_REPO_ROOT
TAG=${1:-v3}
mkdir -p $R/gpurun_out/profiles
cd $R && python -m pytest tests -m gpu -q --durations=10 2>&1 | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path' | tail -30 > $R/gpurun_out/profiles/r01_pytest_gpu_$TAG.log
cd $R && python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -1 >> $R/gpurun_out/profiles/r01_pytest_gpu_$TAG.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o p -- python $R/bench.py --no-cpu-baseline > /tmp/bench_kt.json 2>/dev/null
python $R/tools/kernel_stats_txt.py /tmp/kt/p_kernel_stats.csv 40 > $R/gpurun_out/profiles/r01_kernel_trace_cfg2_$TAG.txt
tail -1 /tmp/bench_kt.json >> $R/gpurun_out/profiles/r01_kernel_trace_cfg2_$TAG.txt
bash $R/tools/pmc_traffic_bench.sh > /dev/null 2>&1
cp $R/gpurun_out/r01_pmc_traffic.json $R/gpurun_out/pmc_bench_fetch.txt $R/gpurun_out/pmc_bench_write.txt $R/gpurun_out/profiles/
cp $R/gpurun_out/r01_pmc_traffic.json $R/profiles/r01_pmc_traffic.json
cd $R && python bench.py > $R/gpurun_out/profiles/r01_bench_cfg2_$TAG.json 2> $R/gpurun_out/profiles/bench_stderr.txt
grep -E 'passed|failed|smoke' $R/gpurun_out/profiles/r01_pytest_gpu_$TAG.log; head -12 $R/gpurun_out/profiles/r01_kernel_trace_cfg2_$TAG.txt; cut -c1-1500 $R/gpurun_out/profiles/r01_bench_cfg2_$TAG.json
