# GPU call A of round 2: new parity / building-block / sharding tests, then the north-star bench + kernel trace.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_lobpcg_blocks.py tests/test_gpu_multirank.py tests/test_gpu_baseline_parity.py \
  "tests/test_gpu_scf.py::test_hamiltonian_blocks_own_their_potential" -q --durations=15 -p no:cacheprovider 2>&1 \
  | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path' | tail -150 > $O/pytest_new.log
tail -5 $O/pytest_new.log
timeout 600 python bench.py > $O/bench_cfg5.json 2> $O/bench_cfg5.err
cut -c1-3000 $O/bench_cfg5.json; tail -3 $O/bench_cfg5.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o p -- python $R/bench.py --no-cpu-baseline > /tmp/bench_kt.json 2>/tmp/bench_kt.err
python $R/tools/kernel_stats_txt.py /tmp/kt/p_kernel_stats.csv 40 > $O/kernel_trace_cfg5.txt
tail -1 /tmp/bench_kt.json >> $O/kernel_trace_cfg5.txt
head -30 $O/kernel_trace_cfg5.txt
