R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --tb=short --durations=10 -p no:cacheprovider 2>&1 \
  | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path' > $O/pytest_gpu.log
tail -30 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench.err; cut -c1-900 $O/bench_cfg5.json
timeout 600 python bench.py --mode kpoints --no-cpu-baseline > $O/bench_cfg3.json 2>> $O/bench.err; cut -c1-300 $O/bench_cfg3.json
