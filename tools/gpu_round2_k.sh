R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02k
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_baseline_parity.py -q --tb=short --durations=5 -p no:cacheprovider 2>&1 \
  | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path' > $O/pytest_parity.log
tail -12 $O/pytest_parity.log
bash tools/gpu_round2_profiles.sh v3 2>&1 | tail -30
