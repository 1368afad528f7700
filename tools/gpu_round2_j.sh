R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02j
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_scf.py tests/test_gpu_symmetry.py -q --tb=short -p no:cacheprovider 2>&1 \
  | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path' > $O/pytest_gpu2.log
tail -30 $O/pytest_gpu2.log
