R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_lobpcg_blocks.py -q --tb=short -p no:cacheprovider 2>&1 \
  | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path' > $O/pytest.log
tail -60 $O/pytest.log
