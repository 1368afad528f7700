cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_final; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --tb=short --durations=8 -p no:cacheprovider 2>&1 | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path' > $O/r06_pytest_gpu.log
tail -14 $O/r06_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/r06_pytest_gpu.log
./tools/bin/abi_c_check 1 2>&1 | tail -3 | tee -a $O/r06_pytest_gpu.log
