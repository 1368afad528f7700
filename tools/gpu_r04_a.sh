# Round 4, call A: new GPU tests (C consumer of the header, tight multirank criterion) + FFT batch-size sweep
# (does a T1/T2 working set that fits the 256 MiB Infinity Cache change the stage rates?)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_kbatch.py -m gpu -q --tb=short -x -p no:cacheprovider 2>&1 | grep -v '^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl' | tail -15 > $O/pytest_multirank.log
tail -5 $O/pytest_multirank.log
for B in 1 2 3 4 8 16; do
  echo "== bands per launch group = $B" >> $O/fft_batch_sweep.txt
  timeout 300 python tools/fft_bench.py 5 64 $B 2>&1 | grep -v amdgpu >> $O/fft_batch_sweep.txt
done
cat $O/fft_batch_sweep.txt
cat $O/fft_batch_sweep.txt
