R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02l
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q --tb=short -p no:cacheprovider 2>&1 | tail -5 | tee $O/pytest.log
python tools/fft_bench.py 5 64 2>&1 | grep -v amdgpu | tee $O/fft_direct.log
DFTK_MI_ZPASS_CLASSIC=1 python tools/fft_bench.py 5 64 2>&1 | grep -v amdgpu | tee $O/fft_classic.log
python tools/fft_bench.py 4 64 2>&1 | grep -v amdgpu | tee -a $O/fft_direct.log
DFTK_MI_ZPASS_CLASSIC=1 python tools/fft_bench.py 4 64 2>&1 | grep -v amdgpu | tee -a $O/fft_classic.log
