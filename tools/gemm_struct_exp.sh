cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_gemm
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gamma_real.py -x -q -k "zgemm or gemm" 2>&1 | tail -3
echo "== default (rotation on)"; python tools/gemm_real_bench.py 264859 503 struct 2>&1 | grep -v amdgpu | tee gpurun_out/r03_gemm/struct_rot.txt
echo "== DFTK_MI_GEMM_NO_ZMAJOR=1 (rotation on)"; DFTK_MI_GEMM_NO_ZMAJOR=1 python tools/gemm_real_bench.py 264859 503 struct 2>&1 | grep -v amdgpu | grep -A1 "flags=1" | tee gpurun_out/r03_gemm/struct_rot_nozm.txt
echo "== DFTK_MI_GEMM_NO_ROT=1"; DFTK_MI_GEMM_NO_ROT=1 python tools/gemm_real_bench.py 264859 503 struct 2>&1 | grep -v amdgpu | grep "flags=1"
