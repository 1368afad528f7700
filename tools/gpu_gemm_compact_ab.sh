# A/B of the compact UPPER launches (live tiles only, contiguous XCD blocks) against the full-grid mappings.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gemm_compact
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gamma_real.py tests/test_gpu_lobpcg_blocks.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -5 | tee $O/pytest.log
run() { echo "== $*"; env "$@" timeout 200 python tools/gemm_real_bench.py 264859 503 struct 2>&1 | grep -B1 "flags=1" | grep -v "^--"; }
(
run DFTK_MI_GEMM_NO_COMPACT=1
run X=1
for NS in 3 6 7 8 13 14 20 24 25 26 50; do run DFTK_MI_GEMM_FORCE_NS=$NS DFTK_MI_GEMM_FORCE_MODE=2; done
) 2>&1 | tee $O/compact_ab.txt
