#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_gamma_real.py tests/test_gpu_kernels.py -x -q -k "zgemm or lobpcg or potrf" 2>&1 | tail -4
python tools/gemm_real_bench.py 264859 503 gramscan 2>&1 | grep -v amdgpu
python tools/gemm_real_bench.py 264859 503 2>&1 | grep -v amdgpu
