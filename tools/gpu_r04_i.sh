#!/bin/bash
mkdir -p gpurun_out/r04_i
timeout 600 python tools/ew_bench.py > gpurun_out/r04_i/ew_bench.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lobpcg_blocks.py tests/test_gpu_gamma_real.py tests/test_gpu_scf.py tests/test_gpu_kbatch.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04_i/pytest.txt
timeout 600 python tools/late_step_profile.py > gpurun_out/r04_i/late_step.txt 2>&1
cat gpurun_out/r04_i/*.txt
