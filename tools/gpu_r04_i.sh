#!/bin/bash
mkdir -p gpurun_out/r04_j
timeout 600 python tools/gemm_shortk_bench.py > gpurun_out/r04_j/gemm_shortk.txt 2>&1
cat gpurun_out/r04_j/gemm_shortk.txt
