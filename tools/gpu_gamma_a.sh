#!/bin/bash
# first GPU contact of the Gamma-real path: its tests, the product microbenchmark, the cfg-5 SCF both ways
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_gamma_real.py -x -q 2>&1 | tail -40 > gpurun_out/gamma_tests.log
tail -5 gpurun_out/gamma_tests.log
timeout 300 python tools/gemm_real_bench.py > gpurun_out/gemm_real_bench.log 2>&1
DFTK_MI_GEMM_PAD_LDS=24576 timeout 300 python tools/gemm_real_bench.py > gpurun_out/gemm_real_bench_pad.log 2>&1
cat gpurun_out/gemm_real_bench.log; echo "--- 2 workgroups per CU (LDS pad)"; cat gpurun_out/gemm_real_bench_pad.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_cfg5_real.json 2> gpurun_out/bench_cfg5_real.err
tail -c 600 gpurun_out/bench_cfg5_real.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_cfg5_real.json").read().strip().splitlines()[-1])
    print("REAL:", d["value"], d["steps"], d["config"]["scf_wall_s"], d["config"]["E_total"], d["roofline"]["achieved"], d["roofline"]["frac"],
          d["roofline"]["families_ms"], d["config"]["lobpcg_iters_per_step"])
except Exception as e:
    print("bench real failed", e)
PY
