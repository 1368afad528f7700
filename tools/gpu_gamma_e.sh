#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gamma_real.py tests/test_gpu_kernels.py -x -q -k "heev or lobpcg or scf" 2>&1 | tail -8
python tools/heev_bench.py 503 1006 1509 2>&1 | grep -v amdgpu; DFTK_MI_HEEV_TRACE=1 python tools/heev_bench.py real 503 1006 1509 2>&1 | grep -v amdgpu
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_cfg5_real.json 2> gpurun_out/bench_cfg5_real.err
tail -c 300 gpurun_out/bench_cfg5_real.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_cfg5_real.json").read().strip().splitlines()[-1])
    print("REAL:", d["value"], d["steps"], d["config"]["scf_wall_s"], d["config"]["E_total"], d["roofline"]["achieved"], d["roofline"]["frac"],
          d["roofline"]["families_ms"], d["config"]["step_wall_s"])
except Exception as e:
    print("bench real failed", e)
PY
