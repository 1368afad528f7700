#!/bin/bash
mkdir -p gpurun_out/r04_k
timeout 600 python bench.py --mode kpoints --system al --no-cpu-baseline > gpurun_out/r04_k/al.json 2> gpurun_out/r04_k/al.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_k/al.json").read().strip().splitlines()[-1])
c = d["config"]
print(round(d["value"], 2), d["steps"], c["scf_wall_s"], c["E_total"], c["step_wall_s"], c["host_timers_ms_per_step"], (d.get("amdahl") or {}).get("predicted_speedup"), (d.get("amdahl") or {}).get("measured_share_step_ms"), (c.get("parity") or {}).get("pass"))
PY
timeout 900 python -m pytest tests/test_gpu_kbatch.py tests/test_gpu_scf.py tests/test_gpu_symmetry.py tests/test_gpu_mixing.py -m gpu -x -q 2>&1 | tail -3
