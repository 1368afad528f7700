# full GPU suite + smoke + the default bench line (driver arguments), timed
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_check
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --tb=short --durations=6 -p no:cacheprovider 2>&1 \
  | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path' > $O/pytest_gpu.log
tail -12 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench.err ) 2> $O/bench.time
tail -3 $O/bench.time
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_check/bench_driver_args.json").read().strip().splitlines()[-1])
r = d["roofline"]; c = d["config"]
print(d["value"], d["steps"], d["warmup"], c["scf_wall_s"], c["converged"], r["frac"], r["traffic"])
print({k: v for k, v in c["complex_iteration"].items() if k in ("value", "steps", "scf_wall_s", "dE_total_vs_real")})
cb = d["cpu_baseline"]; print(cb["kind"], cb["value"], cb.get("timed_step", {}))
PY
