# Final validation of the round: the driver's three steps (pytest -m gpu, smoke, bench) on the committed tree.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_final
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --tb=short --durations=8 -p no:cacheprovider 2>&1 \
  | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path' > $O/r02_pytest_gpu.log
tail -16 $O/r02_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/r02_pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r02_bench_cfg5_driver_args.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_final/r02_bench_cfg5_driver_args.json"))
r = d["roofline"]
print(d["value"], d["steps"], d["warmup"], d["config"]["scf_wall_s"], r["frac"], r["traffic"], r.get("traffic_source"), d["cpu_baseline"]["value"])
PY
