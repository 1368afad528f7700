# Round 4, call B: full GPU suite on the new host_fetch / Anderson / bench code, bench with parity + Amdahl (gamma, kpoints),
# A/B of the zero-copy fetch
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_b
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short --durations=6 -p no:cacheprovider 2>&1 \
  | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path' > $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
./tools/bin/abi_c_check 1 2>&1 | tail -3
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_cfg5_driver_args.json 2> $O/bench_driver.err ) 2> $O/bench_driver.time
tail -3 $O/bench_driver.err; cat $O/bench_driver.time
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_b/bench_cfg5_driver_args.json").read().strip().splitlines()[-1])
print(round(d["value"], 3), d["steps"], d["config"]["scf_wall_s"], round(d["roofline"]["frac"], 3))
print(json.dumps(d["config"]["parity"], indent=1)[:3000])
print(json.dumps(d["amdahl"], indent=1)[:3000])
PY
timeout 900 python bench.py --mode kpoints --system al --no-cpu-baseline > $O/bench_kpoints_al.json 2> $O/bench_kpoints_al.err
tail -2 $O/bench_kpoints_al.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_b/bench_kpoints_al.json").read().strip().splitlines()[-1])
print(round(d["value"], 3), d["steps"], d["config"]["scf_wall_s"])
print(json.dumps(d["amdahl"], indent=1)[:3000])
PY
timeout 300 python tools/late_step_profile.py 5 8 6 > $O/late_step_zero_copy.txt 2>/dev/null
DFTK_MI_FETCH_BLIT=1 timeout 300 python tools/late_step_profile.py 5 8 6 > $O/late_step_blit.txt 2>/dev/null
head -15 $O/late_step_zero_copy.txt; head -15 $O/late_step_blit.txt
