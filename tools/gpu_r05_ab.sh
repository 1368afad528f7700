# A/B of the partial-spectrum Rayleigh-Ritz solver inside the driver's bench window (cfg 5, --steps 20 --warmup 5)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
for P in 1 0; do
  DFTK_MI_HEEV_PARTIAL=$P timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-complex-leg --no-parity > $O/r05_ab_partial$P.json 2> $O/r05_ab_partial$P.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_ab_partial$P.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("PARTIAL=$P value", round(d["value"], 4), "ms/step", round(d["ms_per_step"], 2), "frac", round(r["frac"], 3), "heev", r.get("families_ms", {}).get("heev_jacobi"), "E", d["config"].get("E_total"))
print({k: round(v, 1) for k, v in r.get("families_ms", {}).items()})
PY
done
DFTK_MI_HEEV_TRACE=1 timeout 600 python bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-complex-leg --no-parity 2>&1 | grep "heev lowest\]" | head -40 > $O/r05_heev_trace_scf.txt
head -40 $O/r05_heev_trace_scf.txt
