cd $GRAFT_REPO_ROOT
for W in 16 36 72; do DFTK_MI_KBATCH_CHAIN=$W python bench.py --mode kpoints --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('chain', $W, round(d['value'],2), d['steps'], c['scf_wall_s'], c['step_wall_s'], c['lobpcg_iters_per_step'][:3])"; done
DFTK_MI_KBATCH_TRACE=1 python bench.py --mode kpoints --no-cpu-baseline --warmup 0 2>&1 >/dev/null | grep kbatch | head -40
