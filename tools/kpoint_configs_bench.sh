# the three k-point BASELINE configs, lock-step batched vs the round-2 lane pool
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_kcfg
for S in si al graphene; do
  for KB in 1 0; do
    DFTK_MI_KBATCH=$KB python bench.py --mode kpoints --system $S --no-cpu-baseline > gpurun_out/r03_kcfg/bench_${S}_kbatch${KB}.json 2> gpurun_out/r03_kcfg/bench_${S}_kbatch${KB}.err
    python -c "
import json; d=json.loads(open('gpurun_out/r03_kcfg/bench_${S}_kbatch${KB}.json').read().strip().splitlines()[-1]); c=d['config']
print('$S', 'kbatch=$KB', round(d['value'],2), 'it/s', d['steps'], 'steps', c['scf_wall_s'], 's', 'E', c['E_total'], c['workload'][:110])" || tail -3 gpurun_out/r03_kcfg/bench_${S}_kbatch${KB}.err
  done
done
