R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_kbatch
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_kbatch.py -x -q 2>&1 | tail -15
python bench.py --mode kpoints --no-cpu-baseline > $O/bench_cfg3_kbatch.json 2> $O/bench_cfg3_kbatch.err
python -c "
import json; d=json.loads(open('$O/bench_cfg3_kbatch.json').read().strip().splitlines()[-1]); c=d['config']
print('kbatch', d['value'], d['steps'], c['scf_wall_s'], c['E_total'], c['host_timers_ms_per_step'], c['step_wall_s'])"
tail -3 $O/bench_cfg3_kbatch.err
DFTK_MI_KBATCH=0 python bench.py --mode kpoints --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('lanes ', d['value'], d['steps'], c['scf_wall_s'], c['E_total'], c['host_timers_ms_per_step'])"
