# Round 4, call D: switch-path tests, C consumer, and why the 72-k-point step of tools/kpoints_share_profile.py (90 ms) is
# three times the bench's (30 ms)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_d
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_switches.py tests/test_gpu_multirank.py::test_c_consumer_of_the_header_drives_rccl_shard_apply_density_lobpcg tests/test_gpu_kernels.py tests/test_gpu_kbatch.py tests/test_gpu_lobpcg_blocks.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 \
  | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path' > $O/pytest_gpu.log
tail -30 $O/pytest_gpu.log
DFTK_MI_KBATCH_TRACE=1 timeout 600 python tools/kpoints_share_profile.py 1 > $O/kshare_1.txt 2> $O/kshare_1.err
cat $O/kshare_1.txt; grep kbatch $O/kshare_1.err | tail -12
DFTK_MI_BENCH_STEP_TIMERS=1 DFTK_MI_KBATCH_TRACE=1 timeout 600 python bench.py --mode kpoints --system al --no-cpu-baseline --no-amdahl-probe --no-parity > $O/bench_al.json 2> $O/bench_al.err
grep "^\[step" $O/bench_al.err | tail -6; grep kbatch $O/bench_al.err | tail -12
python -c "
import json; d=json.loads(open('$O/bench_al.json').read().strip().splitlines()[-1]); print(d['value'], d['steps'], d['config']['scf_wall_s'], d['config']['step_wall_s'])"
