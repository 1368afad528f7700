R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
DFTK_MI_HEEV_TRACE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-complex-leg --no-parity 2>&1 | grep "heev lowest timing\|heev real\] n=[0-9]* sweep" > $O/r05_heev_trace_scf.txt
grep timing $O/r05_heev_trace_scf.txt | head -50
grep -c sweep $O/r05_heev_trace_scf.txt
