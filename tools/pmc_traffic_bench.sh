# HBM traffic of every kernel family of the benchmark workload (TCC byte counters, separate passes per
# the gfx950 slot limits; --pmc never combined with other trace domains except --kernel-trace).
# Writes gpurun_out/${TAG}_pmc_traffic.json (copy to profiles/ to have bench.py quote it as roofline.traffic).
# usage: TAG=r04 bash tools/pmc_traffic_bench.sh [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r04}
ARGS="--no-cpu-baseline --no-complex-leg --no-parity --no-amdahl-probe --prof-all $@"
# The counters are attributed to families by KERNEL NAME; the partial Rayleigh-Ritz solver (eig_kernels.hip) runs its own
# small products on the same zgemm kernels without booking them as zgemm calls (they are booked as heev time).  For the
# traffic passes the full Jacobi takes the Rayleigh-Ritz step, so that every k_zgemm dispatch the counters see is one of the
# booked LOBPCG / projector products (the zgemm kernels themselves are the same code either way).
export DFTK_MI_HEEV_PARTIAL=0
export PMC_NOTE="; Rayleigh-Ritz by the full Jacobi in these passes (DFTK_MI_HEEV_PARTIAL=0) so that all k_zgemm dispatches are booked zgemm calls"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf --output-format csv -- python $R/bench.py $ARGS > /tmp/bench_f.json 2>/dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pw --output-format csv -- python $R/bench.py $ARGS > /tmp/bench_w.json 2>/dev/null
python $R/tools/pmc_to_traffic.py /tmp/pf/pf_counter_collection.csv /tmp/pw/pw_counter_collection.csv /tmp/bench_f.json $R/gpurun_out/${TAG}_pmc_traffic.json
python $R/tools/pmc_summary.py /tmp/pf/pf_counter_collection.csv FETCH_SIZE 24 > $R/gpurun_out/${TAG}_pmc_fetch_size.txt
python $R/tools/pmc_summary.py /tmp/pw/pw_counter_collection.csv WRITE_SIZE 24 > $R/gpurun_out/${TAG}_pmc_write_size.txt
cat $R/gpurun_out/${TAG}_pmc_traffic.json
