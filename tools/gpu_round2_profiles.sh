# Round-2 measurement pass: north-star bench, kernel trace, PMC traffic (separate passes), secondary configs.
R=$GRAFT_REPO_ROOT
TAG=${1:-v2}
O=$R/gpurun_out/profiles_$TAG
mkdir -p $O
cd $R
timeout 900 python bench.py > $O/r02_bench_cfg5_$TAG.json 2> $O/bench_cfg5.err
cut -c1-600 $O/r02_bench_cfg5_$TAG.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o p -- python $R/bench.py --no-cpu-baseline > /tmp/bench_kt.json 2>/tmp/bench_kt.err
python $R/tools/kernel_stats_txt.py /tmp/kt/p_kernel_stats.csv 40 > $O/r02_kernel_trace_cfg5_$TAG.txt
tail -1 /tmp/bench_kt.json >> $O/r02_kernel_trace_cfg5_$TAG.txt
head -14 $O/r02_kernel_trace_cfg5_$TAG.txt
timeout 1500 bash $R/tools/pmc_traffic_bench.sh > $O/pmc.log 2>&1
cp $R/gpurun_out/r02_pmc_traffic.json $R/gpurun_out/r02_pmc_fetch_size.txt $R/gpurun_out/r02_pmc_write_size.txt $O/ 2>/dev/null
tail -5 $O/pmc.log
cd $R
# the reference's own iteration (general complex orbitals at Gamma) on the same cell, for comparison
timeout 900 python bench.py --no-gamma-real --no-cpu-baseline > $O/r02_bench_cfg5_complex_$TAG.json 2> $O/bench_cfg5_complex.err
cut -c1-300 $O/r02_bench_cfg5_complex_$TAG.json
timeout 600 python bench.py --supercell 4 --no-cpu-baseline > $O/r02_bench_cfg2_$TAG.json 2> $O/bench_cfg2.err
cut -c1-400 $O/r02_bench_cfg2_$TAG.json
timeout 600 python bench.py --mode kpoints --no-cpu-baseline > $O/r02_bench_cfg3_kpoints_$TAG.json 2> $O/bench_cfg3.err
cut -c1-700 $O/r02_bench_cfg3_kpoints_$TAG.json
