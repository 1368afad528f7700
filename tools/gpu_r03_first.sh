# Round-3 first pass: the headline bench after the default_diagtolalg fix (library unchanged), both algorithms + kernel trace
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_first
mkdir -p $O
cd $R
timeout 900 python bench.py --no-cpu-baseline > $O/bench_cfg5_real.json 2> $O/bench_cfg5_real.err
cut -c1-400 $O/bench_cfg5_real.json
timeout 900 python bench.py --no-gamma-real --no-cpu-baseline > $O/bench_cfg5_complex.json 2> $O/bench_cfg5_complex.err
cut -c1-300 $O/bench_cfg5_complex.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o p -- python $R/bench.py --no-cpu-baseline > /tmp/bench_kt.json 2>/tmp/bench_kt.err
python $R/tools/kernel_stats_txt.py /tmp/kt/p_kernel_stats.csv 40 > $O/kernel_trace_cfg5.txt
tail -1 /tmp/bench_kt.json >> $O/kernel_trace_cfg5.txt
head -24 $O/kernel_trace_cfg5.txt
cd $R
timeout 600 python bench.py --mode kpoints --no-cpu-baseline > $O/bench_cfg3_kpoints.json 2> $O/bench_cfg3.err
cut -c1-300 $O/bench_cfg3_kpoints.json
