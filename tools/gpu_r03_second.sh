# Round-3 second pass: the new bench line (both algorithms + timed CPU step), idle-gap analysis of the kernel trace
# (cfg 5 and cfg 3), heev / fft micro-bench tables
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_second
mkdir -p $O
cd $R
( time timeout 1200 python bench.py > $O/bench_cfg5_default.json 2> $O/bench_cfg5_default.err ) 2> $O/bench_cfg5_default.time
cut -c1-300 $O/bench_cfg5_default.json; tail -3 $O/bench_cfg5_default.time; tail -3 $O/bench_cfg5_default.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o p -- python $R/bench.py --no-cpu-baseline --no-complex-leg > /tmp/bench_kt.json 2>/tmp/bench_kt.err
python $R/tools/kernel_stats_txt.py /tmp/kt/p_kernel_stats.csv 40 > $O/kernel_trace_cfg5.txt
python $R/tools/trace_gaps.py /tmp/kt/p_kernel_trace.csv 30 0.5 > $O/trace_gaps_cfg5_second_half.txt
head -40 $O/trace_gaps_cfg5_second_half.txt
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -o p -- python $R/bench.py --mode kpoints --no-cpu-baseline > $O/bench_cfg3_traced.json 2>/tmp/bench_kt3.err
python $R/tools/kernel_stats_txt.py /tmp/kt3/p_kernel_stats.csv 30 > $O/kernel_trace_cfg3.txt
python $R/tools/trace_gaps.py /tmp/kt3/p_kernel_trace.csv 25 0.5 > $O/trace_gaps_cfg3_second_half.txt
head -30 $O/trace_gaps_cfg3_second_half.txt
cd $R
timeout 300 python tools/heev_bench.py > $O/heev_bench.txt 2>&1; tail -12 $O/heev_bench.txt
