# kernel-level breakdown of dftk_mi_heev_lowest against the full Jacobi (rocprofv3 --kernel-trace --stats)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for N in 1006 1509; do
  rm -rf /tmp/kt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o p -- python $R/tools/heev_bench.py lowest $N > /tmp/hb.txt 2>/tmp/hb.err
  python $R/tools/kernel_stats_txt.py /tmp/kt/p_kernel_stats.csv 30 > $O/r05_heev_lowest_kernels_$N.txt
  cat /tmp/hb.txt >> $O/r05_heev_lowest_kernels_$N.txt
  head -24 $O/r05_heev_lowest_kernels_$N.txt
done
