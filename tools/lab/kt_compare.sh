cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in prev base; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o p -- $R/tools/lab/bin/zgemm_lab_$v N 135491 259 259 C 259 259 135491 > /dev/null 2>&1
echo == $v; python $R/tools/kernel_stats_txt.py /tmp/kt_$v/p_kernel_stats.csv 8 | grep zgemm
done
