# HBM traffic of the zgemm kernels on single shapes (TCC byte counters; FETCH_SIZE / WRITE_SIZE in separate passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS=${ARGS:-"c C 259 259 135491 C 640 259 135491"}
$R/tools/lab/bin/zgemm_lab_${LABV:-base} $ARGS
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf --output-format csv -- $R/tools/lab/bin/zgemm_lab_${LABV:-base} $ARGS > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pf/pf_counter_collection.csv FETCH_SIZE 12
