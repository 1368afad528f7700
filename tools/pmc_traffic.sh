# HBM traffic of the zgemm kernels (TCC byte counters; FETCH_SIZE and WRITE_SIZE need separate passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="c N 135491 259 259 C 259 259 135491 C 640 259 135491 N 135491 259 640"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf --output-format csv -- $R/tools/lab/bin/zgemm_lab_base $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pw --output-format csv -- $R/tools/lab/bin/zgemm_lab_base $ARGS > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pf/pf_counter_collection.csv FETCH_SIZE 12 | tee $R/gpurun_out/pmc_fetch.txt
python $R/tools/pmc_summary.py /tmp/pw/pw_counter_collection.csv WRITE_SIZE 12 | tee $R/gpurun_out/pmc_write.txt
