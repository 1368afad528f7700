cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="N 135491 256 2048 C 256 256 135491"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA -d /tmp/pw -o pw --output-format csv -- $R/tools/lab/bin/zgemm_lab_base $ARGS > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pw/pw_counter_collection.csv SQ_BUSY_CYCLES 4
