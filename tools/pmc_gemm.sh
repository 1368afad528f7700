cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="N 135491 259 2072 C 256 256 135491"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d $R/gpurun_out/pmc2 -o pmc2 --output-format csv -- $R/tools/lab/bin/zgemm_lab_base $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES -d $R/gpurun_out/pmc3 -o pmc3 --output-format csv -- $R/tools/lab/bin/zgemm_lab_base $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES -d $R/gpurun_out/pmc4 -o pmc4 --output-format csv -- $R/tools/lab/bin/zgemm_lab_base $ARGS > /dev/null 2>&1
ls $R/gpurun_out/pmc2 $R/gpurun_out/pmc3 $R/gpurun_out/pmc4
