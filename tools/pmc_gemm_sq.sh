#!/bin/bash
# SQ counters of the REAL vs 3M products at the cfg-5 shapes
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA -d /tmp/p1 -o p1 --output-format csv -- python $R/tools/gemm_real_bench.py 264859 503 gram > $R/gpurun_out/pmc_gram_run1.log 2>&1
python $R/tools/pmc_summary.py /tmp/p1/p1_counter_collection.csv SQ_BUSY_CYCLES 8 > $R/gpurun_out/pmc_gram_sq.txt
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU -d /tmp/p2 -o p2 --output-format csv -- python $R/tools/gemm_real_bench.py 264859 503 gram > $R/gpurun_out/pmc_gram_run2.log 2>&1
python $R/tools/pmc_summary.py /tmp/p2/p2_counter_collection.csv SQ_WAVE_CYCLES 8 > $R/gpurun_out/pmc_gram_lds.txt
cat $R/gpurun_out/pmc_gram_sq.txt; echo; cat $R/gpurun_out/pmc_gram_lds.txt; tail -3 $R/gpurun_out/pmc_gram_run2.log
python $R/tools/kernel_stats_txt.py /tmp/p1/p1_kernel_trace.csv 2>/dev/null | head -12
