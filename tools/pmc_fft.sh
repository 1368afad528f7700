cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d /tmp/pmcf -o pmcf --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmcf/pmcf_counter_collection.csv SQ_LDS_IDX_ACTIVE 12 | grep "k_z\|k_y\|k_x" | tee $R/gpurun_out/pmc_fft_summary.txt
