# SQ counters of the z kernels at 192^3, 8 bands per launch: the register-resident four-step kernels (default) against the
# LDS-pass kernels (DFTK_MI_FFT_REG=0).  Two counter passes each (--pmc only with --kernel-trace).  Output:
# gpurun_out/r03_pmc_fft_zkernels.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pmc_fft_zkernels.txt
: > $O
for REG in 1 0; do
  echo "== DFTK_MI_FFT_REG=$REG" >> $O
  DFTK_MI_FFT_REG=$REG rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/pz$REG -o p --output-format csv -- python $R/tools/fft_bench.py 5 16 > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/pz$REG/p_counter_collection.csv SQ_WAVE_CYCLES 12 | grep "k_zpass\|k_zdensity" >> $O
  DFTK_MI_FFT_REG=$REG rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d /tmp/qz$REG -o q --output-format csv -- python $R/tools/fft_bench.py 5 16 > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/qz$REG/q_counter_collection.csv SQ_WAIT_ANY 12 | grep "k_zpass\|k_zdensity" >> $O
done
cat $O
