#!/bin/bash
# TLB / L2 / TCP counters of the REAL and 3M products at the cfg-5 shapes: K-major ('C', Gram) against M-major ('N', block
# update) operand paths.  Three --pmc passes (never combined with other trace domains except --kernel-trace).
# Output: gpurun_out/pmc_gemm_tlb.txt  (profiles/r02_pmc_tlb_l2_gemm_real_vs_3m.txt is a copy with derived rates)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gemm_tlb.txt
: > $O
pass() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/$name -o $name --output-format csv -- python $R/tools/gemm_real_bench.py 264859 503 gram > /tmp/$name.log 2>&1
  echo "# pass $name: $*" >> $O
  python $R/tools/pmc_summary.py /tmp/$name/${name}_counter_collection.csv $1 6 >> $O || tail -5 /tmp/$name.log >> $O
}
pass pa TCP_TCC_READ_REQ TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TCC_READ_REQ_LATENCY
pass pb TCC_REQ TCC_HIT TCC_MISS TCC_EA0_RDREQ
pass pc TCP_GATE_EN1 TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TA_TCP_STATE_READ
cat $O
