#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY -d /tmp/pa -o pa --output-format csv -- python $R/tools/gemm_real_bench.py 264859 503 gram > /tmp/pa.log 2>&1
python $R/tools/pmc_summary.py /tmp/pa/pa_counter_collection.csv TCP_TCC_READ_REQ 6 || tail -5 /tmp/pa.log
rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_REQ -d /tmp/pb -o pb --output-format csv -- python $R/tools/gemm_real_bench.py 264859 503 gram > /tmp/pb.log 2>&1
python $R/tools/pmc_summary.py /tmp/pb/pb_counter_collection.csv TCC_REQ 6 || tail -5 /tmp/pb.log
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TA_TCP_STATE_READ TCP_GATE_EN1 -d /tmp/pc -o pc --output-format csv -- python $R/tools/gemm_real_bench.py 264859 503 gram > /tmp/pc.log 2>&1
python $R/tools/pmc_summary.py /tmp/pc/pc_counter_collection.csv TCP_GATE_EN1 6 || tail -5 /tmp/pc.log
