#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -60 > gpurun_out/gamma_fulltests.log
tail -30 gpurun_out/gamma_fulltests.log
