#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_gamma_real.py -x -q -k aligns 2>&1 | tail -40
