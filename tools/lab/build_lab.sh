#!/bin/bash
# build_lab.sh <name> [extra -D flags]: builds tools/lab/bin/zgemm_lab_<name> from the library sources + harness
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
mkdir -p tools/lab/bin
S=dftk.jl_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 "$@" \
  -o tools/lab/bin/zgemm_lab_$name -x hip tools/lab/zgemm_lab.cpp -x hip $S/api.cpp -x hip $S/comm.cpp -x hip $S/lobpcg.cpp \
  $S/fft_kernels.hip $S/gemm_kernels.hip $S/dense_kernels.hip -ldl
