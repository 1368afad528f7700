"""Build the HIP library in-tree: hipcc --offload-arch=gfx950 -> dftk.jl_amd/lib/libdftk_mi355x.so."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libdftk_mi355x.so")
SOURCES = ["api.cpp", "comm.cpp", "lobpcg.cpp", "batch.cpp", "batch_kernels.hip", "fft_kernels.hip", "gemm_kernels.hip", "dense_kernels.hip", "xc_kernels.hip", "setup_kernels.hip", "gamma_kernels.hip", "cube_kernels.hip"]


HASHPATH = LIBPATH + ".srchash"


def source_hash() -> str:
    """sha256 over every source the library is built from (order fixed).  The hash is compiled into the
    library (``dftk_mi_version()`` ends with ``src=<hash>``) and written next to it, so that a stale binary
    -- the .so is git-ignored but ships to the GPU box -- is never used silently, whatever its mtime says."""
    import hashlib
    h = hashlib.sha256()
    deps = [os.path.join(CSRC, f) for f in sorted(SOURCES + ["common.h", "batch.h"])]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "dftk_mi355x.h"))
    for d in deps:
        with open(d, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def needs_build() -> bool:
    if not os.path.exists(LIBPATH) or not os.path.exists(HASHPATH):
        return True
    with open(HASHPATH) as fh:
        return fh.read().strip() != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIBPATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libdftk_mi355x.so")
    os.makedirs(LIBDIR, exist_ok=True)
    sh = source_hash()
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           f'-DDFTK_MI_SRC_HASH="{sh}"',
           # keep MFMA accumulators in VGPRs: without it hipcc (ROCm 7.2) shuttles every loop-carried
           # accumulator VGPR<->AGPR around each k-step (256 v_accvgpr moves per 32 f64 MFMAs)
           "-mllvm", "-amdgpu-mfma-vgpr-form=1",
           "-o", LIBPATH] + [os.path.join(CSRC, f) for f in SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    with open(HASHPATH, "w") as fh:
        fh.write(sh + "\n")
    return LIBPATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
