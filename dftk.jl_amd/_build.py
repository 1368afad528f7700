"""Build the HIP library in-tree: hipcc --offload-arch=gfx950 -> dftk.jl_amd/lib/libdftk_mi355x.so."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libdftk_mi355x.so")
SOURCES = ["api.cpp", "comm.cpp", "lobpcg.cpp", "batch.cpp", "batch_kernels.hip", "fft_kernels.hip", "gemm_kernels.hip", "dense_kernels.hip", "eig_kernels.hip", "xc_kernels.hip", "setup_kernels.hip", "gamma_kernels.hip", "cube_kernels.hip", "mix_kernels.hip"]


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


def _tu_hash(path: str, flags: list[str]) -> str:
    import hashlib
    h = hashlib.sha256(" ".join(flags).encode())
    for d in [path, os.path.join(CSRC, "common.h"), os.path.join(CSRC, "batch.h"),
              os.path.join(os.path.dirname(HERE), "include", "dftk_mi355x.h")]:
        with open(d, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 (cross-compiles without a GPU).  One object per
    translation unit, compiled in parallel and cached by content hash under ``lib/obj/`` (git-ignored), then
    linked: a one-file edit rebuilds in seconds."""
    if not force and not needs_build():
        return LIBPATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libdftk_mi355x.so")
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    sh = source_hash()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
             # keep MFMA accumulators in VGPRs: without it hipcc (ROCm 7.2) shuttles every loop-carried
             # accumulator VGPR<->AGPR around each k-step (256 v_accvgpr moves per 32 f64 MFMAs)
             "-mllvm", "-amdgpu-mfma-vgpr-form=1"]

    def compile_one(f: str) -> str:
        src = os.path.join(CSRC, f)
        # the source hash is compiled into api.cpp only (dftk_mi_version); the other objects do not depend on it
        extra = [f'-DDFTK_MI_SRC_HASH="{sh}"'] if f == "api.cpp" else []
        obj = os.path.join(objdir, f"{f}.{_tu_hash(src, flags + extra)}.o")
        if force or not os.path.exists(obj):
            for stale in os.listdir(objdir):
                if stale.startswith(f + "."):
                    os.remove(os.path.join(objdir, stale))
            cmd = [hipcc] + flags + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError(f"hipcc failed on {f}:\n" + res.stdout + res.stderr)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIBPATH] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    with open(HASHPATH, "w") as fh:
        fh.write(sh + "\n")
    return LIBPATH


ABI_CHECK_SRC = os.path.join(os.path.dirname(HERE), "tools", "abi_c_check.c")
ABI_CHECK_BIN = os.path.join(os.path.dirname(HERE), "tools", "bin", "abi_c_check")


def build_abi_check(force: bool = False) -> str:
    """tools/abi_c_check.c -> tools/bin/abi_c_check: the plain-C consumer of include/dftk_mi355x.h, compiled by gcc
    (-std=c99 -pedantic: the header must be C, not C++) and linked against the in-tree library (rpath relative to the
    binary, so the pair travels to the GPU box)."""
    build()
    src_time = max(os.path.getmtime(p) for p in (ABI_CHECK_SRC, LIBPATH,
                                                  os.path.join(os.path.dirname(HERE), "include", "dftk_mi355x.h")))
    if not force and os.path.exists(ABI_CHECK_BIN) and os.path.getmtime(ABI_CHECK_BIN) >= src_time:
        return ABI_CHECK_BIN
    os.makedirs(os.path.dirname(ABI_CHECK_BIN), exist_ok=True)
    rocm_lib = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1",
           "-I", os.path.join(os.path.dirname(HERE), "include"), ABI_CHECK_SRC, "-o", ABI_CHECK_BIN,
           "-L", LIBDIR, "-ldftk_mi355x", "-L", rocm_lib, "-lamdhip64", "-lm",
           "-Wl,-rpath,$ORIGIN/../../dftk.jl_amd/lib", f"-Wl,-rpath,{rocm_lib}"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("gcc failed on tools/abi_c_check.c:\n" + res.stdout + res.stderr)
    return ABI_CHECK_BIN


HOST_CHECK_SRC = os.path.join(os.path.dirname(HERE), "tools", "host_ortho_check.cpp")
HOST_CHECK_BIN = os.path.join(os.path.dirname(HERE), "tools", "bin", "host_ortho_check")


def build_host_ortho_check(force: bool = False) -> str:
    """tools/host_ortho_check.cpp -> tools/bin/host_ortho_check: the host-only check of the LOBPCG driver's
    ``host_ortho_small`` (the translation unit includes csrc/lobpcg.cpp -- the function sits in its anonymous namespace --
    and links the rest from the in-tree library).  Runs without a GPU."""
    build()
    lobpcg_src = os.path.join(CSRC, "lobpcg.cpp")
    src_time = max(os.path.getmtime(p) for p in (HOST_CHECK_SRC, LIBPATH, lobpcg_src))
    if not force and os.path.exists(HOST_CHECK_BIN) and os.path.getmtime(HOST_CHECK_BIN) >= src_time:
        return HOST_CHECK_BIN
    os.makedirs(os.path.dirname(HOST_CHECK_BIN), exist_ok=True)
    cmd = [shutil.which("hipcc") or "/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-x", "hip", "--offload-arch=gfx950", "-I", CSRC, HOST_CHECK_SRC, "-o", HOST_CHECK_BIN,
           "-L", LIBDIR, "-ldftk_mi355x", "-Wl,-rpath,$ORIGIN/../../dftk.jl_amd/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed on tools/host_ortho_check.cpp:\n" + res.stdout + res.stderr)
    return HOST_CHECK_BIN


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_abi_check(force=True))
    print(build_host_ortho_check(force=True))
