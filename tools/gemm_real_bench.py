"""Microbenchmark of the REAL (half-sphere) products against the 3M complex ones at the LOBPCG shapes of the
1000-electron cell: Gram  G = Y' AY  (UPPER) and block update  X C.   python tools/gemm_real_bench.py [n_G] [M]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd._lib import check, cplx  # noqa: E402

n_G = int(sys.argv[1]) if len(sys.argv) > 1 else 264859
M = int(sys.argv[2]) if len(sys.argv) > 2 else 503
lib = dftk.load_library()
h = C.c_void_p()
check(lib.dftk_mi_basis_create(8, 8, 8, 1.0, 0, C.byref(h)))
nh = (n_G + 1) // 2
g = torch.Generator(device="cuda").manual_seed(0)
Y = torch.randn((3 * M, n_G), dtype=torch.complex128, device="cuda", generator=g)
AY = torch.randn((3 * M, n_G), dtype=torch.complex128, device="cuda", generator=g)
Cm = torch.randn((M, 3 * M), dtype=torch.complex128, device="cuda", generator=g)
G = torch.empty((3 * M, 3 * M), dtype=torch.complex128, device="cuda")
out = torch.empty((M, n_G), dtype=torch.complex128, device="cuda")
torch.cuda.synchronize()


def run(name, trans, m, n, k, A, lda, B, ldb, Cc, ldc, flags, flops, reps=5):
    for it in range(reps + 1):
        if it == 1:
            check(lib.dftk_mi_basis_sync(h))
            t0 = time.time()
        check(lib.dftk_mi_zgemm_ex(h, trans, m, n, k, cplx(1.0), A.data_ptr(), lda, B.data_ptr(), ldb, cplx(0.0),
                                   Cc.data_ptr(), ldc, flags))
    check(lib.dftk_mi_basis_sync(h))
    dt = (time.time() - t0) / reps
    print(f"{name:46s} {dt * 1e3:8.3f} ms  {flops / dt / 1e12:7.2f} TF/s (flops of this formulation)", flush=True)
    return dt


k3 = 3 * M
only = sys.argv[3] if len(sys.argv) > 3 else ""
if only == "gram":      # (PMC passes: just the two Gram formulations and the two updates, 2 launches each)
    run("Gram 3M complex", b"C", k3, k3, n_G, Y, n_G, AY, n_G, G, k3, 1, 8.0 * k3 * (k3 + 1) / 2 * n_G, reps=1)
    run("Gram REAL", b"C", k3, k3, nh, Y, n_G, AY, n_G, G, k3, 1 | 8, 4.0 * k3 * (k3 + 1) / 2 * nh, reps=1)
    run("update 3M complex", b"N", n_G, M, k3, Y, n_G, Cm, k3, out, n_G, 0, 8.0 * n_G * M * k3, reps=1)
    run("update REAL", b"N", nh, M, k3, Y, n_G, Cm, k3, out, n_G, 8, 4.0 * nh * M * k3, reps=1)
    check(lib.dftk_mi_basis_destroy(h))
    sys.exit(0)
if only == "kscan":       # does the K-major path recover when the columns are close together (few pages)?
    for kk in (8192, 16384, 32768, 65536, nh):
        Ys = Y.reshape(-1)[: 3 * M * kk].reshape(3 * M, kk)
        AYs = AY.reshape(-1)[: 3 * M * kk].reshape(3 * M, kk)
        for mm in (k3, 2 * M):
            run(f"Gram REAL m=n={mm} K={kk} (packed: lda=K)", b"C", mm, mm, kk, Ys, kk, AYs, kk, G, mm, 8,
                4.0 * mm * mm * kk, reps=5)
            run(f"Gram REAL m=n={mm} K={kk} (lda=n_G)      ", b"C", mm, mm, kk, Y, n_G, AY, n_G, G, mm, 8,
                4.0 * mm * mm * kk, reps=5)
    check(lib.dftk_mi_basis_destroy(h))
    sys.exit(0)
if only == "upper1006":
    for fl in (0, 1):
        run(f"Gram REAL m=n=1006 flags={fl}", b"C", 2 * M, 2 * M, nh, Y, n_G, AY, n_G, G, 2 * M, fl | 8,
            4.0 * (2 * M * (2 * M + 1) / 2 if fl else 4 * M * M) * nh, reps=3)
    check(lib.dftk_mi_basis_destroy(h))
    sys.exit(0)
if only == "struct":      # the structured launches of a late SCF step (UPPER Gram, triangular X inv(R)) and their plans
    def plan(trans, m, n, k, fl):
        out = (C.c_int * 12)()
        check(lib.dftk_mi_zgemm_plan_host(trans, m, n, k, fl, out))
        v = list(out)
        return f"BN={v[0]} full tiles {v[1]}x{v[2]} nsplit={v[5]} kchunk={v[6]} zmajor={v[7]} shift={v[11]}"
    for mm in (M, 2 * M, 3 * M):
        for fl in (0, 1):
            print("   plan:", plan(b"C", mm, mm, nh, fl | 8))
            run(f"Gram REAL m=n={mm} flags={fl}", b"C", mm, mm, nh, Y, n_G, AY, n_G, G, mm, fl | 8,
                4.0 * (mm * (mm + 1) / 2 if fl else mm * mm) * nh, reps=5)
    for fl in (0, 2):
        print("   plan:", plan(b"N", nh, M, M, fl | 8))
        run(f"X inv(R) REAL {nh}x{M}x{M} flags={fl}", b"N", nh, M, M, Y, n_G, Cm, k3, out, n_G, fl | 8,
            4.0 * nh * M * ((M + 1) / 2 if fl else M), reps=5)
    check(lib.dftk_mi_basis_destroy(h))
    sys.exit(0)
if only == "gramscan":
    for mm in (k3, 2 * M, M):
        for fl in (0, 1):
            run(f"Gram 3M  m=n={mm} flags={fl}", b"C", mm, mm, n_G, Y, n_G, AY, n_G, G, mm, fl,
                8.0 * (mm * (mm + 1) / 2 if fl else mm * mm) * n_G, reps=3)
            run(f"Gram REAL m=n={mm} flags={fl}", b"C", mm, mm, nh, Y, n_G, AY, n_G, G, mm, fl | 8,
                4.0 * (mm * (mm + 1) / 2 if fl else mm * mm) * nh, reps=3)
    run("BYX 3M   (2M x M)", b"C", 2 * M, M, n_G, Y, n_G, AY, n_G, G, 2 * M, 0, 8.0 * 2 * M * M * n_G, reps=3)
    run("BYX REAL (2M x M)", b"C", 2 * M, M, nh, Y, n_G, AY, n_G, G, 2 * M, 8, 4.0 * 2 * M * M * nh, reps=3)
    check(lib.dftk_mi_basis_destroy(h))
    sys.exit(0)
t_c = run("Gram 3M complex, full sphere, UPPER", b"C", k3, k3, n_G, Y, n_G, AY, n_G, G, k3, 1, 8.0 * k3 * (k3 + 1) / 2 * n_G)
t_r = run("Gram REAL, half sphere, UPPER", b"C", k3, k3, nh, Y, n_G, AY, n_G, G, k3, 1 | 8, 4.0 * k3 * (k3 + 1) / 2 * nh)
print(f"  -> speed-up of the Gram product: {t_c / t_r:.2f}x")
t_c = run("update X C 3M complex, full sphere", b"N", n_G, M, k3, Y, n_G, Cm, k3, out, n_G, 0, 8.0 * n_G * M * k3)
t_r = run("update X C REAL, half sphere", b"N", nh, M, k3, Y, n_G, Cm, k3, out, n_G, 8, 4.0 * nh * M * k3)
print(f"  -> speed-up of the block update: {t_c / t_r:.2f}x")
t_c = run("P' psi 3M complex (n_p = 1250)", b"C", 1250, M, n_G, Y, n_G, AY, n_G, G, 1250, 0, 8.0 * 1250 * M * n_G)
t_r = run("P' psi REAL, half sphere", b"C", 1250, M, nh, Y, n_G, AY, n_G, G, 1250, 8, 4.0 * 1250 * M * nh)
print(f"  -> speed-up of the projection: {t_c / t_r:.2f}x")
check(lib.dftk_mi_basis_destroy(h))
