#!/usr/bin/env python
"""REAL N-type products  (n_half x K) * (K x 64)  for K = 64 .. 512: the per-column-tile pieces of the triangular
X inv(R) product of the headline cell (tile column j stops at k = 64 (j + 1)), each timed alone, next to the whole
structured call.  python tools/gemm_shortk_bench.py"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd._lib import check, cplx  # noqa: E402

lib = dftk.load_library()
h = C.c_void_p()
check(lib.dftk_mi_basis_create(8, 8, 8, 1.0, 0, C.byref(h)))
nh, M = 132430, 503
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn((M, nh), dtype=torch.complex128, device="cuda", generator=g)
R = torch.randn((M, M), dtype=torch.complex128, device="cuda", generator=g)
out = torch.empty((M, nh), dtype=torch.complex128, device="cuda")


def run(n, k, flags, col0=0, reps=8):
    for it in range(reps + 1):
        if it == 1:
            check(lib.dftk_mi_basis_sync(h))
            t0 = time.time()
        check(lib.dftk_mi_zgemm_ex(h, b"N", nh, n, k, cplx(1.0), X.data_ptr(), nh, R.data_ptr() + 16 * M * col0, M, cplx(0.0),
                                   out.data_ptr() + 16 * nh * col0, nh, flags))
    check(lib.dftk_mi_basis_sync(h))
    return (time.time() - t0) / reps


tot = 0.0
for j in range(8):
    k = min(M, 64 * (j + 1))
    n = 64 if j < 7 else M - 64 * 7
    dt = run(n, k, 8, col0=64 * j)
    tot += dt
    print(f"tile column {j}: n={n:3d} K={k:3d}  {1e3 * dt:7.3f} ms  {4.0 * nh * n * k / dt / 1e12:6.2f} TF/s")
print(f"sum of the pieces {1e3 * tot:.3f} ms")
for n in (128, 256):
    for k in (128, 256, 503):
        dt = run(n, k, 8)
        print(f"dense n={n:3d} K={k:3d}  {1e3 * dt:7.3f} ms  {4.0 * nh * n * k / dt / 1e12:6.2f} TF/s")
print(f"whole triangular call (flags = 2): {1e3 * run(M, M, 8 | 2):.3f} ms;  dense: {1e3 * run(M, M, 8):.3f} ms")
