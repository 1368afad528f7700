#!/usr/bin/env python
"""Effective HBM rate of the n_G-sized streaming kernels of one LOBPCG iteration at the headline cell's block shape
(half-sphere rows x 503 bands) through the C-ABI: column norms / dots, the fused residual pass, the TPA
preconditioner, the half <-> full sphere conversions.  Bytes = the arrays each call must read and write once.
python tools/ew_bench.py [supercell n = 5]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd._lib import check  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lib = dftk.load_library()
lat, atoms, pos = dftk.silicon_cell((n, n, n))
model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
basis = dftk.PlaneWaveBasis(model, 30.0, dftk.MonkhorstPack((1, 1, 1)))
kpt = basis.kpoints[0]
kb, bh = kpt.handle, basis.handle
nG = kpt.n_G
nh = C.c_int64()
check(lib.dftk_mi_gamma_half_size(kb, C.byref(nh)))
nh = nh.value
M = dftk.AdaptiveBands(model).n_bands_compute
gen = torch.Generator(device="cuda").manual_seed(0)


def rnd(r, c):
    return torch.complex(torch.randn((c, r), dtype=torch.float64, device="cuda", generator=gen),
                         torch.randn((c, r), dtype=torch.float64, device="cuda", generator=gen))


X, AX, R = rnd(nh, M), rnd(nh, M), rnd(nh, M)
F = rnd(nG, M)
out = np.zeros(2 * M)
outc = np.zeros(2 * M)
lam = np.linspace(-0.2, 0.4, M)
mk, xx = np.zeros(M), np.zeros(M)
blk, full = 16 * nh * M, 16 * nG * M


def timeit(name, nbytes, fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    print(f"{name:34s} {1e3 * dt:8.3f} ms  {nbytes / dt / 1e12:6.2f} TB/s  ({nbytes / 1e6:.0f} MB)")


print(f"block: {nh} half-sphere rows ({nG} full) x {M} bands")
timeit("torch copy half block (reference)", 2 * blk, lambda: R.copy_(X))
timeit("columnwise_norms", blk, lambda: check(lib.dftk_mi_columnwise_norms(bh, nh, M, X.data_ptr(), nh, out.ctypes.data)))
timeit("columnwise_dots", 2 * blk, lambda: check(lib.dftk_mi_columnwise_dots(bh, nh, M, X.data_ptr(), nh, AX.data_ptr(), nh,
                                                                              outc.ctypes.data)))
timeit("gamma_compress_aligned (2 passes)", full + blk, lambda: check(lib.dftk_mi_gamma_compress_aligned(
    kb, M, F.data_ptr(), nG, X.data_ptr(), nh)))
timeit("gamma_compress", full + blk, lambda: check(lib.dftk_mi_gamma_compress(kb, M, F.data_ptr(), nG, X.data_ptr(), nh)))
timeit("gamma_expand", full + blk, lambda: check(lib.dftk_mi_gamma_expand(kb, M, X.data_ptr(), nh, F.data_ptr(), nG)))
# full-sphere forms of the residual / preconditioner entry points (the block's kinetic vector has n_G rows)
F2, F3 = rnd(nG, M), rnd(nG, M)
timeit("block_residual (full sphere)", 3 * full, lambda: check(lib.dftk_mi_block_residual(
    kb, M, F.data_ptr(), nG, F2.data_ptr(), nG, lam.ctypes.data, F3.data_ptr(), nG, out.ctypes.data, mk.ctypes.data,
    xx.ctypes.data)))
mk[:] = 1.0
timeit("tpa_ldiv (full sphere)", 2 * full, lambda: check(lib.dftk_mi_tpa_ldiv(kb, M, F.data_ptr(), nG, mk.ctypes.data, 1.0,
                                                                               F3.data_ptr(), nG)))
