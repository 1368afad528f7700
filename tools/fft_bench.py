#!/usr/bin/env python
"""Time the five FFT stages of the local H psi (and the density z-pass) on a BASELINE-sized sphere:
python tools/fft_bench.py [supercell n = 5] [bands = 64] [bands per launch group = 32].  Prints ms per launch and algorithmic TB/s per stage
(HIP events inside the library).  DFTK_MI_ZPASS_CLASSIC=1 selects the LDS-staged z-pass for comparison."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd._lib import check  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
fft_batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
lat, atoms, pos = dftk.silicon_cell((n, n, n))
basis = dftk.PlaneWaveBasis(dftk.model_DFT(lat, atoms, pos), 30, dftk.MonkhorstPack((1, 1, 1)), build_terms=False)
kpt = basis.kpoints[0]
lib = basis.lib
check(lib.dftk_mi_basis_set_fft_batch(basis.handle, fft_batch))
g = torch.Generator(device="cuda").manual_seed(1)
V = torch.randn(basis.fft_size[::-1], dtype=torch.float64, device="cuda", generator=g)
H = dftk.DftHamiltonianBlock(basis, kpt, V)
psi = dftk.random_orbitals(basis, kpt, nb, g)
out = torch.empty_like(psi)
rho = torch.zeros(basis.fft_size[::-1], dtype=torch.float64, device="cuda")
w = np.full(nb, 0.5)
for rep in range(2):
    if rep == 1:
        check(lib.dftk_mi_prof_enable(basis.handle, 1))
    for _ in range(5):
        H.mul_(out, psi, 3)
        check(lib.dftk_mi_density_accumulate(kpt.handle, nb, psi.data_ptr(), psi.stride(0), w.ctypes.data, rho.data_ptr()))
    basis.sync()
names = {1: "A xbwd_scatter", 2: "B ybwd", 3: "C z fused V", 4: "D yfwd", 5: "E xfwd_gather", 6: "density z"}
print(f"fft {basis.fft_size}, n_G {kpt.n_G}, {nb} bands, {fft_batch} bands per launch group, classic={os.environ.get('DFTK_MI_ZPASS_CLASSIC') is not None}")
for fam, name in names.items():
    ms, work, cnt = C.c_double(), C.c_double(), C.c_int64()
    check(lib.dftk_mi_prof_get(basis.handle, fam, C.byref(ms), C.byref(work), C.byref(cnt)))
    print(f"  {name:16s} {ms.value / max(cnt.value, 1) * 1e3:8.1f} us/launch  {work.value / (ms.value * 1e-3) / 1e12:6.2f} TB/s  ({cnt.value} launches)")
