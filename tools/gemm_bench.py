#!/usr/bin/env python
"""Micro-benchmark of the library's zgemm on the LOBPCG / projector shapes (GPU only)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd._lib import check, cplx  # noqa: E402

lib = dftk.load_library()
h = C.c_void_p()
check(lib.dftk_mi_basis_create(8, 8, 8, 1.0, 0, C.byref(h)))
for w in (1, 2, 4):
    t = C.c_double()
    check(lib.dftk_mi_diag_mfma_peak(h, w, 4000, C.byref(t)))
    print(f"mfma_f64 peak, {w} wave(s)/SIMD: {t.value:.1f} TFLOP/s")

nG = int(os.environ.get("NG", 135491))
shapes = [] if os.environ.get("PEAK_ONLY") else [("C", 259, 259, nG), ("C", 640, 259, nG), ("C", 518, 259, nG), ("C", 129, 129, nG), ("C", 259, 1, nG),
          ("N", nG, 259, 259), ("N", nG, 259, 640), ("N", nG, 259, 518), ("N", nG, 129, 129), ("N", nG, 1, 259)]
if os.environ.get("SHAPES"):   # e.g. SHAPES="N:135491:259:259,C:259:259:135491"
    shapes = [(t.split(":")[0], *map(int, t.split(":")[1:])) for t in os.environ["SHAPES"].split(",")]
gen = torch.Generator(device="cuda").manual_seed(0)


def rnd(r, c):
    return torch.complex(torch.randn((c, r), dtype=torch.float64, device="cuda", generator=gen),
                         torch.randn((c, r), dtype=torch.float64, device="cuda", generator=gen))


for tr, m, n, k in shapes:
    A = rnd(m, k) if tr == "N" else rnd(k, m)
    B = rnd(k, n)
    Cm = torch.zeros((n, m), dtype=torch.complex128, device="cuda")
    torch.cuda.synchronize()
    lda = m if tr == "N" else k
    for rep in range(2):
        check(lib.dftk_mi_prof_enable(h, 1))
        nrep = 5
        for _ in range(nrep):
            check(lib.dftk_mi_zgemm(h, tr.encode(), m, n, k, cplx(1), A.data_ptr(), lda, B.data_ptr(), k, cplx(0),
                                    Cm.data_ptr(), m))
        ms, work, nl = C.c_double(), C.c_double(), C.c_int64()
        check(lib.dftk_mi_prof_get(h, 0, C.byref(ms), C.byref(work), C.byref(nl)))
    print(f"{tr} m={m:7d} n={n:5d} k={k:7d}: {ms.value / nrep:9.3f} ms  {work.value / (ms.value * 1e-3) / 1e12:7.2f} TFLOP/s")
