#!/usr/bin/env python
"""Time dftk_mi_heev on random Hermitian matrices (GPU only): wall time per call vs the number of sweeps."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd._lib import check  # noqa: E402

lib = dftk.load_library()
h = C.c_void_p()
check(lib.dftk_mi_basis_create(8, 8, 8, 1.0, 0, C.byref(h)))
real = "real" in sys.argv[1:]        # real symmetric input (zero imaginary parts): the real-rotation path
for n in [int(a) for a in sys.argv[1:] if a != "real"] or [259, 518, 777]:
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)) + (0 if real else 1j) * rng.standard_normal((n, n))
    A = (A + A.conj().T) / 2 + np.diag(np.linspace(-1, 30, n)) * 3
    Ad0 = torch.tensor(A.T.copy(), dtype=torch.complex128, device="cuda")
    V = torch.empty_like(Ad0)
    W = np.zeros(n)
    for rep in range(3):
        Ad = Ad0.clone()
        torch.cuda.synchronize()
        t0 = time.time()
        check(lib.dftk_mi_heev(h, n, Ad.data_ptr(), n, W.ctypes.data, V.data_ptr(), n))
        torch.cuda.synchronize()
        dt = time.time() - t0
    ref = np.linalg.eigvalsh(A)
    print(f"n={n}{' real' if real else ''}: {dt * 1e3:.2f} ms per call, max |dW| = {np.abs(W - ref).max():.2e}")
