#!/usr/bin/env python
"""Time dftk_mi_potrf_trtri (Cholesky + inverse + both norm estimates, one host fetch) on Gram matrices of LOBPCG's sizes:
python tools/potrf_bench.py [n ...] (default 259 503 512 600 1006)."""
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
for n in [int(a) for a in sys.argv[1:]] or [259, 503, 512, 600, 1006]:
    rng = np.random.default_rng(n)
    X = rng.standard_normal((3 * n, n))
    O = (X.T @ X).astype(complex)
    Od0 = torch.tensor(O.T.copy(), dtype=torch.complex128, device="cuda")
    Z = torch.empty_like(Od0)
    for name, fn in (("complex", lib.dftk_mi_potrf_trtri), ("real", lib.dftk_mi_potrf_trtri_real)):
        ts = []
        for rep in range(6):
            Od = Od0.clone()
            torch.cuda.synchronize()
            t0 = time.time()
            check(fn(h, n, Od.data_ptr(), n, Z.data_ptr(), n))
            torch.cuda.synchronize()
            ts.append(time.time() - t0)
        R = np.triu(Od.cpu().numpy().T)
        err = np.linalg.norm(R.conj().T @ R - O) / np.linalg.norm(O)
        erri = np.linalg.norm(Z.cpu().numpy().T @ R - np.eye(n))
        print(f"n={n} {name:8s}: {1e3 * min(ts[1:]):.3f} ms per call (best of 5), |R'R - O|/|O| = {err:.1e}, "
              f"|inv(R) R - I| = {erri:.1e}")
