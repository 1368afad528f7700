#!/usr/bin/env python
"""Time the dense eigensolvers (GPU only).

  heev_bench.py [real] n ...            dftk_mi_heev on random Hermitian matrices (all n pairs, blocked Jacobi)
  heev_bench.py lowest [complex] n ...  dftk_mi_heev_lowest (nev = n / 3 for n divisible by 3, else n / 2) against
                                        dftk_mi_heev on matrices with the structure of LOBPCG Rayleigh-Ritz matrices
                                        (tests/test_gpu_eig.py::rr_like), for a loose and a tight coupling of the
                                        leading block; DFTK_MI_HEEV_TRACE=1 shows the iteration counts
"""
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
args = sys.argv[1:]


def timed(fn, reps=3):
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.time()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.time() - t0)
    return best


if args and args[0] == "lowest":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_gpu_eig import rr_like  # noqa: E402
    cplx = "complex" in args
    for n in [int(a) for a in args[1:] if a != "complex"] or [1006, 1509]:
        nev = n // 3 if n % 3 == 0 else n // 2
        for coupling in (0.5, 1e-3):
            rng = np.random.default_rng(n)
            A = rr_like(n, nev, rng, cplx, coupling)
            Ad0 = torch.tensor(np.ascontiguousarray(A.T), dtype=torch.complex128, device="cuda")
            V = torch.empty_like(Ad0)
            W = np.zeros(n)
            Wf = np.zeros(n)
            hold = {}

            def lowest():
                hold["A"] = Ad0.clone()
                torch.cuda.synchronize()
                t0 = time.time()
                check(lib.dftk_mi_heev_lowest(h, n, nev, hold["A"].data_ptr(), n, W.ctypes.data, V.data_ptr(), n))
                torch.cuda.synchronize()
                hold["t"] = min(hold.get("t", 1e30), time.time() - t0)

            def full():
                hold["A"] = Ad0.clone()
                torch.cuda.synchronize()
                t0 = time.time()
                check(lib.dftk_mi_heev(h, n, hold["A"].data_ptr(), n, Wf.ctypes.data, V.data_ptr(), n))
                torch.cuda.synchronize()
                hold["tf"] = min(hold.get("tf", 1e30), time.time() - t0)

            for _ in range(3):
                full()
            for _ in range(3):
                lowest()
            Vh = V.cpu().numpy().T[:, :nev]
            ref = np.linalg.eigvalsh(A)
            print(f"n={n} nev={nev} {'complex' if cplx else 'real'} coupling={coupling:g}: lowest {hold['t'] * 1e3:.2f} ms, "
                  f"full Jacobi {hold['tf'] * 1e3:.2f} ms; max |dW| = {np.abs(W[:nev] - ref[:nev]).max():.2e}, "
                  f"|V'V - I| = {np.abs(Vh.conj().T @ Vh - np.eye(nev)).max():.2e}, "
                  f"|AV - VW| = {np.abs(A @ Vh - Vh * W[:nev]).max():.2e}", flush=True)
    sys.exit(0)

real = "real" in args        # real symmetric input (zero imaginary parts): the real-rotation path
for n in [int(a) for a in args if a != "real"] or [259, 518, 777]:
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
