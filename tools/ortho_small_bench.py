#!/usr/bin/env python
"""Micro-benchmark of the fused small-block orthogonalisation kernel (batch_kernels.hip: k_b_ortho_reg / k_b_ortho) on the
block shapes of the k-point workloads: wall time per stand-alone call (one batched round: launch + stream synchronisation)
and, with DFTK_MI_ORTHO_CLOCKS=1, the kernel's own shader clocks per phase.
python tools/ortho_small_bench.py [reps = 20]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd._lib import check  # noqa: E402
from test_gpu_kernels import Basis, dev  # noqa: E402

EPS = float(np.finfo(float).eps)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = dftk.load_library()
bs = Basis(lib, 8, 8, 8)
rng = np.random.default_rng(1)


def block(n, m):
    return rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m))


for (n, m, ny) in [(1350, 6, 12), (1350, 6, 14), (1350, 6, 0), (725, 7, 14), (2040, 8, 16), (4653, 8, 16)]:
    Y = np.linalg.qr(block(n, max(ny, 1)))[0][:, :ny]
    X = block(n, m)
    Yd = dev(Y.T.copy()) if ny else None
    res = np.zeros(4)
    walls = []
    for r in range(reps):
        Xd = dev(X.T.copy())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        check(lib.dftk_mi_ortho_small(bs.h, n, m, Xd.data_ptr(), n, ny, Yd.data_ptr() if ny else None, n, None, 2 * EPS,
                                      res.ctypes.data))
        walls.append(time.perf_counter() - t0)
    print(f"n={n} m={m} ny={ny}: call wall median {1e6 * np.median(walls):.1f} us, min {1e6 * min(walls):.1f} us; "
          f"status {res[0]}, rounds {res[1]}, Cholesky {res[2]}", flush=True)
