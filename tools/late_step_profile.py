"""Where a LATE SCF step of the benchmark cell spends its time: family timers of the library over steps 4..8 only
(the first two steps diagonalise 59 + 15 LOBPCG iterations and dominate the whole-SCF profile)."""
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
st = dftk.ScfStepper(basis, tol=1e-6)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for _ in range(skip):
    st.step()
torch.cuda.synchronize()
check(lib.dftk_mi_prof_enable(basis.handle, 1))
t0 = time.time()
timers = {}
nst = int(sys.argv[3]) if len(sys.argv) > 3 else 6
for _ in range(nst):
    info = st.step()
    for k, v in info["timers"].items():
        timers[k] = timers.get(k, 0.0) + v
torch.cuda.synchronize()
wall = time.time() - t0
check(lib.dftk_mi_prof_enable(basis.handle, 0))
names = {0: "zgemm", 11: "zgemm_struct", 1: "fftA", 2: "fftB", 3: "fftC", 4: "fftD", 5: "fftE", 6: "dens", 7: "heev", 8: "chol", 9: "applyH(total)"}
tot = 0.0
for f, nm in names.items():
    ms, work, nl = C.c_double(), C.c_double(), C.c_int64()
    check(lib.dftk_mi_prof_get(basis.handle, f, C.byref(ms), C.byref(work), C.byref(nl)))
    print(f"{nm:14s} {ms.value / nst:8.2f} ms/step  {nl.value / nst:7.1f} launches/step")
    if f != 9:
        tot += ms.value
print(f"booked {tot / nst:.1f} ms/step of wall {1e3 * wall / nst:.1f} ms/step; host timers/step:",
      {k: round(1e3 * v / nst, 1) for k, v in timers.items()}, "iters", info["diagonalization"]["n_iter"])
# (DFTK_MI_GEMM_SHAPES=1: the per-shape zgemm table of these steps is printed to stderr by dftk_mi_prof_enable(0) above)
