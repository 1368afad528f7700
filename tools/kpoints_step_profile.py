"""Where a late SCF step of the k-point workload (BASELINE configs[2]: Al, PBE, 12^3 mesh -> 72 k-points) spends its
host time: cProfile of three steps after three warm-up steps, plus the batched call's counters."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd.eigen import batch_stats  # noqa: E402

a = 7.6324708938577865
lat = a / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
Al = dftk.ElementPsp("Al", dftk.load_psp("Al", "pbe"))
model = dftk.model_DFT(lat, [Al], [np.zeros(3)], functionals=("gga_x_pbe", "gga_c_pbe"), temperature=1e-3,
                       smearing="gaussian", symmetries=True)
basis = dftk.PlaneWaveBasis(model, 40.0, dftk.MonkhorstPack((12, 12, 12)))
st = dftk.ScfStepper(basis, tol=1e-6, phase_timers=True)
for _ in range(3):
    st.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
for _ in range(3):
    info = st.step()
pr.disable()
torch.cuda.synchronize()
print(f"{(time.time() - t0) / 3 * 1e3:.1f} ms per step; kbatch={basis.kbatch}; timers {info['timers']}; last batched call {batch_stats(basis)}")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
