#!/usr/bin/env python
"""Host-side profile (cProfile, cumulative) of late SCF steps of one rank's share of the Al k-point workload: where the
Python mirror spends the step outside the library calls.  python tools/kstep_cprofile.py [N = 8]"""
import cProfile
import io
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
a = 7.6324708938577865
lat = a / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
Al = dftk.ElementPsp("Al", dftk.load_psp("Al", "pbe"))
model = dftk.model_DFT(lat, [Al], [np.zeros(3)], functionals=("gga_x_pbe", "gga_c_pbe"), temperature=1e-3,
                       smearing="gaussian", symmetries=True)
full = dftk.PlaneWaveBasis(model, 40.0, dftk.MonkhorstPack((12, 12, 12)), build_terms=False)
n_k = len(full.kcoords_global)
n_loc = -(-n_k // N)
kc = [np.asarray(k) for k in full.kcoords_global[:n_loc]]
kw = np.asarray(full.kweights_global[:n_loc], dtype=float)
os.environ["DFTK_MI_KBATCH"] = "1"
sub = dftk.PlaneWaveBasis(model, 40.0, dftk.ExplicitKpoints(kc, list(kw / kw.sum())), fft_size=full.fft_size)
st = dftk.ScfStepper(sub, tol=1e-12, phase_timers=False)
for i in range(5):
    st.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(6):
    st.step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70)
print(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print(s.getvalue())
