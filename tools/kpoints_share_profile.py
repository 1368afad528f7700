#!/usr/bin/env python
"""Where ONE RANK'S SHARE of a k-point workload spends an SCF step (the floor of k-point strong scaling): the first
ceil(n_k / N) irreducible k-points of BASELINE configs[2] (Al fcc PBE, Ecut 40, 12^3 mesh) as a self-consistent problem of
its own, host timers of the stepper per phase (median over late steps) and, with --torch-profile, the top host-side ops.
python tools/kpoints_share_profile.py [N = 8] [--torch-profile] [--no-phase-timers] [--steps 12]"""
import os
import sys
import time

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
st = dftk.ScfStepper(sub, tol=1e-12, phase_timers="--no-phase-timers" not in sys.argv)
walls, timers = [], []
n_steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 12
for i in range(n_steps):
    torch.cuda.synchronize()
    t0 = time.time()
    info = st.step()
    torch.cuda.synchronize()
    walls.append(time.time() - t0)
    timers.append(dict(info["timers"]))
late = slice(min(4, n_steps - 1), None)
print(f"{n_loc} of {n_k} k-points (N = {N}), fft {sub.fft_size}, kbatch={sub.kbatch}: median late step {1e3 * np.median(walls[late]):.2f} ms "
      f"(all: {[round(1e3 * w, 2) for w in walls]})")
for k in timers[-1]:
    print(f"  {k:22s} {1e3 * np.median([t.get(k, 0.0) for t in timers[late]]):7.2f} ms")
print("  LOBPCG iterations of the late steps:", [float(np.mean(info["diagonalization"]["n_iter"]))])
import ctypes as C  # noqa: E402
a_, b_ = C.c_int64(), C.c_int64()
sub.lib.dftk_mi_lobpcg_small_stats(C.byref(a_), C.byref(b_))
l0, s0 = C.c_int64(), C.c_int64()
sub.lib.dftk_mi_launch_count(C.byref(l0), C.byref(s0))
st.step()
torch.cuda.synchronize()
l1, s1 = C.c_int64(), C.c_int64()
sub.lib.dftk_mi_launch_count(C.byref(l1), C.byref(s1))
from dftk_jl_amd.eigen import batch_stats  # noqa: E402
print(f"  small-block driver: {a_.value} calls, {b_.value} restarts; one more step: {l1.value - l0.value} library launches, "
      f"{s1.value - s0.value} host synchronisations; last batched call {batch_stats(sub)}")
if "--torch-profile" in sys.argv:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            st.step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
