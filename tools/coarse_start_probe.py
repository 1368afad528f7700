#!/usr/bin/env python
"""Probe: how many LOBPCG iterations does the FIRST diagonalisation of the Gamma cell need when it starts from the solution
of the same Hamiltonian on a coarser plane-wave basis (Ecut / 4, half the cube) instead of random orbitals?
python tools/coarse_start_probe.py [supercell = 5] [ecut_ratio = 0.25]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ratio = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
lat, atoms, pos = dftk.silicon_cell((n, n, n))
model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))


def sync():
    torch.cuda.synchronize()


t0 = time.time()
bf = dftk.PlaneWaveBasis(model, 30.0, dftk.MonkhorstPack((1, 1, 1)))
sync()
t_setup_f = time.time() - t0
t0 = time.time()
bc = dftk.PlaneWaveBasis(model, 30.0 * ratio, dftk.MonkhorstPack((1, 1, 1)))
sync()
t_setup_c = time.time() - t0
print(f"fine: fft {bf.fft_size} n_G {bf.kpoints[0].n_G} setup {t_setup_f:.2f} s; coarse: fft {bc.fft_size} n_G {bc.kpoints[0].n_G} "
      f"setup {t_setup_c:.2f} s", flush=True)
nb = dftk.AdaptiveBands(model).n_bands_compute
nconv = dftk.AdaptiveBands(model).n_bands_converge
tol = 0.025


def ham_of(b):
    rho = dftk.guess_density(b)
    _, ham = dftk.energy_hamiltonian(b, None, None, rho=rho)
    return ham[0]


Hf, Hc = ham_of(bf), ham_of(bc)
gen = torch.Generator(device="cuda").manual_seed(1)


def run(H, X0, label):
    sync()
    t = time.time()
    r = dftk.lobpcg_hyper(H, X0, prec=dftk.PreconditionerTPA(H), tol=tol, n_conv_check=nconv)
    sync()
    dt = time.time() - t
    print(f"{label}: {r.n_iter} iterations, {r.n_matvec} H psi, {dt:.3f} s, max residual of the checked bands "
          f"{np.max(r.residual_norms[:nconv]):.2e}, lambda[0, nconv-1] = {r.λ[0]:.6f} {r.λ[nconv - 1]:.6f}", flush=True)
    return r, dt


for rep in range(2):     # (the first pass also pays allocations / first launches)
    Xf0 = dftk.random_orbitals(bf, bf.kpoints[0], nb, gen)
    rf, t_rand = run(Hf, Xf0, "fine, random start")
    Xc0 = dftk.random_orbitals(bc, bc.kpoints[0], nb, gen)
    rc, t_coarse = run(Hc, Xc0, "coarse, random start")
    # transfer: coefficients of the coarse sphere into the fine sphere (zero elsewhere), by integer G
    sync()
    t = time.time()
    kf, kc = bf.kpoints[0], bc.kpoints[0]
    nx, ny, nz = bf.fft_size
    inv = torch.full((nx * ny * nz,), -1, dtype=torch.int64, device="cuda")
    inv[kf.mapping_device] = torch.arange(kf.n_G, device="cuda")
    G = kc.G_vectors
    lin = (G[:, 0] % nx) + nx * ((G[:, 1] % ny) + ny * (G[:, 2] % nz))
    posf = inv[lin]
    assert bool((posf >= 0).all())
    Xg = torch.zeros((nb, kf.n_G), dtype=torch.complex128, device="cuda")
    Xg[:, posf] = rc.X
    sync()
    t_tr = time.time() - t
    rg, t_fine = run(Hf, Xg, "fine, start from the coarse solution")
    print(f"  -> random {t_rand:.3f} s vs coarse {t_coarse:.3f} + transfer {t_tr:.3f} + fine {t_fine:.3f} = "
          f"{t_coarse + t_tr + t_fine:.3f} s (+ coarse set-up {t_setup_c:.2f} s if it is counted); eigenvalue difference "
          f"{np.abs(rg.λ[:nconv] - rf.λ[:nconv]).max():.2e}", flush=True)
