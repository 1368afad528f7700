#!/usr/bin/env python
"""Gamma-only silicon supercell twice: with the real-symmetric orbitals the library uses at k = 0 by default
(psi(-G) = conj psi(G): real matrix products over half the rows, two bands per FFT pass) and with the general complex
orbitals the reference iterates -- same energies, density and eigenvalues, about half the time.

    python examples/silicon_supercell_gamma_real.py [n]        (n x n x n supercell, default 3 = 54 atoms)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lattice, atoms, positions = dftk.silicon_cell((n, n, n))
model = dftk.model_DFT(lattice, atoms, positions)
results = {}
for name, mode in (("real-symmetric (default)", None), ("general complex", False)):
    basis = dftk.PlaneWaveBasis(model, 30, dftk.MonkhorstPack((1, 1, 1)), gamma_real=mode)
    dftk.self_consistent_field(basis, tol=1e-2, maxiter=2)                  # warm the library up (allocations)
    res = dftk.self_consistent_field(basis, tol=1e-8)
    results[name] = (basis, res)
    print(f"{name:26s} E = {res['energies'].total:.10f} Ha, {res['n_iter']} SCF steps, {res['runtime']:.2f} s "
          f"(gamma_real = {basis.kpoints[0].gamma_real})")
(b1, r1), (b0, r0) = results.values()
nconv = r0["n_bands_converge"]
print("|dE|        =", abs(r1["energies"].total - r0["energies"].total))
print("|d eps|max  =", float(np.max(np.abs(r1["eigenvalues"][0][:nconv] - r0["eigenvalues"][0][:nconv]))))
print("|d rho|_2   =", float(torch.linalg.norm(r1["rho"] - r0["rho"])) * np.sqrt(b0.dvol))
