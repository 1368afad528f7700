#!/usr/bin/env python
"""examples/collinear_magnetism.jl of the reference on the MI355X path: bcc iron with collinear spin.  The initial magnetic
moment (4 mu_B) makes the model spin-polarised and breaks the symmetry of the guess density; the SCF relaxes to the
ferromagnetic solution (~2.5 mu_B at this discretisation -- the reference's test/iron_lda.jl, whose ABINIT-pinned energy and
eigenvalues the test-suite reproduces).

    python examples/iron_collinear_magnetism.py            (needs an MI355X; there is no CPU fallback)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402

a = 5.42352                                                            # bcc lattice constant in Bohr
lattice = a / 2 * np.array([[-1, 1, 1], [1, -1, 1], [1, 1, -1.0]])
Fe = dftk.ElementPsp("Fe", dftk.load_psp("Fe", "lda"))                  # HGH Fe-q8
magnetic_moments = [4.0]
model = dftk.model_DFT(lattice, [Fe], [np.zeros(3)], functionals=("lda_xc_teter93",), temperature=0.01,
                       smearing="fermi_dirac", magnetic_moments=magnetic_moments, symmetries=True)
basis = dftk.PlaneWaveBasis(model, 15, dftk.MonkhorstPack((4, 4, 4), (0.5, 0.5, 0.5)), fft_size=(20, 20, 20))
print(f"spin_polarization = {model.spin_polarization}: {len(basis.kpoints)} k-blocks "
      f"({len(basis.kpoints) // 2} irreducible k-points x 2 spins), weights sum to {sum(basis.kweights):.0f}")
rho0 = dftk.guess_density(basis, magnetic_moments)                     # (2, nz, ny, nx): spin up, spin down
scfres = dftk.self_consistent_field(basis, rho=rho0, tol=1e-8, callback=dftk.ScfDefaultCallback())
rho = scfres["rho"]
print(f"total energy {scfres['energies'].total:+.10f} Ha,  magnetisation "
      f"{float((rho[0] - rho[1]).sum()) * basis.dvol:.4f} mu_B,  Fermi level {scfres['eF']:.6f} Ha")
dftk.save_scfres("iron_scfres.json", scfres)                           # eigenvalues[spin][kpoint][band], rho[spin][z][y][x]
