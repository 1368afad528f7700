#!/usr/bin/env python
"""examples/silicon.jl of the reference on the MI355X path: LDA silicon, Ecut 15, 4x4x4 Monkhorst-Pack mesh with the
crystal symmetries (8 irreducible k-points, 30^3 cube), SCF to 1e-6, results written as DFTK-style JSON.

    python examples/silicon.py            (needs an MI355X; there is no CPU fallback)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402

lattice, atoms, positions = dftk.silicon_cell()                       # a = 10.26 Bohr fcc, HGH Si-q4
model = dftk.model_DFT(lattice, atoms, positions, functionals=("lda_x", "lda_c_pw"), symmetries=True)
basis = dftk.PlaneWaveBasis(model, 15, dftk.MonkhorstPack((4, 4, 4)))
print(f"fft_size {basis.fft_size}, {len(basis.symmetries)} symmetries, {len(basis.kpoints)} irreducible k-points")
scfres = dftk.self_consistent_field(basis, tol=1e-6, callback=dftk.ScfDefaultCallback())
for name, value in scfres["energies"].items():
    print(f"    {name:15s} {value:+.10f}")
print(f"    {'total':15s} {scfres['energies'].total:+.10f}   ({scfres['n_iter']} SCF steps, "
      f"{scfres['n_matvec']} H psi applies)")
dftk.save_scfres("silicon_scfres.json", scfres)
