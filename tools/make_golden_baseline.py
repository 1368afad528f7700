#!/usr/bin/env python
"""Generate tests/golden/baseline_cfg*.json: converged oracle SCF results for the BASELINE.json configs at
their REAL Ecut / FFT box (reduced explicit k-lists where the full mesh is only a repetition of the same work).

The oracle (oracle/, NumPy restatement pinned to the reference's golden vectors by tests/test_oracle_golden.py)
cannot run on the GPU box within the test budget for these sizes, so its converged numbers are committed as
fixtures together with this script (task statement, section 3).  Run:  python tools/make_golden_baseline.py [cfg ...]

cfg2  -- SURVEY appendix B identity: the Si 4x4x4 supercell at Gamma with a 160^3 cube == the primitive cell
         (a = 10.26, LDA = lda_x + lda_c_pw, Ecut 30) on the unshifted 4x4x4 Monkhorst-Pack mesh with a 40^3 cube.
cfg5  -- the same identity for the headline cell: Si 5x5x5 supercell at Gamma with a 200^3 cube == the primitive cell on
         the unshifted 5x5x5 mesh with a 40^3 cube.
cfg1  -- Si primitive, LDA, Ecut 15, unreduced 4x4x4 mesh, cube from compute_fft_size (27^3; the reference's
         symmetry-adapted 30^3 is run as a second fixture "cfg1_fft30").
cfg3  -- Al fcc (a = 7.6324708938577865), HGH PBE, Ecut 40, 36^3, Gaussian smearing T = 1e-3, unreduced 3x3x3 mesh
         (27 of the 12^3 mesh's k-point class; same n_G ~ 1.36k, 6 bands per k-point).
cfg4  -- graphene (examples/graphene.jl geometry: a = 4.66, L = 20), HGH PBE, Ecut 40, 30x30x120, Fermi-Dirac
         T = 1e-3, unreduced 3x3x1 mesh (9 of the 9x9x1 mesh's k-point class).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def run(name, model, Ecut, kgrid, fft_size, tol=1e-10, note=""):
    t0 = time.time()
    basis = oracle.PlaneWaveBasis(model, Ecut, kgrid, fft_size=fft_size)
    hist = []

    def cb(info):
        hist.append((info["n_iter"], info["energies"].total, info["history_drho"][-1]))
        print(f"  [{name}] {info['n_iter']:3d} E={info['energies'].total:+.12f} drho={info['history_drho'][-1]:.2e} "
              f"t={time.time() - t0:.0f}s", flush=True)
    res = oracle.self_consistent_field(basis, tol=tol, callback=cb, maxiter=60)
    assert res["converged"], name
    n_conv = res["n_bands_converge"]
    out = dict(
        name=name, note=note, Ecut=Ecut, fft_size=list(basis.fft_size), tol=tol,
        functionals=list(model.functionals), temperature=model.temperature, smearing=model.smearing,
        kcoords=[list(map(float, k.coordinate)) for k in basis.kpoints], kweights=list(map(float, basis.kweights)),
        n_G=[int(len(k.mapping)) for k in basis.kpoints], n_bands_converge=int(n_conv),
        energies={k: float(v) for k, v in res["energies"].items()}, E_total=float(res["energies"].total),
        eF=float(res["eF"]), n_iter=int(res["n_iter"]),
        eigenvalues=[list(map(float, lam)) for lam in res["eigenvalues"]],
        occupation=[list(map(float, o)) for o in res["occupation"]],
        rho_checks=dict(sum_dvol=float(res["rho"].sum() * basis.dvol), max=float(res["rho"].max()),
                        min=float(res["rho"].min()), norm_sqrt_dvol=float(np.linalg.norm(res["rho"]) * np.sqrt(basis.dvol))),
        oracle_wall_s=round(time.time() - t0, 1),
    )
    with open(os.path.join(OUT, f"baseline_{name}.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(f"[{name}] E = {out['E_total']:.12f}  n_iter = {out['n_iter']}  wall = {out['oracle_wall_s']} s", flush=True)


def si_model():
    lat, atoms, pos = oracle.basis.silicon_primitive(a=10.26, functional="lda")
    return oracle.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))


def main(which):
    if "cfg2" in which:
        run("cfg2_prim_4x4x4_ecut30_fft40", si_model(), 30, oracle.MonkhorstPack((4, 4, 4)), (40, 40, 40),
            note="== Si 4x4x4 supercell at Gamma with fft 160^3 (E_total x 64; union spectrum)")
    if "cfg5" in which:
        run("cfg5_prim_5x5x5_ecut30_fft40", si_model(), 30, oracle.MonkhorstPack((5, 5, 5)), (40, 40, 40),
            note="== Si 5x5x5 supercell (250 atoms, 1000 electrons: the headline cell) at Gamma with fft 200^3 "
                 "(E_total x 125; union spectrum)")
    if "cfg1" in which:
        run("cfg1_si_ecut15_k4_fft27", si_model(), 15, oracle.MonkhorstPack((4, 4, 4)), None)
        run("cfg1_si_ecut15_k4_fft30", si_model(), 15, oracle.MonkhorstPack((4, 4, 4)), (30, 30, 30))
    if "cfg3" in which:
        a = 7.6324708938577865
        lat = a / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
        Al = oracle.ElementPsp("Al", oracle.load_psp_hgh("Al", "pbe"))
        model = oracle.model_DFT(lat, [Al], [np.zeros(3)], functionals=("gga_x_pbe", "gga_c_pbe"), temperature=1e-3,
                                 smearing="gaussian")
        run("cfg3_al_pbe_ecut40_k3", model, 40, oracle.MonkhorstPack((3, 3, 3)), None)
    if "cfg4" in which:
        a, L = 4.66, 20.0
        lat = np.array([[a / 2, a / 2, 0.0], [-a * np.sqrt(3) / 2, a * np.sqrt(3) / 2, 0.0], [0.0, 0.0, L]])
        C_ = oracle.ElementPsp("C", oracle.load_psp_hgh("C", "pbe"))
        pos = [np.array([1 / 3, -1 / 3, 0.0]), np.array([-1 / 3, 1 / 3, 0.0])]
        model = oracle.model_DFT(lat, [C_, C_], pos, functionals=("gga_x_pbe", "gga_c_pbe"), temperature=1e-3,
                                 smearing="fermi_dirac")
        run("cfg4_graphene_pbe_ecut40_k3", model, 40, oracle.MonkhorstPack((3, 3, 1)), None)


def main_full(which):
    """The k-point configs EXACTLY as the reference's constructor builds them: crystal symmetries on, irreducible
    Monkhorst-Pack points, symmetry-adapted FFT size, LDOS mixing (the SCF default)."""
    if "cfg3full" in which:
        a = 7.6324708938577865
        lat = a / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
        Al = oracle.ElementPsp("Al", oracle.load_psp_hgh("Al", "pbe"))
        model = oracle.model_DFT(lat, [Al], [np.zeros(3)], functionals=("gga_x_pbe", "gga_c_pbe"), temperature=1e-3,
                                 smearing="gaussian", symmetries=True)
        run("cfg3_al_pbe_ecut40_k12_sym", model, 40, oracle.MonkhorstPack((12, 12, 12)), None,
            note="BASELINE configs[2] in full: 12x12x12 mesh, 48 symmetries -> 72 irreducible k-points")
    if "cfg4full" in which:
        a, L = 4.66, 20.0
        lat = np.array([[a / 2, a / 2, 0.0], [-a * np.sqrt(3) / 2, a * np.sqrt(3) / 2, 0.0], [0.0, 0.0, L]])
        C_ = oracle.ElementPsp("C", oracle.load_psp_hgh("C", "pbe"))
        pos = [np.array([1 / 3, -1 / 3, 0.0]), np.array([-1 / 3, 1 / 3, 0.0])]
        model = oracle.model_DFT(lat, [C_, C_], pos, functionals=("gga_x_pbe", "gga_c_pbe"), temperature=1e-3,
                                 smearing="fermi_dirac", symmetries=True)
        run("cfg4_graphene_pbe_ecut40_k9_sym", model, 40, oracle.MonkhorstPack((9, 9, 1)), None,
            note="BASELINE configs[3] in full: 9x9x1 mesh with the crystal symmetries")


if __name__ == "__main__":
    main_full(sys.argv[1:])
    main(sys.argv[1:] or ["cfg2", "cfg5", "cfg1", "cfg3", "cfg4"])
