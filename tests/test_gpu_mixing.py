"""Device mixing rules (dftk.jl_amd/mixing.py, mirror of src/scf/mixing.jl + chi0models.jl + postprocess/dos.jl)
against the oracle's on identical inputs, and the metal SCF with the reference's default ``LdosMixing``."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import dftk_jl_amd as dftk  # noqa: E402
import oracle  # noqa: E402

A_AL = 7.6324708938577865          # test/testcases.jl:74


def _al_models(supercell=(2, 1, 1), temperature=0.01, smearing="fermi_dirac"):
    lat = A_AL / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
    dAl = dftk.ElementPsp("Al", dftk.load_psp("Al", "lda"))
    oAl = oracle.ElementPsp("Al", oracle.load_psp_hgh("Al", "lda"))
    lat2, datoms, pos = dftk.create_supercell(lat, [dAl], [np.zeros(3)], supercell)
    _, oatoms, _ = oracle.basis.create_supercell(lat, [oAl], [np.zeros(3)], supercell)
    fun = ("lda_x", "lda_c_vwn")
    return (dftk.model_DFT(lat2, datoms, pos, functionals=fun, temperature=temperature, smearing=smearing),
            oracle.model_DFT(lat2, oatoms, pos, functionals=fun, temperature=temperature, smearing=smearing))


@pytest.fixture(scope="module")
def state():
    """One oracle diagonalisation of H[rho_guess] -> (psi, eigenvalues, eF) shared by both sides."""
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    dm, om = _al_models()
    kg = (1, 2, 2)
    ob = oracle.PlaneWaveBasis(om, 6, oracle.MonkhorstPack(kg))
    db = dftk.PlaneWaveBasis(dm, 6, dftk.MonkhorstPack(kg))
    assert db.fft_size == ob.fft_size
    _, oham = oracle.energy_hamiltonian(ob, None, None, rho=oracle.guess_density(ob))
    eig = oracle.diagonalize_all_kblocks(oham, 9, tol=1e-8, n_conv_check=6, interpolate_kpoints=False)
    occ, eF = oracle.compute_occupation(ob, eig["λ"])
    rho_out = oracle.compute_density(ob, eig["X"], occ)
    dF = rho_out - oracle.guess_density(ob)
    dpsi = [torch.from_numpy(np.ascontiguousarray(X.T)).cuda() for X in eig["X"]]
    return dict(ob=ob, db=db, psi=eig["X"], dpsi=dpsi, lam=eig["λ"], eF=eF, dF=dF,
                dFd=torch.from_numpy(dF).cuda())


def test_dos_and_ldos_match_oracle(state):
    s = state
    from oracle import mixing as om
    sm, T = om.default_smearing_temperature(s["ob"].model)
    assert (sm, T) == dftk.mixing.default_smearing_temperature(s["db"].model)
    for kind, temp in ((sm, T), ("fermi_dirac", 0.01)):
        d0 = om.compute_dos(s["eF"], s["ob"], s["lam"], kind, temp)
        d1 = dftk.compute_dos(s["eF"], s["db"], s["lam"], kind, temp)
        assert abs(d0 - d1) < 1e-12 * abs(d0)
        l0 = om.compute_ldos(s["eF"], s["ob"], s["lam"], s["psi"], kind, temp)
        l1 = dftk.compute_ldos(s["eF"], s["db"], s["lam"], s["dpsi"], kind, temp).cpu().numpy()
        assert np.linalg.norm(l1 - l0) < 1e-12 * np.linalg.norm(l0)
        assert abs(l0.sum() * s["ob"].dvol - d0) < 1e-10 * abs(d0)           # LDOS integrates to the DOS


@pytest.mark.parametrize("name", ["simple", "kerker", "kerkerdos", "dielectric", "ldos", "hybrid"])
def test_mix_density_matches_oracle(state, name):
    """mix_density(mixing, basis, dF; eF, eigenvalues, psi) on the device == oracle; the GMRES-based ones with a
    tight tolerance on both sides (their default reltol = 0.01 only bounds the residual)."""
    s = state
    from oracle import mixing as om
    dmix = {"simple": dftk.SimpleMixing(), "kerker": dftk.KerkerMixing(0.7), "kerkerdos": dftk.KerkerDosMixing(),
            "dielectric": dftk.DielectricMixing(0.9, 7.0), "ldos": dftk.LdosMixing(reltol=1e-11),
            "hybrid": dftk.HybridMixing(reltol=1e-11)}[name]
    omix = {"simple": om.SimpleMixing(), "kerker": om.KerkerMixing(0.7), "kerkerdos": om.KerkerDosMixing(),
            "dielectric": om.DielectricMixing(0.9, 7.0), "ldos": om.LdosMixing(reltol=1e-11),
            "hybrid": om.HybridMixing(reltol=1e-11)}[name]
    got = dmix.mix_density(s["db"], s["dFd"].clone(), eF=s["eF"], eigenvalues=s["lam"], psi=s["dpsi"]).cpu().numpy()
    ref = omix.mix_density(s["ob"], s["dF"].copy(), eF=s["eF"], eigenvalues=s["lam"], psi=s["psi"])
    assert np.linalg.norm(got - ref) < 1e-9 * np.linalg.norm(ref)
    if name in ("ldos", "hybrid"):
        assert dmix.last_gmres_applies > 3 and np.linalg.norm(ref - s["dF"]) > 1e-3 * np.linalg.norm(s["dF"])
        # the default tolerance: residual of the dielectric system below 1 % (mixing.jl:232)
        d2 = dftk.LdosMixing() if name == "ldos" else dftk.HybridMixing()
        g2 = d2.mix_density(s["db"], s["dFd"].clone(), eF=s["eF"], eigenvalues=s["lam"], psi=s["dpsi"]).cpu().numpy()
        assert np.linalg.norm(g2 - ref) < 0.05 * np.linalg.norm(ref) and d2.last_gmres_applies <= dmix.last_gmres_applies


def test_metal_scf_default_ldos_mixing_matches_oracle_and_simple():
    """self_consistent_field with the reference's default mixing (LdosMixing, T > 0) on the device: same fixed point
    as the oracle with LDOS mixing and as the device with simple mixing; iteration counts of the same class."""
    dm, om = _al_models(supercell=(3, 1, 1))
    db = dftk.PlaneWaveBasis(dm, 6, dftk.MonkhorstPack((1, 2, 2)))
    ob = oracle.PlaneWaveBasis(om, 6, oracle.MonkhorstPack((1, 2, 2)))
    r_ldos = dftk.self_consistent_field(db, tol=1e-9)
    r_simple = dftk.self_consistent_field(db, tol=1e-9, mixing=dftk.SimpleMixing())
    r_kerker = dftk.self_consistent_field(db, tol=1e-9, mixing=dftk.KerkerMixing())
    o_ldos = oracle.self_consistent_field(ob, tol=1e-9)
    for r in (r_ldos, r_simple, r_kerker, o_ldos):
        assert r["converged"]
    n_atoms = 3
    assert abs(r_ldos["energies"].total - o_ldos["energies"].total) < 1e-8 * n_atoms
    assert abs(r_ldos["energies"].total - r_simple["energies"].total) < 1e-8 * n_atoms
    assert abs(r_kerker["energies"].total - r_simple["energies"].total) < 1e-8 * n_atoms
    assert abs(r_ldos["eF"] - o_ldos["eF"]) < 1e-7
    assert abs(r_ldos["n_iter"] - o_ldos["n_iter"]) <= 3
    assert r_ldos["n_iter"] <= r_simple["n_iter"] + 2


@pytest.mark.parametrize("fft_size", [(16, 18, 20), (15, 16, 25)])
def test_fourier_multiplier_kernels_match_torch_twin(fft_size, monkeypatch):
    """The library's on-the-fly Fourier multipliers (``dftk_mi_mix_kerker``, ``dftk_mi_mix_dielectric``,
    ``dftk_mi_chi0_dielectric_apply``, ``dftk_mi_cube_fourier_filter``; src/scf/mixing.jl:61-72,161-171,
    chi0models.jl:66-77, hartree.jl:68-81) against the torch formulation with stored |G|^2 cubes, on even / odd /
    anisotropic cubes of a non-orthogonal cell (the multiplier needs the cartesian |B G|^2)."""
    dm, _ = _al_models(supercell=(2, 1, 1))
    db = dftk.PlaneWaveBasis(dm, 6, dftk.MonkhorstPack((1, 1, 1)), fft_size=fft_size)
    gen = torch.Generator(device="cuda").manual_seed(11)
    dF = torch.randn(fft_size[::-1], dtype=torch.float64, device="cuda", generator=gen) + 0.3
    mixes = [dftk.KerkerMixing(0.8), dftk.KerkerMixing(0.05), dftk.DielectricMixing(0.8, 10.0),
             dftk.DielectricMixing(1.3, 2.5)]
    chi0 = dftk.mixing.DielectricModel(7.0, 0.9)
    from dftk_jl_amd.mixing import _filter_array
    got = [m.mix_density(db, dF.clone()) for m in mixes]
    got.append(chi0(db)(torch.zeros_like(dF), dF, -1.0))
    got.append(_filter_array(db, db.terms.poisson, dF))
    monkeypatch.setenv("DFTK_MI_TORCH_LOCAL", "1")
    ref = [m.mix_density(db, dF.clone()) for m in mixes]
    ref.append(chi0(db)(torch.zeros_like(dF), dF, -1.0))
    ref.append(db.irfft(db.terms.poisson * db.fft(dF)))
    monkeypatch.delenv("DFTK_MI_TORCH_LOCAL")
    for g, r in zip(got, ref):
        assert float((g - r).norm()) < 1e-13 * float(r.norm())
    # the DC component of dF survives Kerker / dielectric mixing (mixing.jl:70-71)
    assert abs(float(got[0].mean()) - float(dF.mean())) < 1e-14 and abs(float(got[2].mean()) - float(dF.mean())) < 1e-14
