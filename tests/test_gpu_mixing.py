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


def test_native_anderson_matches_torch_twin_and_oracle():
    """``dftk_mi_anderson_step`` (csrc/mix_kernels.hip; src/scf/anderson.jl:36-130) against the torch formulation of the host
    mirror and the oracle's NumPy restatement on a fixed-point problem whose history fills up, rolls over (m = 5) and
    drops badly conditioned entries: identical iterates."""
    from oracle.scf import AndersonAcceleration as OA
    from dftk_jl_amd.scf import AndersonAcceleration, AndersonNative
    assert torch.cuda.is_available()
    lat, atoms, pos = dftk.silicon_cell()
    basis = dftk.PlaneWaveBasis(dftk.model_DFT(lat, atoms, pos), 5, dftk.MonkhorstPack((1, 1, 1)), fft_size=(12, 12, 12))
    rng = np.random.default_rng(0)
    n = 12 ** 3
    Q = np.linalg.qr(rng.standard_normal((n, 40)))[0]
    A = np.eye(n) + Q @ np.diag(np.geomspace(0.05, 30.0, 40) - 1.0) @ Q.T          # SPD, condition 600
    bvec = rng.standard_normal(n)
    Ad, bd = torch.from_numpy(A).cuda(), torch.from_numpy(bvec).cuda()
    for m in (5, 0, 10):
        nat, twin, orc = AndersonNative(basis, m=m), AndersonAcceleration(m=m), OA(m=m)
        x_n = torch.zeros(n, dtype=torch.float64, device="cuda")
        x_t = x_n.clone()
        x_o = np.zeros(n)
        for it in range(30):
            x_n = nat(x_n.reshape(12, 12, 12), 0.05, (bd - Ad @ x_n.reshape(-1)).reshape(12, 12, 12)).reshape(-1)
            x_t = twin(x_t, 0.05, bd - Ad @ x_t)
            x_o = orc(x_o, 0.05, bvec - A @ x_o)
            scale = np.linalg.norm(x_o) + 1e-300
            assert np.linalg.norm(x_n.cpu().numpy() - x_o) < 1e-7 * scale, (m, it)
            assert float((x_n - x_t).norm()) < 1e-7 * scale, (m, it)
        if m:
            assert nat.n_history == min(m, 30)
            assert np.linalg.norm(A @ x_o - bvec) < 0.05 * np.linalg.norm(bvec)      # (accelerated: plain damping is at 0.2)
        nat.close()


@pytest.mark.parametrize("name,reltol", [("ldos", 1e-11), ("ldos", 0.01), ("hybrid", 1e-10), ("dielectric_only", 1e-10)])
def test_native_chi0_mixing_matches_torch_twin(state, name, reltol, monkeypatch):
    """``dftk_mi_chi0_mix`` (LDOS / dielectric models + restarted GMRES as ONE library call) against the host-logic GMRES
    on torch vectors (``DFTK_MI_TORCH_MIX=1``): same solution at a tight tolerance, and at the reference's default
    reltol = 0.01 the same number of operator applications (both walk the same Krylov steps)."""
    s = state
    mk = {"ldos": lambda: dftk.LdosMixing(reltol=reltol), "hybrid": lambda: dftk.HybridMixing(reltol=reltol),
          "dielectric_only": lambda: dftk.mixing.Chi0Mixing([dftk.mixing.DielectricModel(7.0, 0.9)], reltol=reltol)}[name]
    args = dict(eF=s["eF"], eigenvalues=s["lam"], psi=s["dpsi"])
    nat = mk()
    got = nat.mix_density(s["db"], s["dFd"].clone(), **args)
    monkeypatch.setenv("DFTK_MI_TORCH_MIX", "1")
    twin = mk()
    ref = twin.mix_density(s["db"], s["dFd"].clone(), **args)
    monkeypatch.delenv("DFTK_MI_TORCH_MIX")
    tol = 1e-9 if reltol < 1e-6 else 2 * reltol
    assert float((got - ref).norm()) < tol * float(ref.norm())
    assert nat.last_gmres_applies == twin.last_gmres_applies and nat.last_gmres_applies >= 2
    assert abs(float(got.mean()) - float(s["dFd"].mean())) < 1e-13                 # the DC component passes through


def test_native_chi0_mixing_collinear_matches_torch_twin(monkeypatch):
    """Two spin channels: the Hartree kernel sees the total density, the LDOS model both channels (mixing.jl:241-257)."""
    lat = A_AL / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
    Al = dftk.ElementPsp("Al", dftk.load_psp("Al", "lda"))
    model = dftk.model_DFT(lat, [Al], [np.zeros(3)], functionals=("lda_x", "lda_c_pw"), temperature=0.01, smearing="fermi_dirac",
                           magnetic_moments=[0.5])
    basis = dftk.PlaneWaveBasis(model, 6, dftk.MonkhorstPack((2, 2, 2)))
    st = dftk.ScfStepper(basis, tol=1e-12, rho=dftk.guess_density(basis, [0.5]))
    info = st.step()
    dF = (info["rho"] - info["rho_in"]).contiguous()
    assert dF.dim() == 4
    args = dict(eF=info["eF"], eigenvalues=info["eigenvalues"], psi=info["psi"], occupation=info["occupation"])
    nat = dftk.LdosMixing(reltol=1e-10)
    got = nat.mix_density(basis, dF.clone(), **args)
    monkeypatch.setenv("DFTK_MI_TORCH_MIX", "1")
    twin = dftk.LdosMixing(reltol=1e-10)
    ref = twin.mix_density(basis, dF.clone(), **args)
    monkeypatch.delenv("DFTK_MI_TORCH_MIX")
    assert float((got - ref).norm()) < 1e-8 * float(ref.norm())
    assert nat.last_gmres_applies >= 2 and float((ref - dF).norm()) > 1e-3 * float(dF.norm())


def test_step_sums_one_kernel_one_fetch(state, monkeypatch):
    """``dftk_mi_step_sums``: int V_in rho_out and ||rho_out - rho_in||^2 of an SCF step in one kernel / one fetch, against
    NumPy; the stepper's energies and density change with the call equal those of the torch twins (DFTK_MI_TORCH_LOCAL=1)."""
    import ctypes as C
    db = state["db"] if isinstance(state, dict) and "db" in state else None
    if db is None:
        dm, _ = _al_models()
        db = dftk.PlaneWaveBasis(dm, 6, dftk.MonkhorstPack((1, 2, 2)))
    rng = np.random.default_rng(5)
    n = int(np.prod(db.fft_size))
    a, b, c = (rng.standard_normal(n) for _ in range(3))
    ad, bd, cd = (torch.tensor(x, device="cuda") for x in (a, b, c))
    out = (C.c_double * 2)()
    assert db.lib.dftk_mi_step_sums(db.handle, n, ad.data_ptr(), bd.data_ptr(), cd.data_ptr(), out) == 0
    assert abs(out[0] - float(a @ b)) < 1e-11 * np.sqrt(n) and abs(out[1] - float(((a - c) ** 2).sum())) < 1e-11 * n
    assert db.lib.dftk_mi_step_sums(db.handle, n, ad.data_ptr(), None, cd.data_ptr(), out) == 0 and out[0] == 0.0
    assert db.lib.dftk_mi_step_sums(db.handle, n, ad.data_ptr(), None, None, out) != 0
    # the stepper's wrapper on its own arrays against the torch formulas it replaces; the torch twins under DFTK_MI_TORCH_LOCAL=1
    st = dftk.ScfStepper(db, tol=1e-10, seed=3)
    info = st.step()
    rho_out, rho_in, v_in = info["rho"], info["rho_in"], st._ritz_potential(info["ham"])
    s0, s1 = st._step_sums(rho_out, v_in, rho_in)
    assert abs(s0 - float((rho_out * v_in).sum().item())) < 1e-10 * max(1.0, abs(s0))
    assert abs(np.sqrt(s1) - float(torch.linalg.norm(rho_out - rho_in).item())) < 1e-12 * max(1.0, np.sqrt(s1))
    assert abs(info["history_drho"][-1] - float(torch.linalg.norm(rho_out - rho_in).item()) * st.sqrt_dvol) < 1e-12
    monkeypatch.setenv("DFTK_MI_TORCH_LOCAL", "1")
    assert st._step_sums(rho_out, v_in, rho_in) is None
    monkeypatch.delenv("DFTK_MI_TORCH_LOCAL", raising=False)


def test_density_and_ldos_in_one_pass(state):
    """``dftk_mi_density_accumulate_multi2``: compute_density and compute_ldos (dos.jl:43-62, "compute_density with modified
    weights") of the same orbitals from ONE pass over the bands -- both cubes equal to those of the two separate passes
    (same kernels, same order of the bands: 1e-14 relative) and to the oracle's; the hand-over to the chi0 mixing gives the same
    preconditioned residual as the mixing's own LDOS pass."""
    s = state
    db, ob = s["db"], s["ob"]
    from oracle import mixing as om
    assert db.kbatch, "the fixture's k-mesh must take the batched multi-k pipeline"
    occ, _ = dftk.compute_occupation(db, s["lam"])
    mix = dftk.LdosMixing()
    extra = mix.extra_density_weights(db, s["lam"], s["eF"], s["dpsi"])
    assert extra is not None
    rho, ldos = dftk.compute_density(db, s["dpsi"], occ, 1e-6, extra_weights=extra[0], extra_threshold=extra[1])
    rho_1 = dftk.compute_density(db, s["dpsi"], occ, 1e-6)
    sm, T = dftk.mixing.default_smearing_temperature(db.model)
    ldos_1 = dftk.compute_ldos(s["eF"], db, s["lam"], s["dpsi"], sm, T)
    assert float((rho - rho_1).abs().max()) < 1e-14 * float(rho_1.abs().max())
    assert float((ldos - ldos_1).abs().max()) < 1e-14 * float(ldos_1.abs().max())
    l0 = om.compute_ldos(s["eF"], ob, s["lam"], s["psi"], sm, T)
    assert np.linalg.norm(ldos.cpu().numpy() - l0) < 1e-12 * np.linalg.norm(l0)
    a = mix.mix_density(db, s["dFd"], eF=s["eF"], eigenvalues=s["lam"], psi=s["dpsi"], ldos=ldos)
    b = mix.mix_density(db, s["dFd"], eF=s["eF"], eigenvalues=s["lam"], psi=s["dpsi"])
    assert float((a - b).abs().max()) < 1e-12 * float(b.abs().max())
    # argument checks of the entry point: second weights without a second cube (and vice versa), one cube for both
    import ctypes as C
    kbs = (C.c_void_p * 1)(db.kpoints[0].handle.value)
    nbs = (C.c_int * 1)(1)
    pp = (C.c_void_p * 1)(s["dpsi"][0].data_ptr())
    ld = (C.c_int64 * 1)(s["dpsi"][0].stride(0))
    w = np.ones(1)
    r = torch.zeros_like(rho)
    assert db.lib.dftk_mi_density_accumulate_multi2(1, kbs, nbs, pp, ld, w.ctypes.data, r.data_ptr(), w.ctypes.data, None) != 0
    assert db.lib.dftk_mi_density_accumulate_multi2(1, kbs, nbs, pp, ld, w.ctypes.data, r.data_ptr(), None, r.data_ptr()) != 0
    assert db.lib.dftk_mi_density_accumulate_multi2(1, kbs, nbs, pp, ld, w.ctypes.data, r.data_ptr(), w.ctypes.data,
                                                    r.data_ptr()) != 0
