"""A X kept between the LOBPCG calls of consecutive SCF steps (lobpcg.cpp: lobpcg_run_general, dftk_mi_kblock_reuse_AX): the
next call starts from A_new X = (A_old X) inv(R) + (V_new - V_old) X instead of a full H X -- the kinetic and nonlocal parts of
H do not change between SCF steps (src/scf/self_consistent_field.jl:80-129).  Same eigenpairs and the same SCF as with the full
application, for real-symmetric Gamma orbitals and for a general complex k-point."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import dftk_jl_amd as dftk  # noqa: E402


def _count(lib):
    n = C.c_int64()
    assert lib.dftk_mi_ax_reuse_count(C.byref(n)) == 0
    return n.value


def _basis(kcoord):
    lat, atoms, pos = dftk.silicon_cell((2, 2, 2))
    model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
    return dftk.PlaneWaveBasis(model, 10, dftk.ExplicitKpoints([kcoord], [1.0]), coarse_start=False)


@pytest.mark.parametrize("kcoord", [[0.0, 0.0, 0.0], [0.25, 0.0, 0.125]])
def test_second_call_from_the_kept_AX_equals_the_full_application(kcoord):
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    basis = _basis(kcoord)
    lib = basis.lib
    assert bool(basis.kpoints[0].gamma_real) == (not any(kcoord))
    rho1 = dftk.guess_density(basis)
    z = torch.arange(basis.fft_size[2], device="cuda", dtype=torch.float64)
    rho2 = rho1 * (1.0 + 0.2 * torch.cos(2 * np.pi * z / basis.fft_size[2]))[:, None, None]
    gen = torch.Generator(device="cuda").manual_seed(3)
    X0 = dftk.random_orbitals(basis, basis.kpoints[0], 40, gen)
    out = {}
    for reuse in (True, False):
        _, ham1 = dftk.energy_hamiltonian(basis, None, None, rho=rho1)
        r1 = dftk.lobpcg_hyper(ham1[0], X0, prec=dftk.PreconditionerTPA(ham1[0]), tol=1e-7, n_conv_check=32)
        _, ham2 = dftk.energy_hamiltonian(basis, None, None, rho=rho2)
        c0 = _count(lib)
        r2 = dftk.lobpcg_hyper(ham2[0], r1.X, prec=dftk.PreconditionerTPA(ham2[0]), tol=1e-7, n_conv_check=32, reuse_AX=reuse)
        assert _count(lib) - c0 == (1 if reuse else 0)
        # a different number of bands, or a call that was not promised anything: the full application
        c1 = _count(lib)
        r3 = dftk.lobpcg_hyper(ham2[0], r2.X[:36], prec=dftk.PreconditionerTPA(ham2[0]), tol=1e-7, n_conv_check=32, reuse_AX=reuse)
        assert _count(lib) == c1 and r3.converged
        out[reuse] = r2
        if reuse:
            # new projectors (here: the same ones, bound again) drop what is kept: the promise is then ignored
            T = basis.terms
            kpt = basis.kpoints[0]
            D = np.asfortranarray(T.D.cpu().numpy() if torch.is_tensor(T.D) else T.D, dtype=np.float64)
            assert lib.dftk_mi_kblock_set_projectors(kpt.handle, T.P[0].shape[0], T.P[0].data_ptr(), T.P[0].stride(0),
                                                     D.ctypes.data) == 0
            c2 = _count(lib)
            r4 = dftk.lobpcg_hyper(ham2[0], r3.X, prec=dftk.PreconditionerTPA(ham2[0]), tol=1e-7, n_conv_check=32, reuse_AX=True)
            assert _count(lib) == c2 and r4.converged
    a, b = out[True], out[False]
    # (the supercell's spectrum is full of exactly degenerate clusters: the Ritz vectors inside a cluster, and with them the
    #  per-column residual norms, turn by O(1) under a 1e-15 perturbation of the start -- compared are the invariants)
    assert a.converged and b.converged and abs(a.n_iter - b.n_iter) <= 1
    np.testing.assert_allclose(a.λ[:32], b.λ[:32], rtol=0, atol=1e-9)
    assert a.residual_norms[:32].max() < 1e-7 and b.residual_norms[:32].max() < 1e-7
    # the TRUE residuals of the returned pairs under the NEW Hamiltonian (a full application) are below the tolerance: the
    # A X the call carried was the right one
    _, ham2 = dftk.energy_hamiltonian(basis, None, None, rho=rho2)
    HX = ham2[0] @ a.X
    true = torch.linalg.norm(HX - a.X * torch.as_tensor(a.λ, device="cuda")[:, None], dim=1).cpu().numpy()
    assert true[:32].max() < 1.05e-7, true[:32].max()


def test_scf_with_and_without_the_kept_AX(monkeypatch):
    basis = _basis([0.0, 0.0, 0.0])
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("DFTK_MI_AX_REUSE", flag)
        c0 = _count(basis.lib)
        r = dftk.self_consistent_field(basis, tol=1e-9, seed=2)
        assert r["converged"]
        res[flag] = (r["energies"].total, r["n_iter"], _count(basis.lib) - c0, r["eigenvalues"][0][:r["n_bands_converge"]].copy())
    monkeypatch.delenv("DFTK_MI_AX_REUSE", raising=False)
    assert res["1"][2] >= res["1"][1] - 3 and res["0"][2] == 0           # every step but the first (and rare multi-pass starts)
    assert abs(res["1"][0] - res["0"][0]) < 1e-9 * 16
    np.testing.assert_allclose(res["1"][3], res["0"][3], rtol=0, atol=1e-7)
    assert abs(res["1"][1] - res["0"][1]) <= 8
