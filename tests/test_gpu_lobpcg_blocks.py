"""Direct oracle comparisons of the LOBPCG building blocks behind the C ABI (SURVEY.md section 8a rows
``PreconditionerTPA`` and ``columnwise_norms/dots, ortho_qr``) and of the LOBPCG TRAJECTORY: eigenvalues do not
depend on the preconditioner, so a wrong ``mean_kin``, a wrong locking decision or a wrong deferred
preconditioning would only show up as extra iterations -- the residual history from identical start vectors
is what pins them (device vs ``oracle.LOBPCG`` iteration by iteration until round-off separates them).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd._lib import check, dftk_mi_cplx  # noqa: E402

import oracle  # noqa: E402
from oracle.lobpcg import (LOBPCG, PreconditionerTPA, columnwise_dots, columnwise_norms)  # noqa: E402
from test_gpu_kernels import Basis, KBlock, dev, make_oracle_basis, relerr, run_lobpcg  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return dftk.load_library()


def _block(rng, n, m):
    return rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m))


@pytest.mark.parametrize("n,m,ld", [(1000, 7, 1000), (4099, 33, 4200), (135491, 5, 135491)])
def test_columnwise_norms_and_dots(lib, n, m, ld):
    """columnwise_norms / columnwise_dots (src/common/linalg.jl:2-15; GPU forms src/gpu/linalg.jl:17-27)."""
    rng = np.random.default_rng(n + m)
    bs = Basis(lib, 8, 8, 8)
    A, B = _block(rng, n, m), _block(rng, n, m)
    Ad = torch.zeros((m, ld), dtype=torch.complex128, device="cuda")
    Bd = torch.zeros((m, ld), dtype=torch.complex128, device="cuda")
    Ad[:, :n] = dev(A.T.copy())
    Bd[:, :n] = dev(B.T.copy())
    norms = np.zeros(m)
    check(lib.dftk_mi_columnwise_norms(bs.h, n, m, Ad.data_ptr(), ld, norms.ctypes.data))
    np.testing.assert_allclose(norms, columnwise_norms(A), rtol=1e-13)
    dots = (dftk_mi_cplx * m)()
    check(lib.dftk_mi_columnwise_dots(bs.h, n, m, Ad.data_ptr(), ld, Bd.data_ptr(), ld, dots))
    got = np.array([complex(d.re, d.im) for d in dots])
    ref = columnwise_dots(A, B)
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max() * np.sqrt(n)


def _tpa_setup(lib, Ecut=10, fft=(21, 21, 21)):
    obasis = make_oracle_basis(Ecut, fft, kcoords=[[0.1, -0.2, 0.3]])
    _, ham = oracle.energy_hamiltonian(obasis, None, None)
    H, kpt = ham[0], obasis.kpoints[0]
    bs = Basis(lib, *fft, obasis.model.unit_cell_volume)
    kb = KBlock(lib, bs, kpt.mapping, H.kinetic)
    kb.set_projectors(H.P, H.D)
    kb.set_potential(H.potential)
    return obasis, H, bs, kb


def test_tpa_precondprep_and_ldiv_match_oracle(lib):
    """PreconditionerTPA (src/eigen/preconditioners.jl:50-77): mean_kin of precondprep!, ldiv! with and without
    a prepared mean_kin (default_shift form)."""
    _, H, bs, kb = _tpa_setup(lib)
    rng = np.random.default_rng(3)
    n, m = H.n_G, 9
    X = np.linalg.qr(_block(rng, n, m))[0]
    R = _block(rng, n, m)
    Xd, Rd = dev(X.T.copy()), dev(R.T.copy())
    P = PreconditionerTPA(H.kinetic)
    # before precondprep!: Y = R ./ (kin .+ default_shift)
    Yd = torch.empty_like(Rd)
    check(lib.dftk_mi_tpa_ldiv(kb.h, m, Rd.data_ptr(), n, None, 1.0, Yd.data_ptr(), n))
    bs.sync()
    assert relerr(Yd.cpu().numpy().T, P.ldiv(R)) < 1e-14
    check(lib.dftk_mi_tpa_ldiv(kb.h, m, Rd.data_ptr(), n, None, 0.37, Yd.data_ptr(), n))
    bs.sync()
    assert relerr(Yd.cpu().numpy().T, R / (H.kinetic + 0.37)[:, None]) < 1e-14
    # precondprep!
    mk = np.zeros(m)
    check(lib.dftk_mi_tpa_precondprep(kb.h, m, Xd.data_ptr(), n, mk.ctypes.data))
    P.precondprep(X)
    np.testing.assert_allclose(mk, P.mean_kin, rtol=1e-13)
    check(lib.dftk_mi_tpa_ldiv(kb.h, m, Rd.data_ptr(), n, mk.ctypes.data, 1.0, Yd.data_ptr(), n))
    bs.sync()
    assert relerr(Yd.cpu().numpy().T, P.ldiv(R)) < 1e-14


def test_fused_residual_pass_matches_oracle(lib):
    """The fused pass of one iteration (lobpcg_hyper_impl.jl:441-449, :533): R = AX - X lambda, its column norms,
    precondprep!'s mean kinetic energies and <x, x> from ONE read of X."""
    _, H, bs, kb = _tpa_setup(lib)
    rng = np.random.default_rng(4)
    n, m = H.n_G, 11
    X = np.linalg.qr(_block(rng, n, m))[0] * (1 + 1e-3 * rng.standard_normal(m))[None, :]
    AX = H.mul(X)
    lam = np.ascontiguousarray(np.real(columnwise_dots(X, AX) / columnwise_dots(X, X)))
    Xd, AXd = dev(X.T.copy()), dev(AX.T.copy())
    Rd = torch.empty_like(Xd)
    norms, mk, xx = np.zeros(m), np.zeros(m), np.zeros(m)
    check(lib.dftk_mi_block_residual(kb.h, m, AXd.data_ptr(), n, Xd.data_ptr(), n, lam.ctypes.data, Rd.data_ptr(), n,
                                     norms.ctypes.data, mk.ctypes.data, xx.ctypes.data))
    R = AX - X * lam[None, :]
    assert relerr(Rd.cpu().numpy().T, R) < 1e-14
    np.testing.assert_allclose(norms, columnwise_norms(R), rtol=1e-12)
    np.testing.assert_allclose(mk, np.sum(np.abs(X) ** 2 * H.kinetic[:, None], axis=0), rtol=1e-13)
    np.testing.assert_allclose(xx, np.real(columnwise_dots(X, X)), rtol=1e-13)


@pytest.mark.parametrize("n,m", [(2000, 12), (30011, 259)])
def test_ortho_qr_cholesky_path(lib, n, m):
    """ortho_qr (src/common/ortho.jl:1-9) / ortho!(X) (lobpcg_hyper_impl.jl:216-261): same column space as
    Householder QR, Q = Q_lapack * unitary diagonal (Cholesky's R has a positive diagonal)."""
    rng = np.random.default_rng(n)
    bs = Basis(lib, 8, 8, 8)
    X = _block(rng, n, m) @ (np.eye(m) + 0.3 * _block(rng, m, m))       # mildly ill-conditioned
    Xd = dev(X.T.copy())
    nch, svd = C.c_int(), C.c_int()
    check(lib.dftk_mi_ortho_qr(bs.h, n, m, Xd.data_ptr(), n, 0, C.byref(nch), C.byref(svd)))
    Q = Xd.cpu().numpy().T
    assert svd.value == 0 and 1 <= nch.value <= 6
    assert np.linalg.norm(Q.conj().T @ Q - np.eye(m)) < 1e-12 * m
    Qref = np.linalg.qr(X)[0]
    D = np.diag(Qref.conj().T @ Q)
    np.testing.assert_allclose(np.abs(D), 1.0, atol=1e-9)
    assert relerr(Q, Qref * D[None, :]) < 1e-9


def test_ortho_svd_fallback_is_polar_factor(lib):
    """SVD fallback of ortho! (lobpcg_hyper_impl.jl:226-231): X <- U V'.  Full-rank input: the unitary polar
    factor exactly; rank-deficient input (two equal columns, one zero column): still an orthonormal basis whose
    span contains the input's column space."""
    rng = np.random.default_rng(9)
    bs = Basis(lib, 8, 8, 8)
    n, m = 5000, 10
    X = _block(rng, n, m) @ (np.eye(m) + 0.2 * _block(rng, m, m))
    Xd = dev(X.T.copy())
    nch, svd = C.c_int(), C.c_int()
    check(lib.dftk_mi_ortho_qr(bs.h, n, m, Xd.data_ptr(), n, 1, C.byref(nch), C.byref(svd)))
    U, _, Vh = np.linalg.svd(X, full_matrices=False)
    assert svd.value == 1 and nch.value == 100
    assert relerr(Xd.cpu().numpy().T, U @ Vh) < 1e-10
    Xbad = X.copy()
    Xbad[:, 3] = Xbad[:, 1]
    Xbad[:, 7] = 0.0
    Xd = dev(Xbad.T.copy())
    check(lib.dftk_mi_ortho_qr(bs.h, n, m, Xd.data_ptr(), n, 1, C.byref(nch), C.byref(svd)))
    Q = Xd.cpu().numpy().T
    assert np.linalg.norm(Q.conj().T @ Q - np.eye(m)) < 1e-11
    resid = Xbad - Q @ (Q.conj().T @ Xbad)
    assert np.linalg.norm(resid) < 1e-9 * np.linalg.norm(Xbad)


def test_lobpcg_recovers_from_rank_deficient_guess(lib):
    """A start block with a duplicated and a nearly dependent column must not return DFTK_MI_NUM_CHOLESKY: the
    reference recovers through safe_cholesky's shifts (lobpcg_hyper_impl.jl:190-210; the shifted factor turns the
    dependent column into amplified round-off, which the next passes orthonormalise) and converges to the same
    eigenvalues.  (An exactly ZERO column is not a valid input: it stays zero under X * inv(R), and the reference's
    ``while true`` in ortho! never terminates on it; the library reports status 2 after 30 passes.)"""
    _, H, bs, kb = _tpa_setup(lib)
    rng = np.random.default_rng(12)
    M = 6
    X0 = np.linalg.qr(_block(rng, H.n_G, M))[0]
    X0[:, 4] = X0[:, 2]
    X0[:, 5] = X0[:, 0] + 1e-9 * X0[:, 1]
    lam, res, nit, conv, nmv, X = run_lobpcg(lib, kb, X0, 1e-8, maxiter=200)
    assert conv == 1
    dense = np.linalg.eigvalsh(H.to_dense())[:M]
    np.testing.assert_allclose(lam, dense, atol=1e-8)
    assert np.linalg.norm(X.conj().T @ X - np.eye(M)) < 1e-10


@pytest.mark.parametrize("use_tpa", [1, 0])
def test_lobpcg_residual_history_matches_oracle(lib, use_tpa):
    """Same Hamiltonian, same start vectors, same tolerance: the device LOBPCG and ``oracle.LOBPCG`` (restatement of
    lobpcg_hyper_impl.jl:354-582) must walk the SAME trajectory -- number of iterations, locking pattern (zeros in
    the history after a column locks), residual norms per iteration -- until round-off separates them (relative
    1e-6 while the residuals are above 1e-7)."""
    _, H, bs, kb = _tpa_setup(lib, Ecut=12, fft=(24, 24, 24))
    rng = np.random.default_rng(21)
    M, ncc, tol = 10, 7, 1e-6
    X0 = np.linalg.qr(_block(rng, H.n_G, M))[0]
    lam, res, nit, conv, nmv, X = run_lobpcg(lib, kb, X0, tol, n_conv_check=ncc, use_tpa=use_tpa, maxiter=200)
    Mo, nio, nsvd = C.c_int(), C.c_int(), C.c_int()
    check(lib.dftk_mi_lobpcg_history(kb.h, C.byref(Mo), C.byref(nio), None, 0, C.byref(nsvd)))
    hist = np.zeros((nio.value + 1, Mo.value))
    check(lib.dftk_mi_lobpcg_history(kb.h, C.byref(Mo), C.byref(nio), hist.ctypes.data, hist.size, C.byref(nsvd)))
    hist = hist.T
    assert (Mo.value, nio.value, nsvd.value) == (M, nit, 0)
    np.testing.assert_array_equal(hist[:, -1], res)
    prec = PreconditionerTPA(H.kinetic) if use_tpa else None
    ores = LOBPCG(H.mul, X0, prec, tol, 200, miniter=1, n_conv_check=ncc)
    ohist = ores["residual_history"]
    assert conv == 1
    np.testing.assert_allclose(lam[:ncc], ores["λ"][:ncc], atol=1e-10)
    # LOBPCG amplifies round-off (different summation orders, Jacobi vs divide-and-conquer Ritz vectors) by a
    # factor of a few per iteration, so the two runs agree digit by digit early on and then drift apart
    # smoothly: compare the first iterations tightly and the whole run loosely
    nit_o = ohist.shape[1] - 1
    ncmp = min(nit, nit_o, 8) + 1
    dev_rel = np.abs(hist[:, :ncmp] - ohist[:, :ncmp]) / np.maximum(ohist[:, :ncmp], 1e-300)
    print("max relative deviation of the residual norms per iteration:", np.array2string(dev_rel.max(axis=0), precision=2))
    np.testing.assert_array_equal(hist[:, :ncmp] == 0.0, ohist[:, :ncmp] == 0.0)       # same locking pattern
    assert dev_rel[:, :4].max() < 1e-9
    assert dev_rel.max() < 1e-4
    assert abs(nit - nit_o) <= 2 + nit_o // 10, (nit, nit_o)
    assert abs(nmv - ores["n_matvec"]) <= M * (2 + nit_o // 10)


def test_lobpcg_reaches_a_tight_tolerance(lib):
    """LOBPCG driven to tol = 1e-12 (the reference's diagtol_min is 100 eps): every Rayleigh-Ritz solve on the way has
    to be accurate to round-off, or the residuals stall above the tolerance until maxiter.  Part of the suites that
    tests/test_gpu_partial_solver_forced.py re-runs with the partial-spectrum solver forced on from n = 24, where its
    sign iteration's acceptance threshold is what is being tested."""
    _, H, bs, kb = _tpa_setup(lib, Ecut=12, fft=(24, 24, 24))
    rng = np.random.default_rng(33)
    M, ncc, tol = 14, 10, 1e-12
    X0 = np.linalg.qr(_block(rng, H.n_G, M))[0]
    lam, res, nit, conv, nmv, X = run_lobpcg(lib, kb, X0, tol, n_conv_check=ncc, maxiter=100)
    assert conv == 1 and nit < 60, (conv, nit, res)
    assert res[:ncc].max() < tol
    dense = np.linalg.eigvalsh(H.to_dense())[:ncc]
    np.testing.assert_allclose(lam[:ncc], dense, atol=1e-11)
    true_res = np.linalg.norm(H.mul(X[:, :ncc]) - X[:, :ncc] * lam[:ncc], axis=0)
    assert true_res.max() < 5e-12, true_res


def test_host_mirror_helpers_match_oracle():
    """The reference-named helpers of the host mirror (PreconditionerTPA.precondprep_ / ldiv_, columnwise_norms,
    columnwise_dots, ortho_qr) on band-major torch blocks against the oracle."""
    lat, atoms, pos = dftk.silicon_cell()
    basis = dftk.PlaneWaveBasis(dftk.model_DFT(lat, atoms, pos), 10, dftk.ExplicitKpoints([[0.1, -0.2, 0.3]], [1.0]))
    rho = dftk.guess_density(basis)
    _, ham = dftk.energy_hamiltonian(basis, None, None, rho=rho)
    H, kpt = ham[0], basis.kpoints[0]
    X = dftk.random_orbitals(basis, kpt, 6)
    R = dftk.random_orbitals(basis, kpt, 6)
    Xn, Rn = X.cpu().numpy().T, R.cpu().numpy().T
    np.testing.assert_allclose(dftk.columnwise_norms(basis, X), columnwise_norms(Xn), rtol=1e-13)
    np.testing.assert_allclose(dftk.columnwise_dots(basis, X, R), columnwise_dots(Xn, Rn), atol=1e-14)
    P, oP = dftk.PreconditionerTPA(H), PreconditionerTPA(kpt.kinetic.cpu().numpy())
    Y = torch.empty_like(R)
    assert relerr(P.ldiv_(Y, R).cpu().numpy().T, oP.ldiv(Rn)) < 1e-14           # default_shift form
    P.precondprep_(X)
    oP.precondprep(Xn)
    np.testing.assert_allclose(P.mean_kin, oP.mean_kin, rtol=1e-13)
    assert relerr(P.ldiv_(Y, R).cpu().numpy().T, oP.ldiv(Rn)) < 1e-14
    Q = dftk.ortho_qr(basis, X).cpu().numpy().T
    assert np.linalg.norm(Q.conj().T @ Q - np.eye(6)) < 1e-12
    Qref = np.linalg.qr(Xn)[0]
    assert np.linalg.norm(Q - Qref @ (Qref.conj().T @ Q)) < 1e-12                # same column space
