"""Real-symmetric orbitals at the Gamma point (include/dftk_mi355x.h: dftk_mi_kblock_set_gamma_real; an EXTENSION, the
reference has no Gamma special case): every piece against NumPy / the general complex path of the same library /
the CPU oracle, then the eigensolver and the SCF.

The half-sphere format (row 0 = x(G = 0), rows j > 0 = sqrt(2) x(G_j)) must make the REAL matrix products over half
the rows reproduce the complex inner products of the full vectors; H in that format must equal
compress . H_full . expand; LOBPCG must return the eigenvalues of the complex iteration; the SCF must end at the same
energies and density.  Tolerances: fp64 round-off (1e-12 relative) for the linear pieces, the solver tolerances for
the iterations.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd._lib import check, cplx  # noqa: E402

from oracle import energy_hamiltonian  # noqa: E402
from test_gpu_kernels import Basis, KBlock, dev, make_oracle_basis, relerr, run_lobpcg  # noqa: E402

GEMM_REAL = 8
EINVAL = -1         # DFTK_MI_EINVAL
S2 = np.sqrt(2.0)


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return dftk.load_library()


def pair_tables(fft_size, mapping):
    """NumPy restatement of the pair tables: row / partner row of one representative per {G, -G}, G = 0 first."""
    nx, ny, nz = fft_size
    m = np.asarray(mapping)
    ix, iy, iz = m % nx, (m // nx) % ny, m // (nx * ny)
    minus = (-ix) % nx + nx * ((-iy) % ny + ny * ((-iz) % nz))
    row = {int(v): i for i, v in enumerate(m)}
    partner = np.array([row[int(v)] for v in minus])
    first = np.nonzero(partner >= np.arange(len(m)))[0]
    return first, partner[first]


def to_half(x, g, mg):
    h = (x[g] + np.conj(x[mg])) / 2 * S2
    h[0] = np.real(x[g[0]])
    return h


def from_half(h, g, mg, n):
    x = np.zeros((n,) + h.shape[1:], dtype=complex)
    x[g] = h / S2
    x[mg] = np.conj(h) / S2
    x[g[0]] = np.real(h[0])
    return x


def gamma_block(lib, Ecut=10, fft_size=(21, 21, 21), seed=0, terms=("Kinetic", "AtomicLocal", "AtomicNonlocal")):
    """A Gamma k-block with a generic real potential and the HGH projectors, real mode switched on."""
    obasis = make_oracle_basis(Ecut, fft_size, kcoords=[[0, 0, 0]], terms=terms)
    rng = np.random.default_rng(seed)
    nx, ny, nz = fft_size
    _, ham = energy_hamiltonian(obasis, None, None)
    H = ham[0]
    V = None
    if "AtomicLocal" in terms:
        V = obasis.terms.V_loc + 0.1 * rng.standard_normal((nz, ny, nx))
        H.potential = V
    bs = Basis(lib, nx, ny, nz, obasis.model.unit_cell_volume)
    kb = KBlock(lib, bs, obasis.kpoints[0].mapping, H.kinetic)
    if "AtomicNonlocal" in terms:
        kb.set_projectors(H.P, H.D)
    if V is not None:
        kb.set_potential(V)
    check(lib.dftk_mi_kblock_set_gamma_real(kb.h, 1))
    nh = C.c_int64()
    check(lib.dftk_mi_gamma_half_size(kb.h, C.byref(nh)))
    g, mg = pair_tables(fft_size, obasis.kpoints[0].mapping)
    assert nh.value == len(g) == (kb.n_G + 1) // 2
    return obasis, H, bs, kb, g, mg, rng


def symmetric_block(rng, n, m, g, mg):
    h = rng.standard_normal((len(g), m)) + 1j * rng.standard_normal((len(g), m))
    h[0] = h[0].real
    return h, from_half(h, g, mg, n)


@pytest.mark.parametrize("trans,m,n,k,flags", [("C", 37, 50, 1000, 0), ("C", 259, 259, 30011, 1), ("C", 130, 70, 2049, 0),
                                                ("C", 300, 99, 5000, 0), ("C", 5, 3, 40001, 0), ("C", 777, 777, 66000, 1),
                                                ("N", 1000, 37, 50, 0), ("N", 5000, 259, 777, 0), ("N", 3000, 36, 100, 0),
                                                ("N", 66000, 300, 300, 2), ("N", 127, 34, 13, 0), ("N", 1, 1, 1, 0),
                                                # triangular B at the widths of ortho!(X): odd row counts, n = 503 / 512 / 259 / 64
                                                ("N", 4099, 503, 503, 2), ("N", 3001, 512, 512, 2), ("N", 2500, 130, 130, 2),
                                                ("N", 2048, 64, 64, 2), ("N", 20011, 259, 259, 2)])
def test_zgemm_real_flag(lib, trans, m, n, k, flags):
    """DFTK_MI_GEMM_REAL: 'C' = Re(A^H B) with a zero imaginary part, 'N' = A Re(B); with alpha, beta, UPPER (1),
    B_UPPER (2), ragged tiles and the split-K path."""
    rng = np.random.default_rng(m + 3 * n + 7 * k)
    bs = Basis(lib, 8, 8, 8)
    A = rng.standard_normal((m, k) if trans == "N" else (k, m)) + 1j * rng.standard_normal((m, k) if trans == "N" else (k, m))
    B = rng.standard_normal((k, n)) + 1j * rng.standard_normal((k, n))
    if flags & 2:
        B = np.triu(B)
    C0 = rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))
    alpha, beta = 0.7, -0.3
    Ad, Bd, Cd = dev(A.T.copy()), dev(B.T.copy()), dev(C0.T.copy())
    check(lib.dftk_mi_zgemm_ex(bs.h, trans.encode(), m, n, k, cplx(alpha), Ad.data_ptr(), A.shape[0], Bd.data_ptr(), k,
                               cplx(beta), Cd.data_ptr(), m, flags | GEMM_REAL))
    bs.sync()
    prod = np.real(A.conj().T @ B) if trans == "C" else A @ np.real(B)
    ref = alpha * prod + beta * C0
    got = Cd.cpu().numpy().T
    if flags & 1:      # only tiles that intersect the upper triangle are written: compare the upper triangle
        iu = np.triu_indices(m, 0, n)
        assert relerr(got[iu], ref[iu]) < 1e-13
    else:
        assert relerr(got, ref) < 1e-13
    # beta = 0: exact zeros in the imaginary part of a Gram matrix, no read of C
    Cn = torch.full_like(Cd, float("nan"))
    check(lib.dftk_mi_zgemm_ex(bs.h, trans.encode(), m, n, k, cplx(1.0), Ad.data_ptr(), A.shape[0], Bd.data_ptr(), k,
                               cplx(0.0), Cn.data_ptr(), m, flags | GEMM_REAL))
    bs.sync()
    got = Cn.cpu().numpy().T
    sel = np.triu_indices(m, 0, n) if flags & 1 else np.nonzero(np.ones((m, n)))
    assert relerr(got[sel], prod[sel]) < 1e-13
    if trans == "C":
        assert not got[sel].imag.any()


def test_pair_tables_and_roundtrip(lib):
    obasis, H, bs, kb, g, mg, rng = gamma_block(lib)
    n, nh = kb.n_G, len(g)
    nx, ny, nz = obasis.fft_size
    cnt = C.c_int64()
    rows = np.zeros(nh, dtype=np.int32)
    prt = np.zeros(nh, dtype=np.int32)
    m = np.ascontiguousarray(obasis.kpoints[0].mapping, dtype=np.int64)
    check(lib.dftk_mi_gamma_tables_host(nx, ny, nz, n, m.ctypes.data, C.byref(cnt), rows.ctypes.data, prt.ctypes.data))
    assert cnt.value == nh and np.array_equal(rows, g) and np.array_equal(prt, mg) and rows[0] == prt[0] == 0
    # compress = scaled real-symmetric part of ANY complex block; expand its exact inverse on symmetric vectors
    X = rng.standard_normal((n, 5)) + 1j * rng.standard_normal((n, 5))
    Xd = dev(X.T.copy())
    Hd = torch.full((5, nh + 3), float("nan"), dtype=torch.complex128, device="cuda")
    check(lib.dftk_mi_gamma_compress(kb.h, 5, Xd.data_ptr(), n, Hd.data_ptr(), nh + 3))
    bs.sync()
    h = Hd.cpu().numpy()[:, :nh].T
    assert relerr(h, to_half(X, g, mg)) < 1e-14 and not h[0].imag.any()
    Yd = torch.full((5, n), float("nan"), dtype=torch.complex128, device="cuda")
    check(lib.dftk_mi_gamma_expand(kb.h, 5, Hd.data_ptr(), nh + 3, Yd.data_ptr(), n))
    bs.sync()
    Y = Yd.cpu().numpy().T
    assert relerr(Y, from_half(h, g, mg, n)) < 1e-14
    # the symmetric part is what a real field looks like: IFFT of the expanded vector is real
    cube = obasis.ifft(obasis.kpoints[0], Y[:, 0], normalize=False)
    assert np.abs(cube.imag).max() < 1e-12 * np.abs(cube).max()
    # real products of the half format = complex inner products of the full vectors
    Gd = torch.full((5, 5), float("nan"), dtype=torch.complex128, device="cuda")
    check(lib.dftk_mi_zgemm_ex(bs.h, b"C", 5, 5, nh, cplx(1.0), Hd.data_ptr(), nh + 3, Hd.data_ptr(), nh + 3, cplx(0.0),
                               Gd.data_ptr(), 5, GEMM_REAL))
    bs.sync()
    assert relerr(Gd.cpu().numpy().T, Y.conj().T @ Y) < 1e-13


@pytest.mark.parametrize("Ecut,fft_size,nbands", [(10, (21, 21, 21), 5), (15, (27, 27, 27), 12), (12, (24, 25, 27), 1),
                                                   (8, (20, 25, 24), 9)])
def test_gamma_apply_H_equals_general_apply(lib, Ecut, fft_size, nbands):
    """H in the half format = compress . H . expand, per part and in total, odd and even band counts; and against the
    oracle's H on the expanded vectors."""
    obasis, H, bs, kb, g, mg, rng = gamma_block(lib, Ecut, fft_size, seed=nbands)
    n, nh = kb.n_G, len(g)
    h, X = symmetric_block(rng, n, nbands, g, mg)
    hd = dev(h.T.copy())
    for which in (1, 2, 4, 3, 7):
        out = torch.full_like(hd, float("nan"))
        check(lib.dftk_mi_gamma_apply_H(kb.h, which, nbands, hd.data_ptr(), nh, out.data_ptr(), nh))
        bs.sync()
        got = out.cpu().numpy().T
        full = kb.apply(X, which)
        assert relerr(from_half(got, g, mg, n), full) < 1e-12, which
        assert not got[0].imag.any()
    assert relerr(from_half(got, g, mg, n), H.mul(X)) < 1e-12


def test_density_accumulate_real(lib):
    obasis, H, bs, kb, g, mg, rng = gamma_block(lib, 12, (24, 25, 27))
    n = kb.n_G
    nx, ny, nz = obasis.fft_size
    for M in (7, 8, 1):
        _, X = symmetric_block(rng, n, M, g, mg)
        w = rng.uniform(0, 2, M)
        w[M // 2] = 0.0
        Xd = dev(X.T.copy())
        rho_a = torch.zeros((nz, ny, nx), dtype=torch.float64, device="cuda")
        rho_b = torch.full_like(rho_a, 0.25)
        check(lib.dftk_mi_density_accumulate(kb.h, M, Xd.data_ptr(), n, w.ctypes.data, rho_a.data_ptr()))
        check(lib.dftk_mi_density_accumulate_real(kb.h, M, Xd.data_ptr(), n, w.ctypes.data, rho_b.data_ptr()))
        bs.sync()
        assert relerr((rho_b - 0.25).cpu().numpy(), rho_a.cpu().numpy()) < 1e-12


@pytest.mark.parametrize("terms", [("Kinetic",), ("Kinetic", "AtomicLocal", "AtomicNonlocal")])
def test_lobpcg_real_mode_matches_complex(lib, terms):
    """Same block, same random complex start vectors: eigenvalues of the real-symmetric iteration equal those of the
    general complex one and the dense diagonalisation; the returned vectors are real-symmetric, orthonormal
    eigenvectors of the GENERAL operator."""
    obasis, H, bs, kb, g, mg, rng = gamma_block(lib, 10, (21, 21, 21), terms=terms)
    n, M = kb.n_G, 12
    X0 = np.linalg.qr(rng.standard_normal((n, M)) + 1j * rng.standard_normal((n, M)))[0]
    lam_r, res_r, nit_r, conv_r, nmv_r, Xr = run_lobpcg(lib, kb, X0, 1e-9, n_conv_check=8)
    check(lib.dftk_mi_kblock_set_gamma_real(kb.h, 0))
    lam_c, res_c, nit_c, conv_c, nmv_c, Xc = run_lobpcg(lib, kb, X0, 1e-9, n_conv_check=8)
    assert conv_r == 1 and conv_c == 1
    np.testing.assert_allclose(lam_r[:8], lam_c[:8], atol=1e-9)
    dense = np.linalg.eigvalsh(H.to_dense()) if hasattr(H, "to_dense") else None
    if dense is not None:
        np.testing.assert_allclose(lam_r[:8], dense[:8], atol=1e-8)
    assert abs(nit_r - nit_c) <= 10
    assert relerr(from_half(to_half(Xr, g, mg), g, mg, n), Xr) < 1e-14          # real-symmetric
    assert np.linalg.norm(Xr.conj().T @ Xr - np.eye(M)) < 1e-10
    HX = kb.apply(Xr, 7)                                                         # general complex apply
    R = HX - Xr * lam_r[None, :]
    rn = np.linalg.norm(R, axis=0)
    live = res_r > 0                       # (locked columns report 0 at the final iteration, as the reference's history)
    np.testing.assert_allclose(rn[live], res_r[live], atol=1e-10)
    assert rn[:8].max() < 1e-9


def test_lobpcg_entry_phase_alignment(lib):
    """Entry of the Gamma-real LOBPCG (``dftk_mi_gamma_compress_aligned``): every start vector is rotated by the global
    phase that maximises its real-symmetric part, then compressed -- device kernel vs the oracle's ``align_phase``;
    real-symmetric columns pass bit for bit; a real field times ANY phase (incl. i, whose plain projection vanishes)
    keeps all of its norm, so converged orbitals of the complex iteration are taken over losslessly: the Gamma-real
    LOBPCG started from them converges at once instead of re-randomising columns."""
    import oracle.gamma_real as ogr
    obasis, H, bs, kb, g, mg, rng = gamma_block(lib, 10, (21, 21, 21), terms=("Kinetic", "AtomicLocal", "AtomicNonlocal"))
    n, nh, M = kb.n_G, len(g), 12

    def compress(fn, X):
        Xd = dev(X.T.copy())
        Hd = torch.full((X.shape[1], nh), float("nan"), dtype=torch.complex128, device="cuda")
        check(fn(kb.h, X.shape[1], Xd.data_ptr(), n, Hd.data_ptr(), nh))
        bs.sync()
        return Hd.cpu().numpy().T

    X = rng.standard_normal((n, 6)) + 1j * rng.standard_normal((n, 6))
    got = compress(lib.dftk_mi_gamma_compress_aligned, X)
    ref = to_half(ogr.align_phase(X, g, mg), g, mg)
    assert relerr(got, ref) < 1e-13
    assert np.all(np.linalg.norm(got, axis=0) >= np.linalg.norm(compress(lib.dftk_mi_gamma_compress, X), axis=0) - 1e-12)
    S = from_half(to_half(X, g, mg), g, mg, n)                   # real-symmetric columns
    assert np.array_equal(compress(lib.dftk_mi_gamma_compress_aligned, S), compress(lib.dftk_mi_gamma_compress, S))
    ph = np.exp(1j * np.array([0.0, np.pi / 2, 0.3, -2.1, np.pi, 1.0]))
    rot = compress(lib.dftk_mi_gamma_compress_aligned, S * ph[None, :])
    base = compress(lib.dftk_mi_gamma_compress, S)
    for c in range(6):                                            # +- the unrotated image: nothing lost
        assert min(np.linalg.norm(rot[:, c] - base[:, c]), np.linalg.norm(rot[:, c] + base[:, c])) < 1e-12 * np.linalg.norm(base[:, c])
    assert np.linalg.norm(compress(lib.dftk_mi_gamma_compress, S[:, :1] * 1j)) < 1e-12     # the plain projection of i * real
    # warm start from the complex iteration's orbitals, each multiplied by an arbitrary phase
    X0 = np.linalg.qr(rng.standard_normal((n, M)) + 1j * rng.standard_normal((n, M)))[0]
    check(lib.dftk_mi_kblock_set_gamma_real(kb.h, 0))
    lam_c, _, _, conv_c, _, Xc = run_lobpcg(lib, kb, X0, 1e-9, n_conv_check=8)
    check(lib.dftk_mi_kblock_set_gamma_real(kb.h, 1))
    # (non-degenerate eigenvectors of the real-symmetric operator are real fields times a phase)
    warm = Xc * np.exp(1j * rng.uniform(0, 2 * np.pi, M))[None, :]
    lam_w, res_w, nit_w, conv_w, nmv_w, Xw = run_lobpcg(lib, kb, warm, 1e-7, n_conv_check=8)
    assert conv_c == 1 and conv_w == 1 and nit_w <= 3
    np.testing.assert_allclose(lam_w[:8], lam_c[:8], atol=1e-9)


def test_scf_gamma_real_equals_complex():
    """Gamma-only silicon supercell: the SCF with real-symmetric orbitals (automatic at k = 0) ends at the energies,
    density, eigenvalues and SCF length of the general complex path."""
    lat, atoms, pos = dftk.silicon_cell((2, 1, 1))
    out = {}
    for mode in (None, False):
        model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
        basis = dftk.PlaneWaveBasis(model, 12, dftk.MonkhorstPack((1, 1, 1)), gamma_real=mode)
        assert basis.kpoints[0].gamma_real is (mode is None)
        out[mode] = (basis, dftk.self_consistent_field(basis, tol=1e-10))
    (b1, r1), (b0, r0) = out[None], out[False]
    assert r1["converged"] and r0["converged"]
    assert abs(r1["energies"].total - r0["energies"].total) < 1e-9
    for name, v in r0["energies"].items():
        assert abs(r1["energies"][name] - v) < 1e-8, name
    nconv = r0["n_bands_converge"]
    np.testing.assert_allclose(r1["eigenvalues"][0][:nconv], r0["eigenvalues"][0][:nconv], atol=1e-8)
    assert float(torch.linalg.norm(r1["rho"] - r0["rho"])) * np.sqrt(b0.dvol) < 1e-8
    assert abs(r1["n_iter"] - r0["n_iter"]) <= 8      # (tol = 1e-10 sits in the round-off tail of the Anderson iteration)
    # restart from the converged real-symmetric orbitals: one cheap step
    r2 = dftk.self_consistent_field(b1, rho=r1["rho"], psi=r1["psi"], tol=1e-8)
    assert r2["converged"] and r2["n_iter"] <= 3


def test_gamma_real_refusals(lib):
    """k != 0 spheres, Nyquist points and non-symmetric projectors are refused (the general path remains)."""
    obasis = make_oracle_basis(10, (21, 21, 21), kcoords=[[1 / 3, 0.1, -0.25]], terms=("Kinetic",))
    _, ham = energy_hamiltonian(obasis, None, None)
    bs = Basis(lib, 21, 21, 21, obasis.model.unit_cell_volume)
    kb = KBlock(lib, bs, obasis.kpoints[0].mapping, ham[0].kinetic)
    assert lib.dftk_mi_kblock_set_gamma_real(kb.h, 1) == EINVAL
    # the whole 8^3 cube as a "sphere": Nyquist rows are their own partners
    bs2 = Basis(lib, 8, 8, 8)
    kb2 = KBlock(lib, bs2, np.arange(512), np.zeros(512))
    assert lib.dftk_mi_kblock_set_gamma_real(kb2.h, 1) == EINVAL
    # projectors with a random phase per row are not transforms of real functions
    obasis, H, bs3, kb3, g, mg, rng = gamma_block(lib, 10, (21, 21, 21), terms=("Kinetic", "AtomicNonlocal"))
    P = H.P * np.exp(1j * rng.uniform(0, 6, (H.P.shape[0], 1)))
    kb3.set_projectors(P, H.D)
    nh = len(g)
    hd = dev(np.ones((2, nh), dtype=complex))
    out = torch.empty_like(hd)
    assert lib.dftk_mi_gamma_apply_H(kb3.h, 7, 2, hd.data_ptr(), nh, out.data_ptr(), nh) == EINVAL
    assert b"real-symmetric" in lib.dftk_mi_last_error()


@pytest.mark.parametrize("n", [5, 33, 259, 1006, 1509])
def test_heev_real_symmetric_input(lib, n):
    """dftk_mi_heev on a real symmetric matrix (zero imaginary parts, what the Gamma-real Rayleigh-Ritz hands it):
    the real-rotation Jacobi path -- eigenvalues against LAPACK, real orthonormal eigenvectors, A V = V diag(w)."""
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    A = (A + A.T) / 2 + np.diag(np.linspace(-3, 3, n))
    Ad = dev(A.astype(complex).T.copy())
    Vd = torch.full((n, n), float("nan"), dtype=torch.complex128, device="cuda")
    w = np.zeros(n)
    bs = Basis(lib, 8, 8, 8)
    check(lib.dftk_mi_heev(bs.h, n, Ad.data_ptr(), n, w.ctypes.data, Vd.data_ptr(), n))
    bs.sync()
    V = Vd.cpu().numpy().T
    np.testing.assert_allclose(w, np.linalg.eigvalsh(A), atol=1e-12 * max(1.0, np.abs(A).max()) * n)
    assert not V.imag.any()
    assert np.linalg.norm(V.T @ V - np.eye(n)) < 1e-12 * n
    assert np.linalg.norm(A @ V.real - V.real * w[None, :]) < 1e-12 * n * np.linalg.norm(A)
