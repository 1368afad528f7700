"""GPU parity tests at the C-ABI level: every kernel family against NumPy / the CPU oracle.

All calls go through include/dftk_mi355x.h (ctypes), device memory comes from torch.
Tolerances: the hot path is fp64 end to end; FFT/GEMM round-off only => 1e-12 relative
(the reference's own FFT identity tests use 1e-12, test/fourier_transforms.jl:20-28).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd._lib import check, cplx  # noqa: E402

from oracle import (ElementPsp, ExplicitKpoints, Model, MonkhorstPack, PlaneWaveBasis,  # noqa: E402
                    energy_hamiltonian, load_psp_hgh, model_DFT, guess_density, diagonalize_all_kblocks)
from oracle.scf import compute_density  # noqa: E402

RTOL = 1e-12


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return dftk.load_library()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


class Basis:
    def __init__(self, lib, nx, ny, nz, vol=1.0):
        self.lib = lib
        self.h = C.c_void_p()
        check(lib.dftk_mi_basis_create(nx, ny, nz, vol, 0, C.byref(self.h)))

    def sync(self):
        check(self.lib.dftk_mi_basis_sync(self.h))

    def __del__(self):
        try:
            self.lib.dftk_mi_basis_destroy(self.h)
        except Exception:
            pass


class KBlock:
    def __init__(self, lib, basis, mapping, kinetic):
        self.lib, self.basis = lib, basis
        self.h = C.c_void_p()
        m = np.ascontiguousarray(mapping, dtype=np.int64)
        k = np.ascontiguousarray(kinetic, dtype=np.float64)
        check(lib.dftk_mi_kblock_create(basis.h, len(m), m.ctypes.data, k.ctypes.data, C.byref(self.h)))
        self.n_G = len(m)
        self._keep = []

    def set_projectors(self, P, D):
        Pd = dev(np.asfortranarray(P).T.copy())     # (n_p, n_G) C-order == column-major n_G x n_p
        Dh = np.asfortranarray(D, dtype=np.float64)
        self._keep += [Pd, Dh]
        check(self.lib.dftk_mi_kblock_set_projectors(self.h, P.shape[1], Pd.data_ptr(), P.shape[0],
                                                     Dh.ctypes.data))

    def set_potential(self, V):
        Vd = dev(np.asarray(V, dtype=np.float64))
        check(self.lib.dftk_mi_kblock_set_potential(self.h, Vd.data_ptr()))
        self.basis.sync()

    def apply(self, psi, which=7):
        """psi: (n_G, M) numpy -> H psi (n_G, M)"""
        M = psi.shape[1]
        pd = dev(psi.T.copy())
        out = torch.full_like(pd, float("nan"))
        check(self.lib.dftk_mi_apply_H_parts(self.h, which, M, pd.data_ptr(), self.n_G, out.data_ptr(), self.n_G))
        self.basis.sync()
        return out.cpu().numpy().T

    def __del__(self):
        try:
            self.lib.dftk_mi_kblock_destroy(self.h)
        except Exception:
            pass


# ----------------------------------------------------------------------------------- dense algebra
@pytest.mark.parametrize("trans,m,n,k", [("N", 1000, 37, 50), ("N", 128, 64, 4), ("N", 33, 17, 129),
                                         ("C", 37, 50, 1000), ("C", 64, 128, 8192), ("C", 5, 3, 40001),
                                         ("N", 5000, 259, 777), ("C", 259, 259, 30011), ("C", 1, 1, 7),
                                         ("N", 1, 1, 1), ("C", 130, 70, 2049),
                                         # 1..4 remainder columns riding on the last full tile column
                                         ("C", 300, 99, 5000), ("N", 3000, 36, 100), ("C", 131, 33, 777),
                                         ("N", 127, 34, 13), ("C", 256, 68, 20000)])
def test_zgemm(lib, trans, m, n, k):
    rng = np.random.default_rng(m * 7 + n * 3 + k)
    bs = Basis(lib, 8, 8, 8)
    if trans == "N":
        A = rng.standard_normal((m, k)) + 1j * rng.standard_normal((m, k))
    else:
        A = rng.standard_normal((k, m)) + 1j * rng.standard_normal((k, m))
    B = rng.standard_normal((k, n)) + 1j * rng.standard_normal((k, n))
    C0 = rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))
    alpha, beta = 0.7 - 0.2j, -0.3 + 0.5j
    Ad, Bd, Cd = dev(A.T.copy()), dev(B.T.copy()), dev(C0.T.copy())
    check(lib.dftk_mi_zgemm(bs.h, trans.encode(), m, n, k, cplx(alpha), Ad.data_ptr(), A.shape[0], Bd.data_ptr(), k,
                            cplx(beta), Cd.data_ptr(), m))
    bs.sync()
    opA = A if trans == "N" else A.conj().T
    ref = alpha * (opA @ B) + beta * C0
    assert relerr(Cd.cpu().numpy().T, ref) < 1e-13
    # beta = 0 must not read C (NaN-filled output buffer)
    Cn = torch.full_like(Cd, float("nan"))
    check(lib.dftk_mi_zgemm(bs.h, trans.encode(), m, n, k, cplx(1.0), Ad.data_ptr(), A.shape[0], Bd.data_ptr(), k,
                            cplx(0.0), Cn.data_ptr(), m))
    bs.sync()
    assert relerr(Cn.cpu().numpy().T, opA @ B) < 1e-13


@pytest.mark.parametrize("m,k", [(259, 3000), (777, 2500), (141, 4100), (64, 300), (5, 2100), (1006, 9000), (1300, 300)])
def test_zgemm_upper_only(lib, m, k):
    """DFTK_MI_GEMM_UPPER: tiles that intersect the upper triangle hold A^H B, the others are untouched;
    split-K (interior and border planned separately) must agree with the unsplit product.  The interior launch runs
    over the live tiles only (compact grid, contiguous XCD blocks): 1006 = 8 row panels with a K split, 1300 = more
    live tiles than resident workgroups, no split."""
    rng = np.random.default_rng(m + k)
    bs = Basis(lib, 8, 8, 8)
    A = rng.standard_normal((k, m)) + 1j * rng.standard_normal((k, m))
    B = rng.standard_normal((k, m)) + 1j * rng.standard_normal((k, m))
    Ad, Bd = dev(A.T.copy()), dev(B.T.copy())
    Cd = torch.full((m, m), 7.0 + 3.0j, dtype=torch.complex128, device="cuda")
    check(lib.dftk_mi_zgemm_ex(bs.h, b"C", m, m, k, cplx(1.0), Ad.data_ptr(), k, Bd.data_ptr(), k, cplx(0.0),
                               Cd.data_ptr(), m, 1))
    bs.sync()
    C = Cd.cpu().numpy().T
    ref = A.conj().T @ B
    iu = np.triu_indices(m)
    assert np.abs(C[iu] - ref[iu]).max() < 1e-12 * np.abs(ref).max()
    # untouched entries keep the fill value, computed ones match the reference: nothing else may appear
    computed = np.abs(C - ref) < 1e-12 * np.abs(ref).max()
    untouched = C == 7.0 + 3.0j
    assert np.all(computed | untouched)
    i, j = np.indices((m, m))
    assert np.all(untouched[(i // 128) * 128 >= (j // 64) * 64 + 64])   # tiles strictly below the diagonal


@pytest.mark.parametrize("m,n", [(5000, 259), (700, 64), (1300, 141), (260, 17)])
def test_zgemm_upper_triangular_B(lib, m, n):
    """DFTK_MI_GEMM_B_UPPER (X * inv(R)): same result as the full product with triu(B), and the strictly
    lower part of B is never read (NaN there)."""
    rng = np.random.default_rng(m + n)
    bs = Basis(lib, 8, 8, 8)
    A = rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))
    R = np.triu(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    Rnan = R.copy()
    Rnan[np.tril_indices(n, -64)] = np.nan     # beyond the 64-column tile granularity nothing below is read
    Ad, Bd = dev(A.T.copy()), dev(Rnan.T.copy())
    Cd = torch.zeros((n, m), dtype=torch.complex128, device="cuda")
    check(lib.dftk_mi_zgemm_ex(bs.h, b"N", m, n, n, cplx(1.0), Ad.data_ptr(), m, Bd.data_ptr(), n, cplx(0.0),
                               Cd.data_ptr(), m, 2))
    bs.sync()
    assert relerr(Cd.cpu().numpy().T, A @ R) < 1e-13
    assert lib.dftk_mi_zgemm_ex(bs.h, b"N", m, n, n, cplx(1.0), Ad.data_ptr(), m, Bd.data_ptr(), n, cplx(0.0),
                                Cd.data_ptr(), m, 16) < 0   # unknown flag


def test_zgemm_random_shapes_against_torch(lib):
    """Seeded sweep over ragged shapes (every planner branch: no split, one-round split, z-major chunk
    placement, interior + L-shaped border, short K): the library product against torch's complex matmul."""
    rng = np.random.default_rng(2024)
    bs = Basis(lib, 8, 8, 8)
    gen = torch.Generator(device="cuda").manual_seed(7)
    shapes = []
    for _ in range(28):
        trans = "C" if rng.random() < 0.6 else "N"
        if trans == "C":
            m, n = int(rng.integers(1, 900)), int(rng.integers(1, 700))
            k = int(rng.choice([rng.integers(1, 130), rng.integers(130, 2100), rng.integers(2100, 60000)]))
        else:
            m, n, k = int(rng.integers(1, 40000)), int(rng.integers(1, 330)), int(rng.integers(1, 900))
        shapes.append((trans, m, n, k))
    shapes += [("C", 128, 64, 4096), ("C", 129, 65, 4097), ("N", 128, 64, 8), ("N", 257, 129, 9), ("C", 1, 1, 1)]
    for trans, m, n, k in shapes:
        ra, ca = (m, k) if trans == "N" else (k, m)
        A = torch.randn((ca, ra), dtype=torch.complex128, device="cuda", generator=gen)      # column-major (ra x ca)
        B = torch.randn((n, k), dtype=torch.complex128, device="cuda", generator=gen)        # column-major (k x n)
        C0 = torch.randn((n, m), dtype=torch.complex128, device="cuda", generator=gen)
        Cd = C0.clone()
        alpha, beta = 0.3 + 0.4j, (0.0 if (m + n + k) % 3 == 0 else -0.5 + 0.25j)
        torch.cuda.synchronize()
        check(lib.dftk_mi_zgemm(bs.h, trans.encode(), m, n, k, cplx(alpha), A.data_ptr(), ra, B.data_ptr(), k,
                                cplx(beta), Cd.data_ptr(), m))
        bs.sync()
        # A holds the column-major (ra x ca) matrix as a (ca, ra) torch tensor, i.e. A.T is the matrix itself
        opA = A.T if trans == "N" else A.conj()          # op(A) as an (m x k) torch matrix
        ref = alpha * (opA @ B.T) + beta * C0.T
        err = (Cd.T - ref).abs().max().item() / max(ref.abs().max().item(), 1e-300)
        assert err < 2e-13, (trans, m, n, k, err)


def test_zgemm_asymmetric_layout(lib):
    """A = I with an asymmetric B catches transposed MFMA output maps."""
    bs = Basis(lib, 8, 8, 8)
    m = 48
    A = np.eye(m, dtype=complex)
    B = (np.arange(m)[:, None] * 100 + np.arange(m)[None, :]).astype(complex) * (1 + 0.5j)
    Ad, Bd = dev(A.T.copy()), dev(B.T.copy())
    Cd = torch.zeros_like(Bd)
    check(lib.dftk_mi_zgemm(bs.h, b"N", m, m, m, cplx(1), Ad.data_ptr(), m, Bd.data_ptr(), m, cplx(0), Cd.data_ptr(), m))
    bs.sync()
    np.testing.assert_allclose(Cd.cpu().numpy().T, B, rtol=0, atol=1e-12)


@pytest.mark.parametrize("n", [1, 7, 31, 32, 33, 47, 48, 49, 64, 100, 259, 500, 503, 511, 512, 513, 600])
def test_potrf_trtri(lib, n):
    """32 < n <= 512 takes the cooperative one-launch factorisation (16-column blocks, dataflow over flags: exact block
    multiples, one column over, one short, the 503 of the headline cell, the last size before the blocked path); the
    other sizes the three-launches-per-panel path.  ``lda`` > n and NaNs in the strict lower triangle check that only
    the upper triangle is read and written."""
    rng = np.random.default_rng(n)
    bs = Basis(lib, 8, 8, 8)
    X = rng.standard_normal((3 * n + 5, n)) + (1j if n % 2 else 0) * rng.standard_normal((3 * n + 5, n))
    O = (X.conj().T @ X).astype(complex)
    if n in (100, 503):      # strict lower triangle must never be read: poison it
        Od = dev((np.triu(O) + np.tril(np.full((n, n), np.nan), -1)).T.copy())
        Id = torch.full_like(Od, float("nan"))
        check(lib.dftk_mi_potrf_trtri(bs.h, n, Od.data_ptr(), n, Id.data_ptr(), n))
        bs.sync()
        R = np.triu(Od.cpu().numpy().T)
        assert relerr(R.conj().T @ R, O) < 1e-13
        assert np.all(np.isnan(np.tril(Od.cpu().numpy().T, -1)[np.tril_indices(n, -1)]))     # untouched
    Od = dev(O.T.copy())
    Id = torch.full_like(Od, float("nan"))
    check(lib.dftk_mi_potrf_trtri(bs.h, n, Od.data_ptr(), n, Id.data_ptr(), n))
    bs.sync()
    R = np.triu(Od.cpu().numpy().T)
    invR = Id.cpu().numpy().T
    assert relerr(R.conj().T @ R, O) < 1e-13
    assert np.linalg.norm(invR @ R - np.eye(n)) < 1e-10 * np.linalg.cond(R)
    assert np.allclose(np.tril(invR, -1), 0)
    # indefinite matrix must be reported as numerical failure (status > 0), not crash
    Bad = O - 2 * np.trace(O).real / n * np.eye(n)
    Bd = dev(Bad.T.copy())
    st = lib.dftk_mi_potrf_trtri(bs.h, n, Bd.data_ptr(), n, Id.data_ptr(), n)
    assert st == 2


@pytest.mark.parametrize("n", [5, 33, 48, 100, 259, 503, 512, 600])
def test_potrf_trtri_real(lib, n):
    """The real-symmetric entry (Gram matrices of the Gamma-real iteration): imaginary parts are never read (poisoned
    here), results equal to the complex entry's on the same matrix."""
    rng = np.random.default_rng(100 + n)
    bs = Basis(lib, 8, 8, 8)
    X = rng.standard_normal((3 * n + 5, n))
    O = (X.T @ X).astype(complex)
    Od = dev(O.T.copy())
    Id = torch.full_like(Od, float("nan"))
    check(lib.dftk_mi_potrf_trtri(bs.h, n, Od.data_ptr(), n, Id.data_ptr(), n))
    bs.sync()
    R_c, I_c = np.triu(Od.cpu().numpy().T), Id.cpu().numpy().T
    Od = dev(O.T.copy())
    Id = torch.full_like(Od, float("nan"))
    check(lib.dftk_mi_potrf_trtri_real(bs.h, n, Od.data_ptr(), n, Id.data_ptr(), n))
    bs.sync()
    R, invR = np.triu(Od.cpu().numpy().T), Id.cpu().numpy().T
    assert np.all(R.imag == 0) and np.all(invR.imag == 0)
    assert relerr(R.T @ R, O) < 1e-13
    assert relerr(R, R_c) < 1e-12 * np.linalg.cond(R_c) and relerr(invR, I_c) < 1e-12 * np.linalg.cond(R_c)
    assert np.allclose(np.tril(invR, -1), 0)
    Bd = dev((O - 2 * np.trace(O).real / n * np.eye(n)).T.copy())
    assert lib.dftk_mi_potrf_trtri_real(bs.h, n, Bd.data_ptr(), n, Id.data_ptr(), n) == 2


@pytest.mark.parametrize("n", [1, 5, 16, 17, 32, 33, 48, 64, 100, 300, 500, 777])
def test_heev(lib, n):
    rng = np.random.default_rng(n)
    bs = Basis(lib, 8, 8, 8)
    X = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    A = (X + X.conj().T) / 2
    if n == 300:   # LOBPCG-like: nearly diagonal with clusters
        A = np.diag(np.sort(rng.standard_normal(n))) + 1e-3 * A
    if n == 48:    # already diagonal (unsorted): no rotation at all
        A = np.diag(rng.standard_normal(n)).astype(complex)
    if n in (64, 500):   # exactly degenerate spectrum: three eigenvalues with large multiplicities
        Q = np.linalg.qr(X)[0]
        A = (Q * rng.choice([-1.0, 0.0, 2.0], n)[None, :]) @ Q.conj().T
        A = (A + A.conj().T) / 2
    Ad = dev(A.T.copy())
    Vd = torch.full_like(Ad, float("nan"))
    W = np.zeros(n)
    check(lib.dftk_mi_heev(bs.h, n, Ad.data_ptr(), n, W.ctypes.data, Vd.data_ptr(), n))
    bs.sync()
    V = Vd.cpu().numpy().T
    wref = np.linalg.eigvalsh(A)
    scale = max(np.abs(wref).max(), 1.0)
    assert np.abs(W - wref).max() < 1e-12 * scale
    assert np.linalg.norm(V.conj().T @ V - np.eye(n)) < max(1e-12, 5e-14 * n)
    assert np.linalg.norm(A @ V - V * W[None, :]) < 1e-11 * scale * np.sqrt(n)


# ----------------------------------------------------------------------------------- FFT pipeline
A_SI = 5.131570667152971
LATTICE = np.array([[0, A_SI, A_SI], [A_SI, 0, A_SI], [A_SI, A_SI, 0.0]])
POSITIONS = [np.ones(3) / 8, -np.ones(3) / 8]


def si_atoms():
    Si = ElementPsp("Si", load_psp_hgh("Si", "lda"))
    return [Si, Si]


def make_oracle_basis(Ecut, fft_size, kcoords=None, terms=("Kinetic", "AtomicLocal", "AtomicNonlocal"), lattice=None):
    lat = LATTICE if lattice is None else lattice
    model = Model(lat, si_atoms(), POSITIONS, terms=terms)
    kg = ExplicitKpoints(kcoords or [[0, 0, 0], [1 / 3, 0.1, -0.25]], None)
    kg.kweights = [1.0 / len(kg.kcoords)] * len(kg.kcoords)
    return PlaneWaveBasis(model, Ecut, kg, fft_size=fft_size)


FFT_CASES = [(5, (15, 15, 15)), (7, (17, 17, 17)), (15, (27, 27, 27)), (10, (21, 21, 21)), (12, (24, 25, 27)),
             (15, (30, 30, 30)), (20, (36, 40, 32)), (6, (33, 16, 20))]


@pytest.mark.parametrize("Ecut,fft_size", FFT_CASES)
def test_sphere_fft_roundtrip_and_oracle(lib, Ecut, fft_size):
    """ifft!/fft! on the sphere (src/fft.jl:110-122,162-172) for 2-3-5-smooth, prime and mixed sizes."""
    obasis = make_oracle_basis(Ecut, fft_size, terms=("Kinetic",))
    nx, ny, nz = fft_size
    bs = Basis(lib, nx, ny, nz, obasis.model.unit_cell_volume)
    rng = np.random.default_rng(sum(fft_size))
    for kpt in obasis.kpoints:
        kb = KBlock(lib, bs, kpt.mapping, np.zeros(len(kpt.mapping)))
        c = rng.standard_normal(len(kpt.mapping)) + 1j * rng.standard_normal(len(kpt.mapping))
        cd_ = dev(c)
        cube = torch.full((nz, ny, nx), float("nan"), dtype=torch.complex128, device="cuda")
        check(lib.dftk_mi_ifft_sphere(kb.h, cd_.data_ptr(), cube.data_ptr()))
        bs.sync()
        ref = obasis.ifft(kpt, c, normalize=False)
        assert relerr(cube.cpu().numpy(), ref) < RTOL
        # forward: random cube -> sphere
        f = rng.standard_normal((nz, ny, nx)) + 1j * rng.standard_normal((nz, ny, nx))
        fd = dev(f)
        out = torch.full_like(cd_, float("nan"))
        check(lib.dftk_mi_fft_sphere(kb.h, fd.data_ptr(), out.data_ptr()))
        bs.sync()
        assert relerr(out.cpu().numpy(), obasis.fft(kpt, f, normalize=False)) < RTOL
        # round trip: FFT(IFFT(c)) / N == c
        back = torch.full_like(cd_, float("nan"))
        check(lib.dftk_mi_fft_sphere(kb.h, cube.data_ptr(), back.data_ptr()))
        bs.sync()
        assert relerr(back.cpu().numpy() / (nx * ny * nz), c) < RTOL


# the z sizes end their radix plans with 3, 6 (30 = 5.6), 8 (24 = 3.8), 6 (36 = 6.6), 4 (20 = 5.4), 5 (25),
# 2 (16 = 8.2) and a generic prime (21 = 3.7): every variant of the fused middle pass of stage C and its
# fall-back
@pytest.mark.parametrize("Ecut,fft_size,nbands", [(10, (21, 21, 21), 5), (15, (27, 27, 27), 11), (12, (24, 25, 27), 3),
                                                   (15, (30, 30, 30), 19), (8, (20, 25, 24), 9), (9, (24, 20, 36), 4),
                                                   (7, (25, 18, 20), 2), (7, (18, 20, 25), 17), (5, (16, 15, 16), 1)])
def test_apply_H_vs_oracle(lib, Ecut, fft_size, nbands):
    """mul!(Hpsi, H, psi) (Hamiltonian.jl:137-192): each part and the total against the oracle."""
    obasis = make_oracle_basis(Ecut, fft_size)
    rng = np.random.default_rng(42)
    nx, ny, nz = fft_size
    V = obasis.terms.V_loc + 0.1 * rng.standard_normal((nz, ny, nx))   # generic real potential
    _, ham = energy_hamiltonian(obasis, None, None)
    bs = Basis(lib, nx, ny, nz, obasis.model.unit_cell_volume)
    for ik, kpt in enumerate(obasis.kpoints):
        H = ham[ik]
        H.potential = V
        kb = KBlock(lib, bs, kpt.mapping, H.kinetic)
        kb.set_projectors(H.P, H.D)
        kb.set_potential(V)
        psi = np.linalg.qr(rng.standard_normal((H.n_G, nbands)) + 1j * rng.standard_normal((H.n_G, nbands)))[0]
        ref_loc = H.apply_local(psi)
        ref_kin = H.kinetic[:, None] * psi
        ref_nl = H.apply_nonlocal(psi)
        assert relerr(kb.apply(psi, 1), ref_loc) < RTOL
        assert relerr(kb.apply(psi, 2), ref_kin) < RTOL
        assert relerr(kb.apply(psi, 4), ref_nl) < RTOL
        assert relerr(kb.apply(psi, 3), ref_loc + ref_kin) < RTOL
        assert relerr(kb.apply(psi, 7), H.mul(psi)) < RTOL
        # Hermiticity (operator consistency, test/hamiltonian_consistency.jl:54-58)
        Hpsi = kb.apply(psi, 7)
        G = psi.conj().T @ Hpsi
        assert np.linalg.norm(G - G.conj().T) < 1e-11 * np.linalg.norm(G)


def test_apply_H_linearity_batches(lib):
    """Batch boundaries: n_bands not a multiple of the FFT batch, ld > n_G, and linearity."""
    obasis = make_oracle_basis(12, (24, 24, 24))
    rng = np.random.default_rng(3)
    _, ham = energy_hamiltonian(obasis, None, None)
    H, kpt = ham[0], obasis.kpoints[0]
    bs = Basis(lib, 24, 24, 24, obasis.model.unit_cell_volume)
    check(lib.dftk_mi_basis_set_fft_batch(bs.h, 4))
    kb = KBlock(lib, bs, kpt.mapping, H.kinetic)
    kb.set_projectors(H.P, H.D)
    kb.set_potential(H.potential)
    n, M, ld = H.n_G, 10, H.n_G + 13
    psi = rng.standard_normal((n, M)) + 1j * rng.standard_normal((n, M))
    buf = torch.zeros((M, ld), dtype=torch.complex128, device="cuda")
    buf[:, :n] = dev(psi.T.copy())
    out = torch.full((M, ld), float("nan"), dtype=torch.complex128, device="cuda")
    check(lib.dftk_mi_apply_H(kb.h, M, buf.data_ptr(), ld, out.data_ptr(), ld))
    bs.sync()
    got = out.cpu().numpy()[:, :n].T
    assert relerr(got, H.mul(psi)) < RTOL
    assert np.isnan(out.cpu().numpy()[:, n:].real).all()      # padding rows untouched
    a, b2 = 0.3 - 1.1j, 2.0 + 0.4j
    lin = kb.apply(a * psi[:, :3] + b2 * psi[:, 3:6])
    assert relerr(lin, a * got[:, :3] + b2 * got[:, 3:6]) < RTOL
    # zero bands: no-op
    check(lib.dftk_mi_apply_H(kb.h, 0, buf.data_ptr(), ld, out.data_ptr(), ld))


@pytest.mark.parametrize("Ecut,fft_size", [(10, (21, 21, 21)), (15, (27, 27, 27)), (12, (24, 25, 27))])
def test_density_vs_oracle(lib, Ecut, fft_size):
    """compute_density inner loop (densities.jl:35-43)."""
    obasis = make_oracle_basis(Ecut, fft_size, terms=("Kinetic",))
    rng = np.random.default_rng(7)
    nx, ny, nz = fft_size
    bs = Basis(lib, nx, ny, nz, obasis.model.unit_cell_volume)
    check(lib.dftk_mi_basis_set_fft_batch(bs.h, 3))
    rho = torch.zeros((nz, ny, nx), dtype=torch.float64, device="cuda")
    psis, occs = [], []
    for ik, kpt in enumerate(obasis.kpoints):
        n = len(kpt.mapping)
        M = 7
        psi = np.linalg.qr(rng.standard_normal((n, M)) + 1j * rng.standard_normal((n, M)))[0]
        occ = np.array([2.0, 2.0, 1.3, 0.0, 0.5, 0.0, 2.0])
        psis.append(psi)
        occs.append(occ)
        kb = KBlock(lib, bs, kpt.mapping, np.zeros(n))
        w = occ * obasis.kweights[ik] * obasis.ifft_normalization ** 2
        pd = dev(psi.T.copy())
        check(lib.dftk_mi_density_accumulate(kb.h, M, pd.data_ptr(), n, w.ctypes.data, rho.data_ptr()))
        bs.sync()
    ref = compute_density(obasis, psis, occs)
    assert relerr(rho.cpu().numpy(), ref) < RTOL
    # the density integrates to the electron count: sum_k w_k sum_n f_n
    nel = sum(obasis.kweights[ik] * occs[ik].sum() for ik in range(len(psis)))
    assert abs(rho.sum().item() * obasis.dvol - nel) < 1e-10


# ----------------------------------------------------------------------------------- LOBPCG
def run_lobpcg(lib, kb, X0, tol, n_conv_check=0, use_tpa=1, maxiter=100):
    n, M = X0.shape
    Xd = dev(X0.T.copy())
    lam = np.zeros(M)
    res = np.zeros(M)
    n_iter, conv, nmv = C.c_int(), C.c_int(), C.c_int64()
    check(lib.dftk_mi_lobpcg(kb.h, M, Xd.data_ptr(), n, tol, 1, maxiter, n_conv_check, use_tpa, 1234,
                             lam.ctypes.data, res.ctypes.data, C.byref(n_iter), C.byref(conv), C.byref(nmv)))
    return lam, res, n_iter.value, conv.value, nmv.value, Xd.cpu().numpy().T


def test_lobpcg_free_electron_golden(lib):
    """test/lobpcg.jl:13-50: free-electron eigenvalues (reference golden values)."""
    ref = [
        [0.00000000000, 0.56219939834, 0.56219939834, 0.56219939834, 0.56219939834,
         0.56219939834, 0.56219939834, 0.56219939834, 0.56219939834, 0.74959919778],
        [0.06246659981, 0.24986639926, 0.49973279852, 0.49973279852, 0.49973279852,
         0.56219939834, 0.56219939834, 0.56219939834, 0.74959919778, 0.74959919778],
        [0.08328879975, 0.33315519901, 0.39562179883, 0.39562179883, 0.39562179883,
         0.39562179883, 0.83288799753, 0.83288799754, 0.83288799754, 0.83288799754],
        [0.16657759951, 0.22904419932, 0.22904419932, 0.41644399877, 0.41644399877,
         0.66631039803, 0.72877699784, 0.72877699784, 0.72877699784, 0.72877699784],
    ]
    kc = [[0, 0, 0], [1 / 3, 0, 0], [1 / 3, 1 / 3, 0], [-1 / 3, 1 / 3, 0]]
    obasis = make_oracle_basis(5, (15, 15, 15), kcoords=kc, terms=("Kinetic",))
    _, ham = energy_hamiltonian(obasis, None, None)
    bs = Basis(lib, 15, 15, 15, obasis.model.unit_cell_volume)
    rng = np.random.default_rng(0)
    for ik, kpt in enumerate(obasis.kpoints):
        kb = KBlock(lib, bs, kpt.mapping, ham[ik].kinetic)
        n = len(kpt.mapping)
        X0 = np.linalg.qr(rng.standard_normal((n, 10)) + 1j * rng.standard_normal((n, 10)))[0]
        lam, res, nit, conv, nmv, X = run_lobpcg(lib, kb, X0, 1e-8)
        assert conv == 1 and nit < 50
        np.testing.assert_allclose(lam, ref[ik], atol=1e-9)
        assert res.max() < 100 * 1e-8
        assert np.linalg.norm(X.conj().T @ X - np.eye(10)) < 1e-10


def test_lobpcg_core_hamiltonian_vs_oracle_and_dense(lib):
    """test/lobpcg.jl:78-122: kinetic + local + nonlocal; device LOBPCG == oracle LOBPCG == dense."""
    kc = [[0, 0, 0], [1 / 3, 0, 0], [1 / 3, 1 / 3, 0], [-1 / 3, 1 / 3, 0]]
    obasis = make_oracle_basis(10, (21, 21, 21), kcoords=kc)
    _, ham = energy_hamiltonian(obasis, None, None)
    ores = diagonalize_all_kblocks(ham, 5, tol=1e-9, interpolate_kpoints=False)
    bs = Basis(lib, 21, 21, 21, obasis.model.unit_cell_volume)
    rng = np.random.default_rng(1)
    for ik, kpt in enumerate(obasis.kpoints):
        H = ham[ik]
        kb = KBlock(lib, bs, kpt.mapping, H.kinetic)
        kb.set_projectors(H.P, H.D)
        kb.set_potential(H.potential)
        X0 = np.linalg.qr(rng.standard_normal((H.n_G, 5)) + 1j * rng.standard_normal((H.n_G, 5)))[0]
        lam, res, nit, conv, nmv, X = run_lobpcg(lib, kb, X0, 1e-9)
        assert conv == 1
        dense = np.linalg.eigvalsh(H.to_dense())[:5]
        np.testing.assert_allclose(lam, dense, atol=1e-9)
        np.testing.assert_allclose(lam, ores["λ"][ik], atol=1e-9)
        # residuals reported are true residuals
        R = H.mul(X) - X * lam[None, :]
        assert np.linalg.norm(R, axis=0).max() < 1e-8
        assert nmv >= 5 * (nit + 1) - 5 * nit   # at least one block apply
        if ik == 0:   # no preconditioner (prec = I, lobpcg_hyper_impl.jl:354): same eigenvalues, more iterations
            lam0, _, nit0, conv0, _, _ = run_lobpcg(lib, kb, X0, 1e-8, use_tpa=0, maxiter=400)
            assert conv0 == 1 and nit0 >= nit
            np.testing.assert_allclose(lam0, dense, atol=1e-8)


def test_lobpcg_n_conv_check_and_locking(lib):
    """Extra (unconverged) bands, soft locking and n_matvec accounting (lobpcg_hyper_impl.jl:461-484)."""
    obasis = make_oracle_basis(15, (27, 27, 27), kcoords=[[0, 0.25, -1 / 3]])
    rho_terms = energy_hamiltonian(obasis, None, None)[1]
    H, kpt = rho_terms[0], obasis.kpoints[0]
    bs = Basis(lib, 27, 27, 27, obasis.model.unit_cell_volume)
    kb = KBlock(lib, bs, kpt.mapping, H.kinetic)
    kb.set_projectors(H.P, H.D)
    kb.set_potential(H.potential)
    rng = np.random.default_rng(5)
    M, ncc = 12, 8
    X0 = rng.standard_normal((H.n_G, M)) + 1j * rng.standard_normal((H.n_G, M))   # not orthonormal
    lam, res, nit, conv, nmv, X = run_lobpcg(lib, kb, X0, 1e-7, n_conv_check=ncc)
    dense = np.linalg.eigvalsh(H.to_dense())[:M]
    assert conv == 1
    np.testing.assert_allclose(lam[:ncc], dense[:ncc], atol=1e-9)
    assert np.all(np.diff(lam) >= -1e-12)
    assert np.linalg.norm(X.conj().T @ X - np.eye(M)) < 1e-10
    assert M <= nmv <= M * (nit + 1)
    # too-small problem is reported like the reference (N > 3M)
    lam2 = np.zeros(400)
    Xbig = dev(np.zeros((400, H.n_G), dtype=complex))
    st = lib.dftk_mi_lobpcg(kb.h, 400, Xbig.data_ptr(), H.n_G, 1e-6, 1, 10, 0, 1, 1, lam2.ctypes.data,
                            lam2.ctypes.data, C.byref(C.c_int()), C.byref(C.c_int()), C.byref(C.c_int64()))
    assert st == 5
    # a non-finite potential is reported as the reference's "NaN in AX" assertion (lobpcg_hyper_impl.jl:380)
    Vbad = H.potential.copy()
    Vbad[3, 4, 5] = np.nan
    kb.set_potential(Vbad)
    Xd = dev(np.linalg.qr(X0)[0].T.copy())
    st = lib.dftk_mi_lobpcg(kb.h, M, Xd.data_ptr(), H.n_G, 1e-6, 1, 10, 0, 1, 1, lam2.ctypes.data, lam2.ctypes.data,
                            C.byref(C.c_int()), C.byref(C.c_int()), C.byref(C.c_int64()))
    assert st == 1 and b"non-finite" in lib.dftk_mi_last_error()
