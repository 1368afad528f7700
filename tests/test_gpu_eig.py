"""dftk_mi_heev_lowest (csrc/eig_kernels.hip): the lowest nev eigenpairs by one spectral split (Newton-Schulz sign
iteration on the matrix cores) + the blocked Jacobi on the projected matrix -- what rayleigh_ritz consumes
(src/eigen/lobpcg_hyper_impl.jl:141-153).  Checked against LAPACK (numpy.linalg.eigh) on matrices with the structure of
LOBPCG Rayleigh-Ritz matrices (leading block = diag of Ritz values, clustered low end), on matrices where the shift rule's
interlacing argument does not hold (the count check must send them to the full solver) and on a matrix with an
eigenvalue almost exactly at the shift (more held iterations / the full solver, never a wrong answer)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd._lib import check  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return dftk.load_library()


@pytest.fixture(scope="module")
def basis(lib):
    h = C.c_void_p()
    check(lib.dftk_mi_basis_create(8, 8, 8, 1.0, 0, C.byref(h)))
    yield h
    lib.dftk_mi_basis_destroy(h)


def rr_like(n, nev, rng, cplx=False, coupling=0.1, top=8.0):
    """Hermitian matrix with the structure of Y'AY inside LOBPCG: spectrum with a dense, clustered low end (nev + nev/3
    values in [-0.2, 0.6], four-fold near-degenerate) and a spread upper part; the leading nev x nev block is diagonal
    (Ritz values of the previous iteration) and captures the low end up to `coupling`."""
    nlow = nev + nev // 3
    low = np.repeat(rng.uniform(-0.2, 0.6, nlow // 4 + 1), 4)[:nlow] + 1e-6 * rng.standard_normal(nlow)
    lam = np.concatenate([np.sort(low), rng.uniform(0.6, top, n - nlow)])
    W = np.eye(n) + coupling * rng.standard_normal((n, n)) / np.sqrt(n)
    if cplx:
        W = W + 1j * coupling * rng.standard_normal((n, n)) / np.sqrt(n)
    Q, _ = np.linalg.qr(W)
    A = (Q.conj().T * lam) @ Q
    A = (A + A.conj().T) / 2
    w, Z = np.linalg.eigh(A[:nev, :nev])
    T = np.eye(n, dtype=A.dtype)
    T[:nev, :nev] = Z
    A = T.conj().T @ A @ T
    A = (A + A.conj().T) / 2
    A[:nev, :nev] = np.diag(w)          # exactly diagonal, as X'AX = diag(lambda) up to round-off
    return A


def run_lowest(lib, basis, A, nev):
    n = A.shape[0]
    Ad = torch.tensor(np.ascontiguousarray(A.T), dtype=torch.complex128, device="cuda")
    Vd = torch.zeros((n, n), dtype=torch.complex128, device="cuda")
    W = np.zeros(n)
    torch.cuda.synchronize()
    check(lib.dftk_mi_heev_lowest(basis, n, nev, Ad.data_ptr(), n, W.ctypes.data, Vd.data_ptr(), n))
    torch.cuda.synchronize()
    V = Vd.cpu().numpy().T[:, :nev]       # column c of the column-major device array = row c of the torch tensor
    return W[:nev], V


def check_pairs(A, lam, V, nev, scale=None):
    ref = np.linalg.eigvalsh(A)
    scale = scale or max(abs(ref[0]), abs(ref[-1]))
    assert np.abs(lam - ref[:nev]).max() < 1e-11 * max(1.0, scale)
    assert np.all(np.diff(lam) >= -1e-13)
    assert np.abs(V.conj().T @ V - np.eye(nev)).max() < 1e-12
    assert np.abs(A @ V - V * lam).max() < 2e-11 * max(1.0, scale)


@pytest.mark.parametrize("n,nev,cplx,coupling", [(700, 233, False, 0.3), (1006, 503, False, 1e-3), (1006, 503, False, 0.5),
                                                 (1509, 503, False, 0.5), (777, 259, False, 0.2), (640, 213, True, 0.3),
                                                 (1006, 503, True, 0.05)])
def test_heev_lowest_on_rayleigh_ritz_like_matrices(lib, basis, n, nev, cplx, coupling):
    rng = np.random.default_rng(n + nev + int(cplx))
    A = rr_like(n, nev, rng, cplx, coupling)
    lam, V = run_lowest(lib, basis, A, nev)
    check_pairs(A, lam, V, nev)
    if not cplx:
        assert np.abs(V.imag).max() == 0.0      # real symmetric input: exact zeros in the imaginary parts


def test_heev_lowest_without_the_interlacing_structure(lib, basis):
    """A dense random symmetric matrix: the nev smallest diagonal entries say nothing about the spectrum (fewer than nev
    eigenvalues may lie below the shift, or far too many): the count check hands over to the full solver."""
    rng = np.random.default_rng(5)
    n, nev = 768, 256
    A = rng.standard_normal((n, n))
    A = (A + A.T) / 2 + np.diag(np.linspace(-1, 30, n))
    lam, V = run_lowest(lib, basis, A, nev)
    check_pairs(A, lam, V, nev)


def test_heev_lowest_eigenvalue_at_the_shift(lib, basis):
    """An eigenvalue of the trailing block 1e-9 (relative to the norm) above the shift the rule picks: the sign iteration
    needs more held iterations than its estimate (or gives up): the pairs must be right either way."""
    rng = np.random.default_rng(9)
    n, nev = 800, 260
    lam_x = np.sort(rng.uniform(-0.2, 0.5, nev))
    beta = rng.uniform(2.0, 9.0, n - nev)
    Q, _ = np.linalg.qr(rng.standard_normal((n - nev, n - nev)))
    sigma, gap = C.c_double(), C.c_double()
    for _ in range(4):            # the shift depends (weakly) on the diagonal of the trailing block: fixed point
        B = (Q * beta) @ Q.T
        d = np.concatenate([lam_x, np.diag(B)])
        check(lib.dftk_mi_heev_sigma_host(n, d.ctypes.data, nev, C.byref(sigma), C.byref(gap), None, 0.0))
        beta[0] = sigma.value + 1e-8
    B = (Q * beta) @ Q.T
    A = np.zeros((n, n))
    A[:nev, :nev] = np.diag(lam_x)
    A[nev:, nev:] = (B + B.T) / 2
    E = 1e-3 * rng.standard_normal((n - nev, nev))
    A[nev:, :nev] = E
    A[:nev, nev:] = E.T
    lam, V = run_lowest(lib, basis, A, nev)
    check_pairs(A, lam, V, nev)


def test_heev_lowest_small_problem_takes_the_full_solver(lib, basis):
    rng = np.random.default_rng(3)
    A = rr_like(96, 32, rng, False, 0.2)
    lam, V = run_lowest(lib, basis, A, 32)
    check_pairs(A, lam, V, 32)
