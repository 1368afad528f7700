"""Eigensolver seam: ``lobpcg_hyper`` and ``diagonalize_all_kblocks``.

Contract of the reference (src/eigen/diag.jl:1-8,50-64, src/eigen/diag_lobpcg_hyper.jl:5-18):
``eigensolver(A::HamiltonianBlock, X0; prec, tol, miniter, maxiter, n_conv_check) ->
(; lambda, X, residual_norms, n_iter, converged, n_matvec)``, lambda ascending on the host,
X orthonormal on the device.  The whole LOBPCG iteration (src/eigen/lobpcg_hyper_impl.jl:354-582)
runs inside the library (``dftk_mi_lobpcg``).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .hamiltonian import DftHamiltonianBlock

EPS = float(np.finfo(np.float64).eps)


class PreconditionerTPA:
    """Teter-Payne-Allan preconditioner (src/eigen/preconditioners.jl:27-78).  The kinetic vector lives in the
    k-block of the library, ``mean_kin`` (set by ``precondprep_``) on the host; inside ``lobpcg_hyper`` the library
    applies the same two kernels fused into its residual pass."""

    def __init__(self, ham_block: DftHamiltonianBlock | None = None, default_shift: float = 1.0):
        self.ham_block = ham_block
        self.default_shift = default_shift
        self.mean_kin = None

    def precondprep_(self, X: torch.Tensor):
        """``precondprep!(P, X)`` (:75-77): mean kinetic energy of every band of X (bands = rows)."""
        H = self.ham_block
        mk = np.zeros(X.shape[0])
        H.basis.pre_call()
        _lib.check(H.basis.lib.dftk_mi_tpa_precondprep(H.kpoint.handle, X.shape[0], X.data_ptr(), X.stride(0),
                                                       mk.ctypes.data))
        self.mean_kin = mk
        return self

    def ldiv_(self, Y: torch.Tensor, R: torch.Tensor) -> torch.Tensor:
        """``ldiv!(Y, P, R)`` (:50-60): Y = mean_kin / (mean_kin + kin) .* R, or R ./ (kin + default_shift) before
        the first ``precondprep_``."""
        H = self.ham_block
        H.basis.pre_call()
        mk = self.mean_kin.ctypes.data if self.mean_kin is not None else None
        _lib.check(H.basis.lib.dftk_mi_tpa_ldiv(H.kpoint.handle, R.shape[0], R.data_ptr(), R.stride(0), mk,
                                                float(self.default_shift), Y.data_ptr(), Y.stride(0)))
        H.basis.post_call(H.kpoint.lane)
        return Y


def columnwise_norms(basis, X: torch.Tensor) -> np.ndarray:
    """``columnwise_norms(X)`` (src/common/linalg.jl:2-4) of a band-major block through the library."""
    out = np.zeros(X.shape[0])
    basis.pre_call()
    _lib.check(basis.lib.dftk_mi_columnwise_norms(basis.handle, X.shape[1], X.shape[0], X.data_ptr(), X.stride(0),
                                                  out.ctypes.data))
    return out


def columnwise_dots(basis, A: torch.Tensor, B: torch.Tensor) -> np.ndarray:
    """``columnwise_dots(A, B)`` (src/common/linalg.jl:7-9, src/gpu/linalg.jl:17-19): dot(A[:, i], B[:, i])."""
    out = (_lib.dftk_mi_cplx * A.shape[0])()
    basis.pre_call()
    _lib.check(basis.lib.dftk_mi_columnwise_dots(basis.handle, A.shape[1], A.shape[0], A.data_ptr(), A.stride(0),
                                                 B.data_ptr(), B.stride(0), out))
    return np.array([complex(d.re, d.im) for d in out])


def ortho_qr(basis, X: torch.Tensor) -> torch.Tensor:
    """``ortho_qr(X)`` (src/common/ortho.jl:1-9): orthonormal columns spanning those of X (Cholesky-QR with the
    reference's safeguards instead of Householder: Q differs from LAPACK's by a unitary diagonal)."""
    Q = X.clone().contiguous()
    basis.pre_call()
    _lib.check(basis.lib.dftk_mi_ortho_qr(basis.handle, Q.shape[1], Q.shape[0], Q.data_ptr(), Q.stride(0), 0,
                                          C.byref(C.c_int()), C.byref(C.c_int())))
    return Q


@dataclass
class EigResult:
    λ: np.ndarray
    X: torch.Tensor
    residual_norms: np.ndarray
    n_iter: int
    converged: bool
    n_matvec: int
    real_symmetric: bool = False     # X satisfies psi(-G) = conj(psi(G)) (Gamma-real iteration of the library)


def lobpcg_hyper(A: DftHamiltonianBlock, X0: torch.Tensor, maxiter: int = 100, prec=None, tol: float | None = None,
                 n_conv_check: int | None = None, miniter: int = 1, seed: int = 0, reuse_AX: bool = False) -> EigResult:
    """``lobpcg_hyper(A, X0; prec, tol, maxiter, miniter, n_conv_check)``.  X0: (M, n_G) complex128
    CUDA tensor (rows = bands); it is not modified."""
    basis = A.basis
    if not (X0.is_cuda and X0.dtype == torch.complex128):
        raise TypeError("lobpcg_hyper: complex128 CUDA block required (no CPU fallback)")
    M, n_loc = X0.shape
    if n_loc != A.n_loc:
        raise ValueError(f"Mismatch in dimension between guess ({n_loc}) and Hamiltonian ({A.n_loc})")
    if tol is None:
        tol = 20 * A.n_G * EPS                  # diag_lobpcg_hyper.jl:6
    A.bind()
    X = X0.clone().contiguous()
    lam = np.zeros(M)
    res = np.zeros(M)
    n_iter, conv, nmv = C.c_int(), C.c_int(), C.c_int64()
    basis.pre_call()
    # reuse_AX: X0 IS the block the last call on this k-point returned (the orbitals of the previous SCF step): the library
    # then starts from A_new X = A_old X inv(R) + (V_new - V_old) X instead of a full H X (dftk_mi_kblock_reuse_AX)
    _lib.check(basis.lib.dftk_mi_kblock_reuse_AX(A.kpoint.handle, 1 if reuse_AX else 0))
    _lib.check(basis.lib.dftk_mi_lobpcg(A.kpoint.handle, M, X.data_ptr(), X.stride(0), float(tol), int(miniter),
                                        int(maxiter), int(n_conv_check or 0), 1 if prec is not None else 0,
                                        int(seed) & (2 ** 64 - 1), lam.ctypes.data, res.ctypes.data,
                                        C.byref(n_iter), C.byref(conv), C.byref(nmv)))
    return EigResult(lam, X, res, n_iter.value, bool(conv.value), int(nmv.value),
                     real_symmetric=bool(getattr(A.kpoint, "gamma_real", False)))


def lobpcg_hyper_multi(As, X0s, maxiter: int = 100, prec=True, tol: float | None = None, n_conv_check: int | None = None,
                       miniter: int = 1, seeds=None):
    """``[lobpcg_hyper(A, X0; ...) for (A, X0) in zip(As, X0s)]`` -- the k loop of ``diagonalize_all_kblocks``
    (diag.jl:24-48) -- as ONE library call (``dftk_mi_lobpcg_multi``): the k-blocks iterate in lock-step, their device
    operations merged into batched launches.  All blocks need the same number of bands and one basis handle (lane)."""
    if not As:
        return []
    basis = As[0].basis
    n, M = len(As), X0s[0].shape[0]
    for A, X0 in zip(As, X0s):
        if not (X0.is_cuda and X0.dtype == torch.complex128):
            raise TypeError("lobpcg_hyper_multi: complex128 CUDA blocks required (no CPU fallback)")
        if X0.shape != (M, A.n_loc):
            raise ValueError(f"Mismatch in dimension between guess {tuple(X0.shape)} and Hamiltonian ({M}, {A.n_loc})")
        A.bind()
    if tol is None:
        tol = 20 * max(A.n_G for A in As) * EPS
    # the solver works on copies (the guesses are the previous step's orbitals, which the caller may still hold): ONE
    # multi-tensor copy instead of a launch per k-point
    Xs = [torch.empty(X0.shape, dtype=X0.dtype, device=X0.device) for X0 in X0s]
    torch._foreach_copy_(Xs, list(X0s))
    kbs = (C.c_void_p * n)(*[A.kpoint.handle.value for A in As])
    Xp = (C.c_void_p * n)(*[X.data_ptr() for X in Xs])
    ld = (C.c_int64 * n)(*[X.stride(0) for X in Xs])
    sd = (C.c_uint64 * n)(*[(int(s_) & (2 ** 64 - 1)) for s_ in (seeds if seeds is not None else range(n))])
    lam, res = np.zeros((n, M)), np.zeros((n, M))
    nit, conv, status = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    nmv = np.zeros(n, dtype=np.int64)
    basis.pre_call()
    _lib.check(basis.lib.dftk_mi_lobpcg_multi(n, kbs, M, Xp, ld, float(tol), int(miniter), int(maxiter),
                                              int(n_conv_check or 0), 1 if prec else 0, sd, lam.ctypes.data,
                                              res.ctypes.data, nit.ctypes.data, conv.ctypes.data, nmv.ctypes.data,
                                              status.ctypes.data))
    for i, st in enumerate(status):
        if st != 0:
            raise _lib.DftkMiError(int(st), f"k-block {i} of a batched LOBPCG call: "
                                            + basis.lib.dftk_mi_last_error().decode(errors="replace"))
    return [EigResult(lam[i].copy(), Xs[i], res[i].copy(), int(nit[i]), bool(conv[i]), int(nmv[i])) for i in range(n)]


def batch_stats(basis):
    """(rounds, recorded operations, merged launches, one-by-one operations) of this thread's last batched call."""
    r, o, m_, q = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    _lib.check(basis.lib.dftk_mi_batch_stats(C.byref(r), C.byref(o), C.byref(m_), C.byref(q)))
    return dict(rounds=r.value, ops=o.value, merged_launches=m_.value, sequential_ops=q.value)


def lobpcg_residual_history(A: DftHamiltonianBlock):
    """``resid_history`` of the last ``lobpcg_hyper`` call on this block (lobpcg_hyper_impl.jl:368,443-446):
    array (M, n_iter + 1), rows ordered like the returned eigenpairs; second value = number of SVD fallbacks."""
    lib = A.basis.lib
    M, nit, nsvd = C.c_int(), C.c_int(), C.c_int()
    _lib.check(lib.dftk_mi_lobpcg_history(A.kpoint.handle, C.byref(M), C.byref(nit), None, 0, C.byref(nsvd)))
    hist = np.zeros((nit.value + 1, M.value))
    _lib.check(lib.dftk_mi_lobpcg_history(A.kpoint.handle, C.byref(M), C.byref(nit), hist.ctypes.data, hist.size,
                                          C.byref(nsvd)))
    return hist.T.copy(), nsvd.value


def _splitmix64(x: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser on int64 tensors (wrapping arithmetic; logical shifts emulated with masks)."""
    def lsr(v, k):
        return (v >> k) & ((1 << (64 - k)) - 1)
    x = x + (-7046029254386353131)                      # 0x9E3779B97F4A7C15 as a signed 64-bit integer
    x = (x ^ lsr(x, 30)) * (-4658895280553007687)       # 0xBF58476D1CE4E5B9
    x = (x ^ lsr(x, 27)) * (-7723592293110705685)       # 0x94D049BB133111EB
    return x ^ lsr(x, 31)


def _counter_normal(n, seed: int, device) -> torch.Tensor:
    """Standard normal numbers that depend only on (seed, index): Box-Muller on two splitmix64 streams; ``n`` is
    a count (indices 0 .. n-1) or an int64 tensor of indices.  torch.randn on the GPU assigns Philox subsequences
    per launched thread, so its output depends on the number of CUs of the device; this does not (same start
    vectors on every box, and a rank of a plane-wave-sharded block can draw exactly its rows)."""
    idx = n if torch.is_tensor(n) else torch.arange(n, dtype=torch.int64, device=device)
    base = _splitmix64(torch.tensor([seed & (2 ** 63 - 1)], dtype=torch.int64, device=device))
    a = _splitmix64(idx * 2 + base)
    b = _splitmix64(idx * 2 + 1 + base)
    u1 = (((a >> 11) & ((1 << 53) - 1)).to(torch.float64) + 1.0) * 2.0 ** -53        # (0, 1]
    u2 = ((b >> 11) & ((1 << 53) - 1)).to(torch.float64) * 2.0 ** -53                # [0, 1)
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * np.pi * u2)


def random_orbitals(basis, kpt, howmany: int, generator: torch.Generator | None = None) -> torch.Tensor:
    """orbitals.jl:82-86: complex normal entries; orthonormalisation is left to LOBPCG's first
    Cholesky-QR (``X = ortho!(copy(X))``, lobpcg_hyper_impl.jl:370), which spans the same space."""
    if generator is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    else:   # one draw advances the generator; a single-element draw does not depend on the launch geometry
        seed = int(torch.randint(0, 2 ** 62, (1,), device=generator.device, generator=generator).item())
    if basis.comm_pw.size > 1:
        # the ranks of comm_pw hold row slabs of ONE random block: same seed everywhere (rank 0's draw)
        seed = int(basis.comm_pw.gather_lists(seed)[0])
    # element (c, band, g) of the (2, howmany, n_G) block has index (c * howmany + band) * n_G + g
    rows = torch.arange(kpt.row0, kpt.row1, dtype=torch.int64, device=basis.device)
    lead = torch.arange(2 * howmany, dtype=torch.int64, device=basis.device) * kpt.n_G
    z = _counter_normal((lead[:, None] + rows[None, :]).reshape(-1), seed, basis.device).reshape(2, howmany, kpt.n_loc)
    return torch.complex(z[0], z[1]) / np.sqrt(2 * kpt.n_G)


def interpolate_kpoint(data_in: torch.Tensor, kpoint_in, kpoint_out) -> torch.Tensor:
    """``interpolate_kpoint`` (src/interpolation.jl:96-115): carry the coefficients of the plane waves both spheres
    share from one k-point to another (zero elsewhere) -- a fast, inexact guess for the iterative solver.  The
    re-orthonormalisation (``ortho_qr`` there) is LOBPCG's own first step ``X = ortho!(copy(X))``."""
    if kpoint_in is kpoint_out:
        return data_in.clone()
    m_in, m_out = kpoint_in.mapping_device, kpoint_out.mapping_device
    pos = torch.searchsorted(m_in, m_out).clamp_(max=m_in.numel() - 1)
    hit = m_in[pos] == m_out
    out = torch.zeros((data_in.shape[0], m_out.numel()), dtype=data_in.dtype, device=data_in.device)
    out[:, hit] = data_in[:, pos[hit]]
    return out


COARSE_TOL_FACTOR = 0.25      # tolerance of the companion-basis solve of the two-level start, relative to the caller's


def lowpass_to_coarse(basis, coarse, f: torch.Tensor) -> torch.Tensor:
    """A real cube of ``basis`` on the (smaller) cube of ``coarse`` by Fourier truncation: the coefficients of the plane waves
    the coarse cube represents (|g_i| < n_i / 2; the unpaired Nyquist planes are dropped) through the library's cube FFTs."""
    fG = basis.fft(f)
    nx, ny, nz = basis.fft_size
    cx, cy, cz = coarse.fft_size
    dev = f.device

    def axis(nc, nf):
        g = torch.arange(nc, device=dev)
        g = torch.where(g < (nc + 1) // 2, g, g - nc)               # frequency of coarse index
        keep = (2 * g.abs() < nc) if nc % 2 == 0 else torch.ones_like(g, dtype=torch.bool)
        return g % nf, keep
    ix, kx = axis(cx, nx)
    iy, ky = axis(cy, ny)
    iz, kz = axis(cz, nz)
    out = fG[iz[:, None, None], iy[None, :, None], ix[None, None, :]]
    out = out * (kz[:, None, None] & ky[None, :, None] & kx[None, None, :])
    return coarse.irfft(out.contiguous())


def zero_pad_to_fine(X_coarse: torch.Tensor, kpt_c, kpt_f, basis_f) -> torch.Tensor:
    """Coefficients on the coarse sphere -> the fine sphere of the same k-point (zero elsewhere), matched by integer G."""
    nx, ny, nz = basis_f.fft_size
    inv = getattr(kpt_f, "_inv_mapping", None)
    if inv is None:
        inv = torch.full((nx * ny * nz,), -1, dtype=torch.int32, device=X_coarse.device)
        inv[kpt_f.mapping_device] = torch.arange(kpt_f.n_G, dtype=torch.int32, device=X_coarse.device)
        kpt_f._inv_mapping = inv
    G = kpt_c.G_vectors
    lin = (G[:, 0] % nx) + nx * ((G[:, 1] % ny) + ny * (G[:, 2] % nz))
    pos = inv[lin].to(torch.int64)
    if bool((pos < 0).any()):
        raise RuntimeError("two-level start: the coarse sphere is not contained in the fine one")
    pw = basis_f.comm_pw
    if pw.size > 1:
        # plane-wave sharded block: every rank needs the coarse rows that land in ITS slab of the fine sphere -- the slabs of
        # the (8x smaller) coarse block are summed into the whole block on every rank (one all-reduce per SCF), then cut
        full = torch.zeros((X_coarse.shape[0], kpt_c.n_G), dtype=X_coarse.dtype, device=X_coarse.device)
        full[:, kpt_c.row0:kpt_c.row1] = X_coarse
        basis_f.pre_call()
        pw.sum_(torch.view_as_real(full).reshape(-1), basis_f.stream_ptr)
        basis_f.post_call()
        mine = (pos >= kpt_f.row0) & (pos < kpt_f.row1)
        out = torch.zeros((X_coarse.shape[0], kpt_f.n_loc), dtype=X_coarse.dtype, device=X_coarse.device)
        out[:, pos[mine] - kpt_f.row0] = full[:, mine]
        return out
    out = torch.zeros((X_coarse.shape[0], kpt_f.n_G), dtype=X_coarse.dtype, device=X_coarse.device)
    out[:, pos] = X_coarse
    return out


def _coarse_start_vectors(eigensolver, Hk, ik, nev, prec_type, tol, miniter, maxiter, n_conv_check, generator, seed):
    """Start vectors of the two-level start for one k-point: LOBPCG from random orbitals on the companion basis (the SAME
    potential, Fourier-truncated), zero-padded into the fine sphere.  Returns (X, n_matvec on the coarse basis)."""
    basis = Hk.basis
    coarse = basis.coarse
    kc = coarse.kpoints[ik]
    Vc = lowpass_to_coarse(basis, coarse, Hk.potential)
    Hc = DftHamiltonianBlock(coarse, kc, Vc)
    g = random_orbitals(coarse, kc, nev, generator)
    # (the companion basis is solved to a QUARTER of the caller's tolerance: its iterations cost 1/8 of a fine one, and the
    #  better start saves a fine iteration -- cfg 5, 20-step window 5.62 -> 6.0 it/s, whole SCF 6.10 -> 6.44, cfg 2 19.6 -> 20.3)
    rc = eigensolver(Hc, g, prec=prec_type(Hc) if prec_type is not None else None, tol=COARSE_TOL_FACTOR * tol, miniter=miniter,
                     maxiter=maxiter, n_conv_check=n_conv_check, seed=seed + ik)
    return zero_pad_to_fine(rc.X, kc, Hk.kpoint, basis), int(rc.n_matvec)


def first_wave_indices(n_k: int, width: int, spread: bool):
    """The k-points of a mesh that start from random orbitals when no guess is given: ``width`` of them SPREAD evenly over the
    list (two-wave start of a batched mesh: every other k-point then has a solved neighbour), else the first ``width``."""
    if spread and n_k > width:
        return sorted({int(i * n_k / width) for i in range(width)})
    return list(range(min(width, n_k)))


def nearest_index(coords, target):
    """Index of the row of ``coords`` (fractional k-coordinates) nearest to ``target`` up to reciprocal lattice vectors."""
    d = np.asarray(coords, dtype=float) - np.asarray(target, dtype=float)[None, :]
    d -= np.round(d)
    return int(np.argmin((d * d).sum(axis=1)))


def _chain_width(basis) -> int:
    """How many k-points start from random orbitals when no guess is given; k-point ik > width interpolates the solution
    of an already solved k-point (the reference: the previous one, diag.jl:39-42; here the k-point of the same lane, or -- when
    the k loop is one batched library call -- the nearest k-point of a small first wave, see diagonalize_all_kblocks)."""
    if getattr(basis, "kbatch", False) and basis.n_lanes == 1:
        # at most two waves, at least 16 k-points wide: a wave costs (iterations of its slowest k-point) x (one lock-step
        # round), and a round is latency, not throughput, up to dozens of k-points -- Al 72 k-points, first SCF step:
        # 57 / 46 / 42 / 39 / 45 ms at widths 8 / 16 / 24 / 36 / 72 (more random starts cost iterations, fewer waves save rounds)
        n_k = len(basis.kpoints)
        if os.environ.get("DFTK_MI_KBATCH_WAVES", "two") == "equal":
            default = max(16, -(-n_k // 2))
        else:
            # two waves (a few k-points spread over the mesh from random orbitals, then all the others from their nearest
            # solved neighbour) pay from a few dozen k-points on: Al 72 k-points, first step 39 ms (two equal waves of 36) ->
            # 25-27 ms at 1 ... 6 first-wave k-points (4.0-4.8 instead of 7.3 LOBPCG iterations per k-point); 8 / 12 k-points
            # (Si, graphene) are faster as ONE wave of random starts (277 vs 258-288, 75 vs 68-70 SCF it/s)
            default = n_k if n_k <= 24 else max(4, n_k // 16)
        return max(1, int(os.environ.get("DFTK_MI_KBATCH_CHAIN", str(default))))
    return basis.n_lanes


def diagonalize_all_kblocks(eigensolver, ham, nev_per_kpoint: int, psiguess=None, prec_type=PreconditionerTPA,
                            tol: float = 1e-6, miniter: int = 1, maxiter: int = 100, n_conv_check=None,
                            generator: torch.Generator | None = None, seed: int = 0, interpolate_kpoints: bool = True,
                            coarse_start: bool = True):
    """diag.jl:9-65.  ``interpolate_kpoints`` (default true as the reference): without a guess, a k-point starts
    from the interpolated solution of the PREVIOUS k-point (:39-42).  The reference's k loop is sequential; here the
    local k-points run concurrently on the basis' stream lanes, so "previous" means the previous k-point of the same
    lane (identical to the reference for ``n_lanes = 1``); the first k-point of every lane starts from random
    orbitals.  ``coarse_start`` (an extension, used when the basis carries a companion basis ``basis.coarse``): a k-point without
    a guess is first solved on the companion basis at Ecut / 4 from random orbitals, the result zero-padded into its sphere."""
    # k-points that start from random orbitals when no guess is given (the others interpolate): the first W of the k loop, or --
    # batched small blocks, two waves -- W k-points SPREAD over the list, so that every other k-point has a solved neighbour
    guesses = []
    basis_ = ham[0].basis if ham else None
    W_ = _chain_width(basis_) if ham else 1
    two_waves = (bool(ham) and eigensolver is lobpcg_hyper and getattr(basis_, "kbatch", False) and basis_.n_lanes == 1
                 and os.environ.get("DFTK_MI_KBATCH_WAVES", "two") != "equal")
    first_set = first_wave_indices(len(ham), W_, two_waves)
    in_first = set(first_set)
    n_matvec_coarse = 0
    for ik, Hk in enumerate(ham):                 # start vectors first, in k order: one deterministic RNG stream
        kpt, basis = Hk.kpoint, Hk.basis
        if kpt.n_G < nev_per_kpoint:
            raise ValueError(f"The size of the plane wave basis is {kpt.n_G}, and you are asking for "
                             f"{nev_per_kpoint} eigenvalues. Increase Ecut.")
        if psiguess is not None:
            g = psiguess[ik]
            if g.shape[1] != kpt.n_loc:
                raise ValueError(f"Mismatch in dimension between guess ({g.shape[1]}) and Hamiltonian ({kpt.n_loc})")
            if g.shape[0] > nev_per_kpoint:
                g = g[:nev_per_kpoint]
            elif g.shape[0] < nev_per_kpoint:
                extra = random_orbitals(basis, kpt, nev_per_kpoint - g.shape[0], generator)
                g = torch.cat([g, extra * np.sqrt(2 * kpt.n_G)], dim=0)
        elif interpolate_kpoints and ik not in in_first and basis.comm_pw.size == 1:
            g = None                                   # filled in by the lane from its previous k-point
        elif (coarse_start and getattr(basis, "coarse", None) is not None and eigensolver is lobpcg_hyper
              and kpt.spin == 1 and getattr(Hk, "potential", None) is not None and kpt.n_G >= 8 * nev_per_kpoint):
            # two-level start: solve on the companion basis first (counted in n_matvec_coarse, inside the caller's timing)
            g, nmv_c = _coarse_start_vectors(eigensolver, Hk, ik, nev_per_kpoint, prec_type, tol, miniter, maxiter,
                                             n_conv_check, generator, seed)
            n_matvec_coarse += nmv_c
        else:
            g = random_orbitals(basis, kpt, nev_per_kpoint, generator)
        guesses.append(g)
    if guesses:
        ham[0].basis.pre_call()

    basis0 = ham[0].basis if ham else None
    if ham and eigensolver is lobpcg_hyper and getattr(basis0, "kbatch", False) and basis0.n_lanes == 1:
        # many small k-blocks: ONE library call iterates a whole wave of them in lock-step (dftk_mi_lobpcg_multi); the
        # Gamma point, if it runs the real-symmetric iteration, keeps its own call.  With guesses for everybody (every
        # SCF step but the first) there is one wave; without, k-point ik starts from the interpolated solution of
        # k-point ik - W like the lanes of the non-batched path (W = _chain_width), i.e. waves of W k-points.
        W = _chain_width(basis0)
        results = [None] * len(ham)
        # waves: [0, W) from random orbitals, then ALL the others at once, each from the interpolated solution of the NEAREST
        # k-point of the first wave (DFTK_MI_KBATCH_WAVES=equal: waves of W, k-point ik from k-point ik - W, as before round 6)
        no_guess = any(g is None for g in guesses)
        equal_waves = os.environ.get("DFTK_MI_KBATCH_WAVES", "two") == "equal"
        if not no_guess:
            waves = [list(range(len(ham)))]
        elif equal_waves:
            waves = [list(range(w0, min(len(ham), w0 + W))) for w0 in range(0, len(ham), W)]
        else:
            waves = [first_set] + ([[ik for ik in range(len(ham)) if ik not in in_first]] if len(ham) > len(first_set) else [])
        first = np.array([np.asarray(ham[i].kpoint.coordinate, dtype=float) for i in first_set])
        for wave in waves:
            for ik in wave:
                if guesses[ik] is None:
                    if equal_waves:
                        src = ik - W
                    else:      # nearest solved k-point (fractional coordinates, periodic images)
                        src = first_set[nearest_index(first, ham[ik].kpoint.coordinate)]
                    guesses[ik] = interpolate_kpoint(results[src].X, ham[src].kpoint, ham[ik].kpoint)
            multi = [ik for ik in wave if not getattr(ham[ik].kpoint, "gamma_real", False)]
            if len(multi) > 1:
                out = lobpcg_hyper_multi([ham[ik] for ik in multi], [guesses[ik] for ik in multi], maxiter=maxiter,
                                         prec=prec_type is not None, tol=tol, n_conv_check=n_conv_check, miniter=miniter,
                                         seeds=[seed + ik for ik in multi])
                for ik, r_ in zip(multi, out):
                    results[ik] = r_
            for ik in wave:
                if results[ik] is None:
                    Hk = ham[ik]
                    results[ik] = eigensolver(Hk, guesses[ik], prec=prec_type(Hk) if prec_type is not None else None,
                                              tol=tol, miniter=miniter, maxiter=maxiter, n_conv_check=n_conv_check,
                                              seed=seed + ik)
        return dict(λ=[r.λ for r in results], X=[r.X for r in results],
                    residual_norms=[r.residual_norms for r in results], n_iter=[r.n_iter for r in results],
                    converged=all(r.converged for r in results), n_matvec=sum(r.n_matvec for r in results),
                    real_symmetric=[bool(getattr(r, "real_symmetric", False)) for r in results],
                    n_matvec_coarse=n_matvec_coarse)

    done = {}

    def solve(ik, Hk):                            # the k loop of diag.jl:24-48, lanes concurrently
        prec = prec_type(Hk) if prec_type is not None else None
        g = guesses[ik]
        if g is None:
            prev = ik - Hk.basis.n_lanes           # previous k-point of this lane (already solved: lanes run in order)
            g = interpolate_kpoint(done[prev].X, ham[prev].kpoint, Hk.kpoint)
        kw = {}
        if eigensolver is lobpcg_hyper and os.environ.get("DFTK_MI_AX_REUSE", "1") != "0":
            # the guess is the very block this k-point's last solve returned (an SCF step hands the orbitals back)
            kw["reuse_AX"] = psiguess is not None and g is psiguess[ik] and g is getattr(Hk.kpoint, "_last_X", None)
        done[ik] = eigensolver(Hk, g, prec=prec, tol=tol, miniter=miniter, maxiter=maxiter,
                               n_conv_check=n_conv_check, seed=seed + ik, **kw)
        if eigensolver is lobpcg_hyper:
            Hk.kpoint._last_X = done[ik].X
        return done[ik]
    results = ham[0].basis.run_on_lanes(solve, ham) if ham else []
    return dict(λ=[r.λ for r in results], X=[r.X for r in results],
                residual_norms=[r.residual_norms for r in results], n_iter=[r.n_iter for r in results],
                converged=all(r.converged for r in results), n_matvec=sum(r.n_matvec for r in results),
                real_symmetric=[bool(getattr(r, "real_symmetric", False)) for r in results],
                n_matvec_coarse=n_matvec_coarse)
