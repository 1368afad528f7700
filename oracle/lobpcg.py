"""LOBPCG ("hyper" variant), TPA preconditioner and the k-loop driver (oracle restatement).

Restates ``src/eigen/lobpcg_hyper_impl.jl`` (rayleigh_ritz :141-171, B_ortho! not needed for
B = I, safe_cholesky :190-210, normest :212, ortho!(X) :216-261, drop_small! :264-268,
ortho!(X,Y,BY) :271-323, final_retval :325-338, compute_lambda :341-344, LOBPCG :354-582),
``src/eigen/diag_lobpcg_hyper.jl:5-18``, ``src/eigen/preconditioners.jl:27-78`` and
``src/eigen/diag.jl:9-65``.  B = I throughout (plane-wave basis is orthonormal).
Test infrastructure only.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg as sla

EPS = np.finfo(float).eps


def columnwise_norms(X):
    return np.sqrt(np.sum(np.abs(X) ** 2, axis=0))


def columnwise_dots(A, B):
    return np.sum(np.conj(A) * B, axis=0)


def mul_hermi(A, B):
    return A.conj().T @ B


def rayleigh_ritz(Y, AY, N):
    """lobpcg_hyper_impl.jl:141-171 (LAPACK divide & conquer, as Julia >= 1.12)."""
    XAX = mul_hermi(Y, AY)
    assert not np.any(np.isnan(XAX))
    XAX = (XAX + XAX.conj().T) / 2    # Hermitian(.) view of the upper triangle, symmetrised
    vals, vecs = sla.eigh(XAX, driver="evd")
    return vecs[:, :N], vals[:N]


def safe_cholesky(O, nchol=0, alpha=100.0):
    """lobpcg_hyper_impl.jl:190-210."""
    if nchol >= 5:
        return None, None, 10000
    nchol += 1
    try:
        R = sla.cholesky(O, lower=False)
        invR = sla.solve_triangular(R, np.eye(R.shape[0], dtype=R.dtype), lower=False)
        if np.any(np.isnan(invR)):
            raise np.linalg.LinAlgError("nan")
    except np.linalg.LinAlgError:
        O = O + alpha * EPS * np.linalg.norm(O) * np.eye(O.shape[0])
        return safe_cholesky(O, nchol, alpha * 10)
    return R, invR, nchol


def normest(M):
    d = np.diag(M)
    return np.max(np.abs(d)) + np.linalg.norm(M - np.diag(d))


def ortho_X(X, tol=2 * EPS):
    """ortho!(X) (lobpcg_hyper_impl.jl:216-261).  Returns (X, nchol, growth_factor)."""
    growth_factor = 1.0
    nchol_total = 0
    while True:
        O = mul_hermi(X, X)
        O = (O + O.conj().T) / 2
        R, invR, nchol = safe_cholesky(O)
        nchol_total += nchol
        if nchol > 10:
            U, _, Vh = np.linalg.svd(X, full_matrices=False)
            return U @ Vh, 100, 1.0
        X = X @ invR
        norminvR = normest(invR)
        growth_factor *= norminvR
        condR = normest(R) * norminvR
        estimated_error = EPS * condR ** 2
        if nchol == 1 and estimated_error < tol:
            break
    return X, nchol_total, growth_factor


def drop_small(X, rng, tol=2 * EPS):
    """drop_small! (:264-268): re-randomise columns whose norm fell below tol."""
    dropped = np.nonzero(columnwise_norms(X) <= tol)[0]
    for j in dropped:
        X[:, j] = rng.standard_normal(X.shape[0]) + 1j * rng.standard_normal(X.shape[0])
    return dropped


def ortho_XY(X, Y, BY, rng, tol=2 * EPS):
    """ortho!(X, Y, BY) (:271-323): make X orthonormal and orthogonal to Y."""
    X = X / columnwise_norms(X)[None, :]
    niter = 1
    while True:
        BYX = BY.conj().T @ X
        X = X - Y @ BYX
        dropped = drop_small(X, rng, tol)
        if len(dropped):
            X[:, dropped] -= Y @ (BY.conj().T @ X[:, dropped])
        if np.linalg.norm(BYX) < tol and niter > 1:
            break
        X, _ninner, growth_factor = ortho_X(X, tol)
        if growth_factor * EPS < tol:
            break
        if niter > 10:
            U, _, Vh = np.linalg.svd(X, full_matrices=False)
            return U @ Vh
        niter += 1
    return X


def compute_lambda(X, AX):
    return np.real(columnwise_dots(X, AX) / columnwise_dots(X, X))


class PreconditionerTPA:
    """Teter-Payne-Allan preconditioner (preconditioners.jl:27-78)."""

    def __init__(self, kin, default_shift=1.0):
        self.kin = np.asarray(kin, dtype=float)
        self.mean_kin = None
        self.default_shift = default_shift

    def precondprep(self, X):
        self.mean_kin = np.real(np.sum(np.abs(X) ** 2 * self.kin[:, None], axis=0))

    def ldiv(self, R):
        if self.mean_kin is None:
            return R / (self.kin + self.default_shift)[:, None]
        mk = self.mean_kin[None, :]
        return mk / (mk + self.kin[:, None]) * R


def LOBPCG(A, X, precon=None, tol=1e-10, maxiter=100, miniter=1, ortho_tol=2 * EPS,
           n_conv_check=None, rng=None, callback=None):
    """LOBPCG(A, X, I, precon, tol, maxiter; ...) (lobpcg_hyper_impl.jl:354-582), B = I.

    ``A`` is a callable applying H to a block; ``X`` the (N, M) initial guess.
    """
    rng = np.random.default_rng(0) if rng is None else rng
    N, M = X.shape
    if not N > 3 * M:
        raise ValueError("The eigenproblem is too small, and the iterative eigensolver will fail")
    if n_conv_check is None:
        n_conv_check = M
    resid_history = np.zeros((M, maxiter + 1))

    full_X = ortho_X(np.array(X, dtype=complex, copy=True), ortho_tol)[0]
    n_matvec = M
    full_AX = A(full_X)
    assert not np.any(np.isnan(full_AX))
    full_lam = compute_lambda(full_X, full_AX)
    # active-column storage; column j of these arrays is global column (lo + j)
    lo = 0                     # == nlocked: first active column
    P = np.zeros((N, M), dtype=complex)
    AP = np.zeros((N, M), dtype=complex)
    R = np.zeros((N, M), dtype=complex)
    new_X = full_X.copy()
    new_AX = full_AX.copy()
    nlocked = 0
    niter = 0
    cX = None
    Y = AY = None

    def final(niter_):
        lam, Xo, AXo, hist = full_lam, full_X, full_AX, resid_history
        if np.any(np.diff(lam) < 0):
            p = np.argsort(lam, kind="stable")
            lam, Xo, AXo, hist = lam[p], Xo[:, p], AXo[:, p], hist[p, :]
        return dict(λ=lam.copy(), X=Xo, AX=AXo, residual_norms=hist[:, niter_].copy(),
                    residual_history=hist[:, :niter_ + 1].copy(), n_matvec=n_matvec)

    while True:
        X = full_X[:, lo:]          # views of the active parts
        AX = full_AX[:, lo:]
        nact = X.shape[1]
        if niter > 0:
            Ract = R[:, :nact]
            AR = A(Ract)
            n_matvec += nact
            if niter > 1:
                Y = np.concatenate([X, Ract, P[:, :nact]], axis=1)
                AY = np.concatenate([AX, AR, AP[:, :nact]], axis=1)
            else:
                Y = np.concatenate([X, Ract], axis=1)
                AY = np.concatenate([AX, AR], axis=1)
            cX, lam_RR = rayleigh_ritz(Y, AY, M - nlocked)
            full_lam[lo:] = lam_RR
            new_X = Y @ cX
            new_AX = AY @ cX
        else:
            new_X = X.copy()
            new_AX = AX.copy()

        # residuals
        new_R = new_AX - new_X * full_lam[lo:][None, :]
        norms = columnwise_norms(new_R)
        resid_history[nlocked:nlocked + nact, niter] = norms

        if precon is not None:
            precon.precondprep(new_X)
            new_R = precon.ldiv(new_R)

        prev_nlocked = nlocked
        if niter >= miniter:
            for i in range(nlocked, M):
                if resid_history[i, niter] < tol:
                    nlocked += 1
                else:
                    break
        if callback is not None:
            callback(dict(n_iter=niter, n_matvec=n_matvec, n_locked=nlocked,
                          resid_history=resid_history, λ=full_lam))

        if nlocked >= n_conv_check:
            full_X[:, lo:] = new_X
            full_AX[:, lo:] = new_AX
            return final(niter)
        newly_locked = nlocked - prev_nlocked

        if niter > 0:
            # cP = (cX - e)[:, newly_locked:], e has ones on a lower diagonal (:488-501)
            ncx = M - prev_nlocked
            lenXn = ncx - newly_locked
            e = np.zeros((cX.shape[0], ncx), dtype=complex)
            for a in range(lenXn):
                e[newly_locked + a, a] = 1.0
            cP = (cX - e)[:, newly_locked:]
            cP = ortho_XY(cP, cX, cX, rng, tol=ortho_tol)
            new_P = Y @ cP
            new_AP = AY @ cP

        # update all X (even newly locked), R
        full_X[:, lo:] = new_X
        full_AX[:, lo:] = new_AX
        diffs = np.abs(columnwise_dots(full_X[:, lo:], full_X[:, lo:]) - 1)
        if np.any(diffs >= math.sqrt(EPS)):
            raise RuntimeError("LOBPCG is badly failing to keep the vectors normalized")

        # restrict to active
        lo = nlocked
        nact = M - nlocked
        R[:, :nact] = new_R[:, newly_locked:]
        if niter > 0:
            P[:, :nact] = new_P
            AP[:, :nact] = new_AP
            Z = np.concatenate([full_X, P[:, :nact]], axis=1)
        else:
            Z = full_X
        R[:, :nact] = ortho_XY(R[:, :nact], Z, Z, rng, tol=ortho_tol)

        if niter >= maxiter:
            break
        niter += 1
    return final(maxiter)


def lobpcg_hyper(A, X0, maxiter=100, prec=None, tol=None, n_conv_check=None, miniter=1, **kw):
    """diag_lobpcg_hyper.jl:5-18."""
    if tol is None:
        tol = 20 * X0.shape[0] * EPS
    res = LOBPCG(A, X0, prec, tol, maxiter, miniter=miniter, n_conv_check=n_conv_check, **kw)
    ncc = X0.shape[1] if n_conv_check is None else n_conv_check
    res["converged"] = bool(np.max(res["residual_norms"][:ncc]) < tol)
    res["n_iter"] = res["residual_history"].shape[1] - 1
    return res


def random_orbitals(n_G, howmany, rng):
    """orbitals.jl:82-86: randn + QR."""
    X = rng.standard_normal((n_G, howmany)) + 1j * rng.standard_normal((n_G, howmany))
    Q, _ = np.linalg.qr(X)
    return Q


def interpolate_kpoint(data_in, kpoint_in, kpoint_out):
    """interpolation.jl:96-115: coefficients of the shared plane waves, zero elsewhere, then ortho_qr."""
    if kpoint_in is kpoint_out:
        return data_in.copy()
    pos = np.searchsorted(kpoint_in.mapping, kpoint_out.mapping)
    pos = np.minimum(pos, len(kpoint_in.mapping) - 1)
    hit = kpoint_in.mapping[pos] == kpoint_out.mapping
    out = np.zeros((len(kpoint_out.mapping), data_in.shape[1]), dtype=complex)
    out[hit, :] = data_in[pos[hit], :]
    return np.linalg.qr(out)[0]


def diagonalize_all_kblocks(ham, nev_per_kpoint, psiguess=None, tol=1e-6, miniter=1, maxiter=100,
                            n_conv_check=None, prec=True, rng=None, interpolate_kpoints=True):
    """diag.jl:9-65 (``interpolate_kpoints=true`` by default as the reference: without a guess, k-point ik > 1
    starts from the interpolated solution of k-point ik - 1)."""
    rng = np.random.default_rng(0) if rng is None else rng
    results = []
    for ik, H in enumerate(ham):
        n_Gk = H.n_G
        if psiguess is not None:
            g = psiguess[ik]
            if g.shape[1] > nev_per_kpoint:
                g = g[:, :nev_per_kpoint]
            elif g.shape[1] < nev_per_kpoint:
                extra = nev_per_kpoint - g.shape[1]
                X0 = np.concatenate([g, rng.standard_normal((n_Gk, extra))
                                     + 1j * rng.standard_normal((n_Gk, extra))], axis=1)
                g = np.linalg.qr(X0)[0]
        elif interpolate_kpoints and ik > 0:
            g = interpolate_kpoint(results[ik - 1]["X"], ham[ik - 1].kpoint, H.kpoint)
        else:
            g = random_orbitals(n_Gk, nev_per_kpoint, rng)
        P = PreconditionerTPA(H.kinetic) if (prec and H.kinetic is not None) else None
        results.append(lobpcg_hyper(H.mul, g, prec=P, tol=tol, miniter=miniter, maxiter=maxiter,
                                    n_conv_check=n_conv_check, rng=rng))
    return dict(λ=[np.real(r["λ"]) for r in results], X=[r["X"] for r in results],
                residual_norms=[r["residual_norms"] for r in results],
                n_iter=[r["n_iter"] for r in results],
                converged=all(r["converged"] for r in results),
                n_matvec=sum(r["n_matvec"] for r in results))
