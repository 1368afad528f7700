#!/usr/bin/env python
"""NumPy model of dense_heev_lowest (csrc/eig_kernels.hip): the SAME schedule of Newton-Schulz iterations, column
selection and Cholesky-QR passes, on the host -- used to fix the constants (l_hat, iteration counts, tolerances)
before the kernels run on the GPU, and by tests/test_host_side.py as the statement of the algorithm."""
import numpy as np

L_HAT = 1e-2
GROW = 0.5 * np.sqrt(3.0 / (1 + L_HAT + L_HAT * L_HAT)) * 3.0      # growth per scaled iteration of |x| << l_hat


def a_of(l):
    return np.sqrt(3.0 / (1.0 + l + l * l))


def l_next(l, a):
    return 0.5 * a * l * (3.0 - a * a * l * l)


def choose_sigma(d, nev):
    """eig_choose_sigma_host (csrc/eig_kernels.hip): the nev-th smallest diagonal entry + half the mean level spacing of
    the nev smallest entries (Cauchy interlacing: at least nev eigenvalues lie below the largest eigenvalue of the
    principal submatrix of those entries, which for an LOBPCG Rayleigh-Ritz matrix is diagonal); inside a wide diagonal
    gap up to four spacings.  Returns (sigma, estimated distance to the nearest eigenvalue)."""
    ds = np.sort(d)
    dn, d1, dnext = ds[nev - 1], ds[0], ds[nev]
    spacing = (dn - d1) / max(nev - 1, 1)
    if not spacing > 0:
        spacing = 1e-8 * max(1.0, abs(dn))
    margin = 0.5 * spacing
    gap = dnext - dn
    if gap > 2 * margin:
        margin = min(0.5 * gap, 4 * spacing)
    return dn + margin, 0.2 * margin


def sign_split(A, sigma, gap_guess, log=None, max_rescue=3):
    n = A.shape[0]
    S = A - sigma * np.eye(n)
    nrm = np.abs(S).sum(axis=1).max()
    X = S / nrm
    l_true_guess = max(gap_guess / nrm, 1e-12)
    m1 = int(np.ceil(np.log(max(L_HAT / l_true_guess, 1.0)) / np.log(GROW)))
    m1 = min(m1, 30)
    its = 0
    I = np.eye(n)

    def step(X, a):
        Y = X.T @ X
        Bc = 1.5 * a * I - 0.5 * a ** 3 * Y
        return X @ Bc

    def err(X):
        Y = X.T @ X
        return np.linalg.norm(I - Y)

    for rescue in range(max_rescue + 1):
        for _ in range(m1):
            X = step(X, a_of(L_HAT)); its += 1
        l = L_HAT
        while l < 1 - 1e-9:
            a = a_of(l)
            X = step(X, a); its += 1
            l = l_next(l, a)
        # checked phase (a = 1)
        for _ in range(10):
            e = err(X)
            if log is not None: log.append(("check", its, e))
            if e < 1e-10:
                return X, its, nrm
            if e > 0.9: break
            X = step(X, 1.0); its += 1
            if e < 3e-6:
                return X, its, nrm
        m1 = 6
    return None, its, nrm


def heev_lowest(A, nev, log=None):
    n = A.shape[0]
    d = np.diag(A).copy()
    sigma, gap_guess = choose_sigma(d, nev)
    U, its, nrm = sign_split(A, sigma, gap_guess, log)
    if U is None:
        return None
    p = 0.5 * (1.0 - np.diag(U))
    k = int(round(p.sum()))
    if k < nev or k > n - 1:
        return None
    S = np.sort(np.argsort(-p, kind="stable")[:k])
    B = -0.5 * U[:, S]
    B[S, np.arange(k)] += 0.5
    conds = []
    for _ in range(2):
        O = B.T @ B
        R = np.linalg.cholesky(O).T
        conds.append(np.linalg.cond(R))
        B = B @ np.linalg.inv(R)
    X0 = (A - sigma * np.eye(n)) / nrm
    Gk = B.T @ (X0 @ B)
    th, W = np.linalg.eigh(0.5 * (Gk + Gk.T))
    V = B @ W[:, :nev]
    lam = sigma + nrm * th[:nev]
    if log is not None: log.append(("summary", its, k, conds, sigma))
    return lam, V


if __name__ == "__main__":
    rng = np.random.default_rng(1)

    def late(n, M, coupling):
        lam = np.sort(np.repeat(rng.uniform(-0.2, 0.3, M // 4 + 1), 4)[:M] + 1e-7 * rng.standard_normal(M))
        Bq, _ = np.linalg.qr(rng.standard_normal((n - M, n - M)))
        Bb = (Bq * rng.uniform(0.8, 40, n - M)) @ Bq.T
        E = coupling * rng.standard_normal((n - M, M))
        return np.block([[np.diag(lam), E.T], [E, Bb]])

    def early(n, M):
        lam = np.concatenate([np.sort(rng.uniform(-0.2, 2.0, M)), rng.uniform(0.0, 40, n - M)])
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        A = (Q * lam) @ Q.T
        # leading block diagonal like an LOBPCG RR matrix: rotate so that X block = Ritz vectors of the leading block
        w, Z = np.linalg.eigh(A[:M, :M])
        T = np.eye(n); T[:M, :M] = Z
        return T.T @ A @ T

    for name, A, M in [("late 1006 c=1e-3", late(1006, 503, 1e-3), 503), ("late 1509 c=1e-2", late(1509, 503, 1e-2), 503),
                       ("early 1006", early(1006, 503), 503), ("early 1509", early(1509, 503), 503),
                       ("late 518", late(518, 259, 1e-3), 259)]:
        log = []
        out = heev_lowest(A, M, log)
        ref = np.linalg.eigvalsh(A)[:M]
        if out is None:
            print(name, "FALLBACK", log)
            continue
        lam, V = out
        print(name, "its", log[-1][1], "k", log[-1][2], "condR %.1e %.1e" % tuple(log[-1][3]), "dlam %.1e" % np.abs(lam - ref).max(),
              "orth %.1e" % np.abs(V.T @ V - np.eye(M)).max(), "resid %.1e" % np.abs(A @ V - V * lam).max(),
              [("%d:%.1e" % (c[1], c[2])) for c in log if c[0] == "check"])
