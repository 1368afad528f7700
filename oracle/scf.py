"""Density, occupation and the SCF driver (oracle restatement).  Test infrastructure only.

Restates ``src/densities.jl:13-57`` (symmetrisation over ``basis.symmetries``: oracle/symmetry.py),
``src/occupation.jl:30-211`` + ``src/Smearing.jl`` (None / Fermi-Dirac / Gaussian),
``src/scf/nbands_algorithm.jl``, ``src/scf/scf_callbacks.jl:138-230`` (convergence, AdaptiveDiagtol),
``src/scf/anderson.jl:36-130``, ``src/scf/scf_solvers.jl:68-102`` (mixing rules: oracle/mixing.py) and
``src/scf/self_consistent_field.jl:80-289``.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import erfc

from .terms import energy_hamiltonian, guess_density
from .lobpcg import diagonalize_all_kblocks

EPS = np.finfo(float).eps


# ----------------------------------------------------------------------------- density
def compute_density(basis, psi, occupation, occupation_threshold=0.0):
    """rho(r) = sum_k w_k sum_n f_nk |psi_nk(r)|^2 (densities.jl:13-57)."""
    nx, ny, nz = basis.fft_size
    n_spin = basis.model.n_spin_components
    rho = np.zeros((n_spin, nz, ny, nx))                           # rho[:, :, :, kpt.spin] (densities.jl:29, :39)
    for ik, kpt in enumerate(basis.kpoints):
        occ = np.asarray(occupation[ik], dtype=float)
        for n in range(len(occ)):
            if abs(occ[n]) < occupation_threshold:
                continue
            psi_real = basis.ifft(kpt, psi[ik][:, n], normalize=False)
            rho[kpt.spin - 1] += occ[n] * basis.kweights[ik] * basis.ifft_normalization ** 2 * np.abs(psi_real) ** 2
    from .symmetry import symmetrize_rho
    rho = np.stack([symmetrize_rho(basis, r, do_lowpass=False) for r in rho])          # densities.jl:47
    return rho if n_spin == 2 else rho[0]


# ----------------------------------------------------------------------------- occupations
def smearing_occupation(kind, x):
    x = np.asarray(x, dtype=float)
    if kind == "none":
        return np.where(x > 0, 0.0, 1.0)
    if kind == "fermi_dirac":
        out = np.empty_like(x)
        pos = x > 0
        y = np.exp(-x[pos])
        out[pos] = y / (1 + y)
        out[~pos] = 1 / (1 + np.exp(x[~pos]))
        return out
    if kind == "gaussian":
        return erfc(x) / 2
    raise NotImplementedError(kind)


def _occupation_for(basis, eigenvalues, eF, temperature, smearing):
    inv_t = math.inf if temperature == 0 else 1 / temperature
    filled = basis.model.filled_occupation
    occ = []
    for ek in eigenvalues:
        if temperature == 0:
            x = np.where(ek - eF > 0, np.inf, -np.inf)
            x = np.where(ek == eF, 0.0, x)
        else:
            x = (ek - eF) * inv_t
        occ.append(filled * smearing_occupation(smearing if temperature > 0 else "none", x))
    return occ


def _excess(basis, eigenvalues, eF, temperature, smearing):
    occ = _occupation_for(basis, eigenvalues, eF, temperature, smearing)
    return sum(w * np.sum(o) for w, o in zip(basis.kweights, occ)) - basis.model.n_electrons


def _guess_fermi_level_intocc(basis, eigenvalues):
    """occupation.jl:190-211."""
    filled = basis.model.filled_occupation
    n_fill = -(-basis.model.n_electrons // (getattr(basis.model, "n_spin_components", 1) * filled))
    homo = max(ek[n_fill - 1] for ek in eigenvalues)
    lumo = min((np.min(ek[n_fill:]) if len(ek) > n_fill else math.inf) for ek in eigenvalues)
    return homo + 1 if lumo == math.inf else (homo + lumo) / 2


def compute_occupation(basis, eigenvalues, tol_n_elec=1e-6):
    """compute_occupation(basis, eigenvalues, fermialg) (occupation.jl:53-132,160-181)."""
    model = basis.model
    T, sm = model.temperature, model.smearing
    eF = _guess_fermi_level_intocc(basis, eigenvalues)
    if T == 0:
        if abs(_excess(basis, eigenvalues, eF, 0.0, "none")) > tol_n_elec:
            raise RuntimeError("Unable to find non-fractional occupations; add a temperature")
    else:
        ex = _excess(basis, eigenvalues, eF, T, sm)
        if abs(ex) >= tol_n_elec / 10:
            if ex < 0:
                lo, hi = eF, max(np.max(e) for e in eigenvalues) + 1
            else:
                lo, hi = min(np.min(e) for e in eigenvalues) - 1, eF
            for _ in range(200):     # Roots.Bisection to machine precision
                mid = (lo + hi) / 2
                if mid == lo or mid == hi:
                    break
                if _excess(basis, eigenvalues, mid, T, sm) < 0:
                    lo = mid
                else:
                    hi = mid
            eF = (lo + hi) / 2
    return _occupation_for(basis, eigenvalues, eF, T, sm), eF


# ----------------------------------------------------------------------------- band counts
def default_n_bands(model, temperature_factor=1.05):
    min_n = -(-model.n_electrons // (model.n_spin_components * model.filled_occupation))
    factor = 1.0 if model.temperature == 0 else temperature_factor
    return int(math.ceil(min_n * factor))


class AdaptiveBands:
    """nbands_algorithm.jl:52-110."""

    def __init__(self, model, n_bands_converge=None, occupation_threshold=1e-6, gap_min=1e-2):
        self.n_bands_converge = default_n_bands(model, 1.05) if n_bands_converge is None else n_bands_converge
        self.n_bands_compute = max(3 + self.n_bands_converge, default_n_bands(model, 1.20))
        self.occupation_threshold = occupation_threshold
        self.gap_min = gap_min

    def determine_n_bands(self, occupation, eigenvalues, psi):
        if occupation is None:
            n_compute = self.n_bands_compute if psi is None else max(
                self.n_bands_compute, max(p.shape[1] for p in psi))
            n_converge = (self.n_bands_converge + self.n_bands_compute) // 2
            return n_converge, n_compute
        n_occ = 0
        for occk in occupation:
            idx = np.nonzero(np.abs(occk) >= self.occupation_threshold)[0]
            n_occ = max(n_occ, (idx[-1] + 1) if len(idx) else len(occk) + 1)
        n_converge = max(self.n_bands_converge, n_occ)
        n_compute_eps = 0
        if eigenvalues is not None:
            for ek in eigenvalues:
                if n_converge > len(ek):
                    n_compute_eps = max(n_compute_eps, len(ek) + 1)
                    continue
                idx = np.nonzero(ek <= ek[n_converge - 1] + self.gap_min)[0]
                n_compute_eps = max(n_compute_eps, (idx[-1] + 1) if len(idx) else len(ek) + 1)
        n_compute = max(self.n_bands_compute, n_compute_eps, n_converge + 3)
        if psi is not None:
            n_compute = max(n_compute, max(p.shape[1] for p in psi))
        return n_converge, n_compute


def next_density(basis, ham, nbandsalg, psi=None, eigenvalues=None, occupation=None, tol=1e-6,
                 rng=None):
    """self_consistent_field.jl:80-129."""
    n_conv, n_comp = nbandsalg.determine_n_bands(occupation, eigenvalues, psi)
    if psi is not None:
        n_comp = max(n_comp, max(p.shape[1] for p in psi))
    eig = diagonalize_all_kblocks(ham, n_comp, psiguess=psi, n_conv_check=n_conv, tol=tol,
                                  miniter=1, rng=rng)
    occ, eF = compute_occupation(basis, eig["λ"], tol_n_elec=nbandsalg.occupation_threshold)
    rho = compute_density(basis, eig["X"], occ, nbandsalg.occupation_threshold)
    return dict(psi=eig["X"], eigenvalues=eig["λ"], occupation=occ, eF=eF, rho=rho,
                diagonalization=eig, n_bands_converge=n_conv, n_matvec=eig["n_matvec"])


# ----------------------------------------------------------------------------- Anderson
class AndersonAcceleration:
    """anderson.jl:36-130."""

    def __init__(self, m=10, maxcond=1e6, errorfactor=1e5):
        self.iterates, self.residuals, self.errors = [], [], []
        self.m, self.maxcond, self.errorfactor = m, maxcond, errorfactor

    def _push(self, x, Pfx):
        self.iterates.append(x.ravel().copy())
        self.residuals.append(Pfx.ravel().copy())
        self.errors.append(float(np.linalg.norm(Pfx)))
        if len(self.iterates) > self.m:
            for lst in (self.iterates, self.residuals, self.errors):
                lst.pop(0)

    def _delete(self, idxs):
        for i in sorted(idxs, reverse=True):
            for lst in (self.iterates, self.residuals, self.errors):
                lst.pop(i)

    def __call__(self, x, alpha, Pfx):
        if self.m == 0 or not self.iterates:
            if self.m != 0:
                self._push(x, Pfx)
            return x + alpha * Pfx
        min_error = min(self.errors + [float(np.linalg.norm(Pfx))])
        drop = [i for i, e in enumerate(self.errors[:-1]) if e > self.errorfactor * min_error]
        if drop:
            self._delete(drop)
        pf = Pfx.ravel()
        Mmat = np.stack(self.residuals, axis=1) - pf[:, None]
        while True:
            Q, Rm = np.linalg.qr(Mmat)
            if Mmat.shape[1] > 1 and np.linalg.cond(Rm) > self.maxcond:
                idrop = int(np.argmax(self.errors[:-1]))
                self._delete([idrop])
                Mmat = np.delete(Mmat, idrop, axis=1)
                continue
            break
        xn = x.ravel() + alpha * pf
        betas = -np.linalg.solve(Rm, Q.T @ pf)
        for ib, beta in enumerate(betas):
            xn = xn + beta * (self.iterates[ib] - x.ravel() + alpha * (self.residuals[ib] - pf))
        self._push(x, Pfx)
        return xn.reshape(x.shape)


def determine_diagtol(n_iter, history_drho, ratio=0.2, diagtol_max=0.005, diagtol_first=None):
    """AdaptiveDiagtol (scf_callbacks.jl:191-212)."""
    if diagtol_first is None:
        diagtol_first = 6 * diagtol_max
    if n_iter <= 1:
        return min(diagtol_first, 5 * diagtol_max)
    diagtol = min(history_drho) * ratio
    assert math.isfinite(diagtol)
    return float(np.clip(diagtol, 100 * EPS, diagtol_max))


def default_diagtol_params(model, tol):
    """``default_diagtolalg`` (scf_callbacks.jl:220-230) as keyword arguments of ``determine_diagtol``:
    ``TermExactExchange`` -> ratio 5e-4; any ``TermNonlinear`` (Hartree ``hartree.jl:24``, Xc ``xc.jl:75`` unless it
    has no functionals, ``xc.jl:33``; LocalNonlinearity) -> the plain ``AdaptiveDiagtol()``; linear models only ->
    ``diagtol_first = tol / 5``."""
    terms = tuple(model.terms)
    if "ExactExchange" in terms:
        return dict(ratio=5e-4)
    if "Hartree" in terms or "LocalNonlinearity" in terms or ("Xc" in terms and len(model.functionals) > 0):
        return dict()
    return dict(diagtol_first=tol / 5)


def self_consistent_field(basis, rho=None, psi=None, tol=1e-6, maxiter=100, damping=0.8,
                          nbandsalg=None, is_converged=None, callback=None, rng=None,
                          anderson_m=10, mixing=None):
    """self_consistent_field.jl:164-289 with ScfAndersonDensitySolver; ``mixing`` defaults to ``LdosMixing()`` as
    the reference (:177), which is simple mixing at T = 0 (chi0models.jl:32)."""
    rng = np.random.default_rng(0) if rng is None else rng
    if mixing is None:
        from .mixing import LdosMixing
        mixing = LdosMixing()
    if rho is None:
        rho = guess_density(basis)
    if nbandsalg is None:
        nbandsalg = AdaptiveBands(basis.model)
    if is_converged is None:
        is_converged = lambda info: info["history_drho"][-1] < tol  # noqa: E731  (ScfConvergenceDensity)
    info = dict(psi=psi, occupation=None, eigenvalues=None, eF=None, n_iter=0, n_matvec=0,
                converged=False, history_Etot=[], history_drho=[], rho=rho)
    accel = AndersonAcceleration(m=anderson_m)
    rho_in = rho
    for _i in range(maxiter):
        n_iter = info["n_iter"] + 1
        _, ham = energy_hamiltonian(basis, info["psi"], info["occupation"], rho=rho_in)
        info_for_tol = dict(n_iter=info["n_iter"], history_drho=info["history_drho"])
        diagtol = determine_diagtol(info_for_tol["n_iter"], info_for_tol["history_drho"],
                                    **default_diagtol_params(basis.model, tol))
        nxt = next_density(basis, ham, nbandsalg, psi=info["psi"], eigenvalues=info["eigenvalues"],
                           occupation=info["occupation"], tol=diagtol, rng=rng)
        energies, _ = energy_hamiltonian(basis, nxt["psi"], nxt["occupation"], rho=nxt["rho"],
                                         eigenvalues=nxt["eigenvalues"], eF=nxt["eF"])
        drho = nxt["rho"] - rho_in
        info = dict(info, **nxt)
        info["n_iter"] = n_iter
        info["n_matvec"] = info.get("n_matvec_total", 0) + nxt["n_matvec"]
        info["n_matvec_total"] = info["n_matvec"]
        info["energies"] = energies
        info["history_Etot"] = info["history_Etot"] + [energies.total]
        info["history_drho"] = info["history_drho"] + [float(np.linalg.norm(drho) * math.sqrt(basis.dvol))]
        info["converged"] = bool(is_converged(info))
        if callback is not None:
            callback(info)
        if info["converged"]:
            break
        # fixpoint map returns rho_in + mix_density(mixing, drho) (:247); the solver damps + accelerates the
        # preconditioned residual (scf_solvers.jl:85-98)
        pf = mixing.mix_density(basis, drho, eF=nxt["eF"], eigenvalues=nxt["eigenvalues"], psi=nxt["psi"],
                                occupation=nxt["occupation"], rho_in=rho_in)
        rho_in = accel(rho_in, damping, pf)
    energies, ham = energy_hamiltonian(basis, info["psi"], info["occupation"], rho=info["rho"],
                                       eigenvalues=info["eigenvalues"], eF=info["eF"])
    info["energies"] = energies
    info["ham"] = ham
    return info
