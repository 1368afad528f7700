"""SCF mixing (preconditioners of the density fixed point), oracle restatement.  Test infrastructure only.

Restates ``src/scf/mixing.jl`` (SimpleMixing :36-39, KerkerMixing :54-105, KerkerDosMixing :117-137,
DielectricMixing :152-172, HybridMixing / LdosMixing / chi0Mixing :199-290, default_smearing_temperature
:296-301), ``src/scf/chi0models.jl:21-80`` (LdosModel, DielectricModel), ``src/postprocess/dos.jl:18-62``
(compute_dos, compute_ldos), ``src/terms/hartree.jl:68-81`` (RPA kernel) and the GMRES of KrylovKit's
``linsolve`` (krylovdim 30, restarts, tol = max(1e-12, rtol |b|), zero start vector).
"""
from __future__ import annotations

import math

import numpy as np

from .scf import compute_density

SQRT_PI = math.sqrt(math.pi)


def occupation_derivative(kind, x):
    """d/dx of Smearing.occupation (Smearing.jl:29): Gaussian erfc(x)/2, Fermi-Dirac 1/(1+e^x)."""
    x = np.asarray(x, dtype=float)
    if kind == "gaussian":
        return -np.exp(-x * x) / SQRT_PI
    if kind == "fermi_dirac":
        e = np.exp(-np.abs(x))
        return -e / (1 + e) ** 2
    raise NotImplementedError(kind)


def default_smearing_temperature(model):
    """mixing.jl:296-301."""
    return "gaussian", max(model.temperature, min(0.1, 100 * model.temperature))


def compute_dos(eps, basis, eigenvalues, smearing, temperature):
    """dos.jl:18-34: one entry per spin component (the k-point list holds all spin-up blocks, then all spin-down)."""
    filled = basis.model.filled_occupation
    D = np.zeros(basis.model.n_spin_components)
    for kpt, w, ek in zip(basis.kpoints, basis.kweights, eigenvalues):
        x = (np.asarray(ek) - eps) / temperature
        D[kpt.spin - 1] -= filled * w / temperature * float(np.sum(occupation_derivative(smearing, x)))
    return D


def compute_ldos(eps, basis, eigenvalues, psi, smearing, temperature, weight_threshold=np.finfo(float).eps):
    """dos.jl:43-62: the density routine with weights -filled/T f'((e - eps)/T)."""
    filled = basis.model.filled_occupation
    weights = [-filled / temperature * occupation_derivative(smearing, (np.asarray(ek) - eps) / temperature)
               for ek in eigenvalues]
    weights = [w[:p.shape[1]] for w, p in zip(weights, psi)]
    return compute_density(basis, psi, weights, weight_threshold)


def gmres(apply, b, rtol, krylovdim=30, maxiter=100, atol=1e-12):
    """Restarted GMRES (KrylovKit linsolve, non-hermitian), zero start vector."""
    shape = b.shape
    b = b.ravel()
    x = np.zeros_like(b)
    tol = max(atol, rtol * float(np.linalg.norm(b)))
    r = b.copy()
    beta = float(np.linalg.norm(r))
    for _ in range(maxiter):
        if beta <= tol:
            break
        V = [r / beta]
        H = np.zeros((krylovdim + 1, krylovdim))
        g = np.zeros(krylovdim + 1)
        g[0] = beta
        cs, sn = np.zeros(krylovdim), np.zeros(krylovdim)
        k_used = 0
        for k in range(krylovdim):
            w = apply(V[k].reshape(shape)).ravel()
            for j in range(k + 1):
                H[j, k] = float(np.dot(V[j], w))
                w = w - H[j, k] * V[j]
            H[k + 1, k] = float(np.linalg.norm(w))
            for j in range(k):
                t = cs[j] * H[j, k] + sn[j] * H[j + 1, k]
                H[j + 1, k] = -sn[j] * H[j, k] + cs[j] * H[j + 1, k]
                H[j, k] = t
            d = math.hypot(H[k, k], H[k + 1, k])
            cs[k], sn[k] = (1.0, 0.0) if d == 0 else (H[k, k] / d, H[k + 1, k] / d)
            H[k, k] = d
            hk1 = H[k + 1, k]
            H[k + 1, k] = 0.0
            g[k + 1] = -sn[k] * g[k]
            g[k] = cs[k] * g[k]
            k_used = k + 1
            if abs(g[k + 1]) <= tol or hk1 == 0.0:
                break
            V.append(w / hk1)
        y = np.linalg.solve(np.triu(H[:k_used, :k_used]), g[:k_used])
        for j in range(k_used):
            x = x + y[j] * V[j]
        r = b - apply(x.reshape(shape)).ravel()
        beta = float(np.linalg.norm(r))
    return x.reshape(shape), beta <= tol


class SimpleMixing:
    def mix_density(self, basis, dF, **kw):
        return dF


def _tot(x):
    return x if x.ndim == 3 else x.sum(axis=0)


def _from_total_and_spin(tot, spin):
    return np.stack([(tot + spin) / 2, (tot - spin) / 2])           # rho_from_total_and_spin (densities.jl:158-166)


class KerkerMixing:
    """mixing.jl:54-105: the total density is preconditioned with G^2 / (kTF^2 + G^2); the spin density is left alone
    unless ``dDOS_vol`` = (DOS_up - DOS_down) / volume is given (:62-84)."""

    def __init__(self, kTF=0.8, dDOS_vol=0.0):
        self.kTF, self.dDOS_vol = kTF, dDOS_vol

    def mix_density(self, basis, dF, **kw):
        G2 = np.sum(basis.G_vectors_cart_cube() ** 2, axis=-1)
        dFtot = _tot(dF)
        dFtot_f = basis.fft_cube(dFtot)
        drho = basis.irfft_cube(basis.enforce_real(dFtot_f * G2 / (self.kTF ** 2 + G2)))
        drho = drho + (np.mean(dFtot) - np.mean(drho))    # copy the DC component, otherwise it never gets updated
        if dF.ndim == 3:
            return drho
        dFspin = dF[0] - dF[1]
        if abs(self.dDOS_vol) < np.finfo(float).eps:
            return _from_total_and_spin(drho, dFspin)
        dspin_f = basis.fft_cube(dFspin) - dFtot_f * (4 * math.pi * self.dDOS_vol) / (self.kTF ** 2 + G2)
        return _from_total_and_spin(drho, basis.irfft_cube(basis.enforce_real(dspin_f)))


class KerkerDosMixing:
    """mixing.jl:117-137."""

    def __init__(self, smearing=None, temperature=None):
        self.smearing, self.temperature = smearing, temperature

    def mix_density(self, basis, dF, eF=None, eigenvalues=None, **kw):
        sm, T = default_smearing_temperature(basis.model)
        sm = self.smearing or sm
        T = self.temperature if self.temperature is not None else T
        if T == 0:
            return dF
        dos_per_vol = compute_dos(eF, basis, eigenvalues, sm, T) / basis.model.unit_cell_volume
        ddos = dos_per_vol[0] - dos_per_vol[1] if len(dos_per_vol) == 2 else 0.0
        return KerkerMixing(kTF=math.sqrt(4 * math.pi * float(np.sum(dos_per_vol))), dDOS_vol=ddos).mix_density(basis, dF)


class DielectricMixing:
    """mixing.jl:152-172."""

    def __init__(self, kTF=0.8, eps_r=10.0):
        self.kTF, self.eps_r = kTF, eps_r

    def mix_density(self, basis, dF, **kw):
        er, kTF = self.eps_r, self.kTF
        if er == 1:
            return dF
        if er > 1 / math.sqrt(np.finfo(float).eps):
            return KerkerMixing(kTF).mix_density(basis, dF)
        C0 = 1 - er
        G2 = np.sum(basis.G_vectors_cart_cube() ** 2, axis=-1)
        mult = (kTF ** 2 - C0 * G2) / (er * kTF ** 2 - C0 * G2)
        if dF.ndim == 4:      # "applied to rho and rho_spin in the same way" (mixing.jl:152): per channel, one DC shift
            drho = np.stack([basis.irfft_cube(basis.fft_cube(x) * mult) for x in dF])
        else:
            drho = basis.irfft_cube(basis.fft_cube(dF) * mult)
        return drho + (np.mean(dF) - np.mean(drho))


class LdosModel:
    """chi0models.jl:21-45: chi0 = -Dloc(r) delta(r, r') + Dloc(r) Dloc(r') / D."""

    def __init__(self, smearing=None, temperature=None):
        self.smearing, self.temperature = smearing, temperature

    def __call__(self, basis, eigenvalues=None, psi=None, eF=None, **kw):
        sm, T = default_smearing_temperature(basis.model)
        sm = self.smearing or sm
        T = self.temperature if self.temperature is not None else T
        if T == 0:
            return None
        ldos = compute_ldos(eF, basis, eigenvalues, psi, sm, T)
        if np.max(np.abs(ldos)) < math.sqrt(np.finfo(float).eps):
            return None
        tdos = float(np.sum(ldos)) * basis.dvol

        def apply(drho, dV, alpha=1.0):
            deF = float(np.sum(ldos * dV)) * basis.dvol
            return drho + alpha * (-ldos * dV + ldos * deF / tdos)
        return apply


class DielectricModel:
    """chi0models.jl:54-80 (localization = identity)."""

    def __init__(self, eps_r=10.0, kTF=0.8):
        self.eps_r, self.kTF = eps_r, kTF

    def __call__(self, basis, **kw):
        C0 = 1 - self.eps_r
        if C0 == 0:
            return None
        kTF = self.kTF
        G2 = np.sum(basis.G_vectors_cart_cube() ** 2, axis=-1)
        mult = C0 * kTF ** 2 * G2 / (4 * math.pi) / (kTF ** 2 - C0 * G2)

        def apply(drho, dV, alpha=1.0):
            if dV.ndim == 4:
                return drho + alpha * np.stack([basis.irfft_cube(mult * basis.fft_cube(v)) for v in dV])
            return drho + alpha * basis.irfft_cube(mult * basis.fft_cube(dV))
        return apply


class Chi0Mixing:
    """mixing.jl:228-290: solve (1 - chi0 vc)^dagger-like system eps drho = dF in real space with GMRES, RPA kernel."""

    def __init__(self, chi0terms, RPA=True, reltol=0.01):
        if not RPA:
            raise NotImplementedError("only the RPA (Hartree) kernel is restated")
        self.chi0terms, self.reltol = chi0terms, reltol
        self.last_gmres_applies = 0

    def mix_density(self, basis, dF, **info):
        applies = [a for a in (t(basis, **info) for t in self.chi0terms) if a is not None]
        if not applies:
            return dF
        poisson = basis.terms.poisson
        count = [0]

        def dielectric_adjoint(x):
            count[0] += 1
            # apply_kernel with RPA = true: the Hartree kernel acts on the TOTAL density, the same dV for both spins
            dV = basis.irfft_cube(poisson * basis.fft_cube(_tot(x))) if poisson is not None else np.zeros_like(_tot(x))
            if x.ndim == 4:
                dV = np.stack([dV, dV])
            dV = dV - np.mean(dV)
            out = x.copy()
            for a in applies:
                out = a(out, dV, -1.0)
            return out - np.mean(out)
        dc = float(np.mean(dF))
        drho, _ = gmres(dielectric_adjoint, dF - dc, self.reltol)
        self.last_gmres_applies = count[0]
        return drho + dc


def LdosMixing(smearing=None, temperature=None, **kw):
    return Chi0Mixing([LdosModel(smearing, temperature)], **kw)


def HybridMixing(eps_r=10.0, kTF=0.8, smearing=None, temperature=None, **kw):
    return Chi0Mixing([DielectricModel(eps_r, kTF), LdosModel(smearing, temperature)], **kw)
