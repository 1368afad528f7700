"""SCF mixing rules on the device (host mirror of src/scf/mixing.jl, src/scf/chi0models.jl, src/postprocess/dos.jl).

``mix_density(mixing, basis, dF; info...) -> d_rho`` approximates the inverse Jacobian of the SCF map
(mixing.jl:1-20).  Everything cube-sized stays in HBM: the Fourier multipliers run through the library's cube
FFTs (``basis.fft`` / ``basis.irfft``), the LDOS is a second pass of the density kernel with the weights
``-f'((e - eF)/T)/T`` (dos.jl:43-62, ``dftk_mi_density_accumulate``), GMRES works on cube-sized torch vectors and
only its Hessenberg matrix lives on the host.

* ``SimpleMixing`` (:36-39), ``KerkerMixing`` (:54-105), ``KerkerDosMixing`` (:117-137), ``DielectricMixing`` (:152-172)
* ``LdosMixing`` / ``HybridMixing`` / ``Chi0Mixing`` (:199-290) with ``LdosModel`` and ``DielectricModel``
  (chi0models.jl:21-80), RPA kernel (hartree.jl:68-81), GMRES as KrylovKit's ``linsolve`` (krylovdim 30,
  tol = max(1e-12, reltol |b|), zero start vector).  ``LdosMixing()`` is the reference's default and reduces to
  simple mixing at T = 0 (chi0models.jl:32).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .densities import compute_density

SQRT_PI = math.sqrt(math.pi)
EPS = float(np.finfo(np.float64).eps)


def occupation_derivative(kind, x):
    """d/dx of Smearing.occupation (Smearing.jl:29) on host arrays."""
    x = np.asarray(x, dtype=float)
    if kind == "gaussian":
        return -np.exp(-x * x) / SQRT_PI
    if kind == "fermi_dirac":
        e = np.exp(-np.abs(x))
        return -e / (1 + e) ** 2
    raise NotImplementedError(f"smearing {kind}")


def default_smearing_temperature(model):
    """mixing.jl:296-301."""
    return "gaussian", max(model.temperature, min(0.1, 100 * model.temperature))


def compute_dos(eps, basis, eigenvalues, smearing=None, temperature=None):
    """Density of states at ``eps`` per spin component (dos.jl:18-34: ``D[sigma]``), summed over ``comm_kpts``."""
    smearing = smearing or basis.model.smearing
    temperature = basis.model.temperature if temperature is None else temperature
    if temperature == 0 or smearing == "none":
        raise ValueError("compute_dos only supports finite temperature")
    filled = basis.model.filled_occupation
    D = np.zeros(basis.model.n_spin_components)
    for kpt, w, ek in zip(basis.kpoints, basis.kweights, eigenvalues):
        x = (np.asarray(ek, dtype=float) - eps) / temperature
        D[kpt.spin - 1] -= filled * w / temperature * float(np.sum(occupation_derivative(smearing, x)))
    return np.asarray(basis.comm_kpts.sum_scalars(list(D)))


def ldos_weights(eps, basis, eigenvalues, psi, smearing=None, temperature=None):
    """The band weights of ``compute_ldos`` (dos.jl:52-58): ``-filled / T * f'((eps_kn - eps) / T)``."""
    smearing = smearing or basis.model.smearing
    temperature = basis.model.temperature if temperature is None else temperature
    if temperature == 0 or smearing == "none":
        raise ValueError("compute_ldos only supports finite temperature")
    filled = basis.model.filled_occupation
    weights = []
    for ek, p in zip(eigenvalues, psi):
        x = (np.asarray(ek, dtype=float)[:p.shape[0]] - eps) / temperature
        weights.append(-filled / temperature * occupation_derivative(smearing, x))
    return weights


def compute_ldos(eps, basis, eigenvalues, psi, smearing=None, temperature=None, weight_threshold=EPS):
    """Local density of states in real space (dos.jl:43-62): ``compute_density`` with modified weights."""
    return compute_density(basis, psi, ldos_weights(eps, basis, eigenvalues, psi, smearing, temperature), weight_threshold)


def gmres(apply, b: torch.Tensor, rtol: float, krylovdim: int = 30, maxiter: int = 100, atol: float = 1e-12):
    """Restarted GMRES on device vectors (Givens rotations on the host).  ONE host fetch per Krylov step: the new
    direction is orthogonalised against the whole basis with one stacked reduction (classical Gram-Schmidt,
    h = V' w together with <w, w>; the norm of the remainder follows from Pythagoras and is recomputed explicitly only
    when cancellation has eaten its digits) -- a fetch is a device synchronisation, and with a handful of cube-sized
    vectors the k + 2 sequential fetches of the modified scheme were most of the solver's time."""
    x = torch.zeros_like(b)
    nb = float(torch.linalg.norm(b).item())
    tol = max(atol, rtol * nb)
    r, beta = b.clone(), nb
    n = b.numel()
    # the Krylov basis grows geometrically (8 rows, doubling up to krylovdim + 1): a solve that ends in 2-3 steps must not
    # reserve 31 cube-sized vectors (1.75 GB at 192^3) next to a nearly full HBM
    Vbuf = torch.empty((min(8, krylovdim + 1), n), dtype=b.dtype, device=b.device)
    for _ in range(maxiter):
        if beta <= tol:
            break
        Vbuf[0] = (r / beta).reshape(-1)
        H = np.zeros((krylovdim + 1, krylovdim))
        g = np.zeros(krylovdim + 1)
        g[0] = beta
        cs, sn = np.zeros(krylovdim), np.zeros(krylovdim)
        k_used = 0
        for k in range(krylovdim):
            w = apply(Vbuf[k].reshape(b.shape)).reshape(-1)
            Vm = Vbuf[:k + 1]                                              # (k + 1, n) view, no copy
            vals = torch.cat([Vm @ w, (w @ w).reshape(1)]).cpu().numpy()
            h, ww = vals[:-1].copy(), float(vals[-1])
            w = w - torch.as_tensor(h, device=w.device) @ Vm
            rest = ww - float(np.dot(h, h))
            if rest < 1e-2 * ww:
                # cancellation: one pass of classical Gram-Schmidt leaves components of size eps * ||w|| / ||rest|| along
                # the basis, and Pythagoras has lost its digits -- a second pass ("twice is enough") and the remainder's
                # own norm; costs one more fetch on the steps where it is needed
                vals2 = torch.cat([Vm @ w, (w @ w).reshape(1)]).cpu().numpy()
                h2 = vals2[:-1]
                w = w - torch.as_tensor(h2, device=w.device) @ Vm
                h = h + h2
                rest = float(vals2[-1]) - float(np.dot(h2, h2))
                if rest < 1e-2 * float(vals2[-1]):
                    rest = float((w @ w).item())
            H[:k + 1, k] = h
            H[k + 1, k] = math.sqrt(max(rest, 0.0))
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
            if k + 1 >= Vbuf.shape[0]:
                grown = torch.empty((min(krylovdim + 1, 2 * Vbuf.shape[0]), n), dtype=b.dtype, device=b.device)
                grown[:Vbuf.shape[0]] = Vbuf
                Vbuf = grown
            Vbuf[k + 1] = w / hk1
        y = np.linalg.solve(np.triu(H[:k_used, :k_used]), g[:k_used])
        x = x + (torch.as_tensor(y, device=b.device) @ Vbuf[:k_used]).reshape(b.shape)
        # |g[k_used]| is the residual norm of the minimiser as long as the basis is orthonormal, which the second pass above
        # maintains; the TRUE residual is measured at every restart and once before an accepted exit of a long cycle
        if abs(g[k_used]) <= tol and k_used <= 8:
            beta = abs(g[k_used])
            break
        r = b - apply(x)
        beta = float(torch.linalg.norm(r).item())
    return x, beta <= tol


def _G2(basis):
    Gc = basis.G_vectors_cart_cube()
    return (Gc * Gc).sum(dim=-1)


def _torch_twin():
    """``DFTK_MI_TORCH_LOCAL=1``: the torch formulation of the cube multipliers (parity twin of the library calls)."""
    import os
    return os.environ.get("DFTK_MI_TORCH_LOCAL") is not None


def _torch_mix():
    """``DFTK_MI_TORCH_MIX=1`` (or ``DFTK_MI_TORCH_LOCAL=1``): GMRES and Anderson as host logic on torch vectors -- the parity
    twins of ``dftk_mi_chi0_mix`` / ``dftk_mi_anderson_step`` in the test-suite."""
    import os
    return os.environ.get("DFTK_MI_TORCH_MIX") is not None or os.environ.get("DFTK_MI_TORCH_LOCAL") is not None


def _filter(basis, entry, x, *params):
    """One of the library's Fourier-multiplier passes on a real cube (``dftk_mi_mix_kerker`` / ``_mix_dielectric`` /
    ``_chi0_dielectric_apply``): fft, multiplier evaluated per G in the kernel, irfft."""
    from . import _lib
    Bh = np.asfortranarray(basis.model.recip_lattice, dtype=np.float64)
    xin = x.to(torch.float64).contiguous()
    out = torch.empty_like(xin)
    basis.pre_call()
    _lib.check(getattr(basis.lib, entry)(basis._cube_handle, Bh.ctypes.data, *params, xin.data_ptr(), out.data_ptr()))
    basis.post_call()
    return out


def _filter_array(basis, mult, x):
    """irfft(mult .* fft(x)) with a real multiplier cube (``dftk_mi_cube_fourier_filter``)."""
    from . import _lib
    xin, m = x.to(torch.float64).contiguous(), mult.to(torch.float64).contiguous()
    out = torch.empty_like(xin)
    basis.pre_call()
    _lib.check(basis.lib.dftk_mi_cube_fourier_filter(basis._cube_handle, m.data_ptr(), xin.data_ptr(), out.data_ptr()))
    basis.post_call()
    return out


class SimpleMixing:
    """mixing.jl:36-39: J^-1 ~ 1."""

    def mix_density(self, basis, dF, **info):
        return dF


def _total(x):
    return x if x.dim() == 3 else x.sum(dim=0)


def _from_total_and_spin(tot, spin):
    return torch.stack([(tot + spin) / 2, (tot - spin) / 2])           # rho_from_total_and_spin (densities.jl:158-166)


class KerkerMixing:
    """mixing.jl:54-105: J^-1 ~ |G|^2 / (kTF^2 + |G|^2) on the TOTAL density; with collinear spin the spin density is
    left alone unless ``dDOS_vol`` = (DOS_up - DOS_down) / volume is given (:62-84, :97-103)."""

    def __init__(self, kTF: float = 0.8, dDOS_vol: float = 0.0):
        self.kTF, self.dDOS_vol = float(kTF), float(dDOS_vol)

    def _total_part(self, basis, dFtot):
        if not _torch_twin():
            return _filter(basis, "dftk_mi_mix_kerker", dFtot, self.kTF)
        G2 = _G2(basis)
        drho_f = basis.fft(dFtot) * (G2 / (self.kTF ** 2 + G2)) * basis.enforce_real_mask()
        drho = basis.irfft(drho_f)
        return drho + (dFtot.mean() - drho.mean())      # copy the DC component, otherwise it never gets updated

    def mix_density(self, basis, dF, **info):
        if dF.dim() == 3:
            return self._total_part(basis, dF)
        dFtot = _total(dF).contiguous()
        drho_tot = self._total_part(basis, dFtot)
        dFspin = dF[0] - dF[1]
        if abs(self.dDOS_vol) < EPS:
            return _from_total_and_spin(drho_tot, dFspin)
        G2 = _G2(basis)
        dspin_f = (basis.fft(dFspin) - basis.fft(dFtot) * (4 * math.pi * self.dDOS_vol) / (self.kTF ** 2 + G2))
        return _from_total_and_spin(drho_tot, basis.irfft(dspin_f * basis.enforce_real_mask()))


class KerkerDosMixing:
    """mixing.jl:117-137: Kerker with kTF (and the spin coupling) from the density of states at the Fermi level."""

    def __init__(self, smearing=None, temperature=None):
        self.smearing, self.temperature = smearing, temperature

    def mix_density(self, basis, dF, eF=None, eigenvalues=None, **info):
        sm, T = default_smearing_temperature(basis.model)
        sm = self.smearing or sm
        T = self.temperature if self.temperature is not None else T
        if T == 0:
            return dF
        dos_per_vol = compute_dos(eF, basis, eigenvalues, sm, T) / basis.model.unit_cell_volume
        ddos = float(dos_per_vol[0] - dos_per_vol[1]) if len(dos_per_vol) == 2 else 0.0
        return KerkerMixing(kTF=math.sqrt(4 * math.pi * float(np.sum(dos_per_vol))), dDOS_vol=ddos).mix_density(basis, dF)


class DielectricMixing:
    """mixing.jl:152-172: J^-1 ~ (kTF^2 - C0 G^2) / (eps_r kTF^2 - C0 G^2), C0 = 1 - eps_r; "applied to rho and
    rho_spin in the same way": per spin channel, with ONE shift of the mean over the whole array."""

    def __init__(self, kTF: float = 0.8, eps_r: float = 10.0):
        self.kTF, self.eps_r = float(kTF), float(eps_r)

    def mix_density(self, basis, dF, **info):
        er, kTF = self.eps_r, self.kTF
        if er == 1:
            return dF
        if er > 1 / math.sqrt(EPS):
            return KerkerMixing(kTF).mix_density(basis, dF)
        if dF.dim() == 4:
            # the one-channel pass copies each channel's own mean; the reference multiplies the G = 0 entry of every
            # channel by 1 / eps_r and then adds mean(dF) - mean(d_rho) over the WHOLE 4-d array to all entries
            out = torch.stack([self.mix_density(basis, x.contiguous()) for x in dF])
            means = dF.mean(dim=(1, 2, 3), keepdim=True)
            return out - means + means / er + dF.mean() * (1 - 1 / er)
        if not _torch_twin():
            return _filter(basis, "dftk_mi_mix_dielectric", dF, kTF, er)
        C0 = 1 - er
        G2 = _G2(basis)
        drho = basis.irfft(basis.fft(dF) * ((kTF ** 2 - C0 * G2) / (er * kTF ** 2 - C0 * G2)))
        return drho + (dF.mean() - drho.mean())


class LdosModel:
    """chi0models.jl:21-45: chi0 = -Dloc(r) delta(r, r') + Dloc(r) Dloc(r') / D."""

    def __init__(self, smearing=None, temperature=None):
        self.smearing, self.temperature = smearing, temperature

    def __call__(self, basis, eigenvalues=None, psi=None, eF=None, **info):
        sm, T = default_smearing_temperature(basis.model)
        sm = self.smearing or sm
        T = self.temperature if self.temperature is not None else T
        if T == 0:
            return None
        ldos = compute_ldos(eF, basis, eigenvalues, psi, sm, T)
        amax, total = torch.stack([ldos.abs().max(), ldos.sum()]).cpu().numpy()        # one fetch
        if float(amax) < math.sqrt(EPS):
            return None
        tdos = float(total) * basis.dvol

        def apply(drho, dV, alpha=1.0):
            deF = (ldos * dV).sum() * (basis.dvol / tdos)        # stays on the device: no synchronisation per apply
            return drho + alpha * (ldos * deF - ldos * dV)
        return apply


class DielectricModel:
    """chi0models.jl:54-80 (localization = identity)."""

    def __init__(self, eps_r: float = 10.0, kTF: float = 0.8):
        self.eps_r, self.kTF = float(eps_r), float(kTF)

    def __call__(self, basis, **info):
        C0 = 1 - self.eps_r
        if C0 == 0:
            return None
        kTF = self.kTF
        if not _torch_twin():
            def apply(drho, dV, alpha=1.0):
                if dV.dim() == 4:
                    return drho + alpha * torch.stack([_filter(basis, "dftk_mi_chi0_dielectric_apply", v.contiguous(),
                                                               kTF, self.eps_r) for v in dV])
                return drho + alpha * _filter(basis, "dftk_mi_chi0_dielectric_apply", dV, kTF, self.eps_r)
            return apply
        G2 = _G2(basis)
        mult = C0 * kTF ** 2 * G2 / (4 * math.pi) / (kTF ** 2 - C0 * G2)

        def apply(drho, dV, alpha=1.0):
            return drho + alpha * basis.irfft(mult * basis.fft(dV))
        return apply


class Chi0Mixing:
    """mixing.jl:228-290: GMRES solve of (1 - chi0 vc) d_rho = dF in real space, RPA kernel."""

    def __init__(self, chi0terms, RPA: bool = True, reltol: float = 0.01):
        if not RPA:
            raise NotImplementedError("only the RPA (Hartree) kernel is on this path")
        self.chi0terms, self.reltol = list(chi0terms), float(reltol)
        self.last_gmres_applies = 0

    def _native(self, basis, dF, info):
        """The whole solve behind the C ABI (``dftk_mi_chi0_mix``: the LDOS / dielectric models and the restarted GMRES
        of this class as ONE library call, 12 launches and one host synchronisation per Krylov step); None when a chi0
        term is not one of the two models the library restates."""
        import ctypes as C
        from . import _lib
        ldos_model = diel = None
        for t in self.chi0terms:
            if type(t) is LdosModel and ldos_model is None:
                ldos_model = t
            elif type(t) is DielectricModel and diel is None:
                diel = t
            else:
                return None
        ldos = None
        if ldos_model is not None:
            sm, T = default_smearing_temperature(basis.model)
            sm = ldos_model.smearing or sm
            T = ldos_model.temperature if ldos_model.temperature is not None else T
            if T != 0:
                # (the SCF stepper hands over the LDOS it accumulated in the density pass of the same orbitals)
                ldos = info.get("ldos")
                if ldos is None:
                    ldos = compute_ldos(info.get("eF"), basis, info.get("eigenvalues"), info.get("psi"), sm, T)
                ldos = ldos.to(torch.float64).contiguous()
        if ldos is None and diel is None:
            return dF
        n_comp = 2 if dF.dim() == 4 else 1
        if ldos is not None and ldos.dim() != dF.dim():
            return None
        xin = dF.to(torch.float64).contiguous()
        out = torch.empty_like(xin)
        poisson = basis.terms.poisson
        Bh = np.asfortranarray(basis.model.recip_lattice, dtype=np.float64)
        n_app, conv = C.c_int(), C.c_int()
        basis.pre_call()
        _lib.check(basis.lib.dftk_mi_chi0_mix(
            basis._cube_handle, n_comp, Bh.ctypes.data, poisson.data_ptr() if poisson is not None else None,
            ldos.data_ptr() if ldos is not None else None, float(basis.dvol), 1 if diel is not None else 0,
            diel.kTF if diel is not None else 0.0, diel.eps_r if diel is not None else 1.0, xin.data_ptr(), self.reltol, 30, 100,
            out.data_ptr(), C.byref(n_app), C.byref(conv)))
        self.last_gmres_applies = n_app.value
        return out

    def extra_density_weights(self, basis, eigenvalues, eF, psi):
        """Band weights (and their threshold) of the LDOS this mixing will ask for, so that ``compute_density`` can accumulate
        it in the SAME pass over the orbitals (``dftk_mi_density_accumulate_multi2``); None when no LDOS is needed or the
        torch twins are switched on."""
        if _torch_mix() or eF is None:
            return None
        models = [t for t in self.chi0terms if type(t) is LdosModel]
        if len(models) != 1 or any(type(t) not in (LdosModel, DielectricModel) for t in self.chi0terms):
            return None
        sm, T = default_smearing_temperature(basis.model)
        sm = models[0].smearing or sm
        T = models[0].temperature if models[0].temperature is not None else T
        if T == 0 or sm == "none":
            return None
        return ldos_weights(eF, basis, eigenvalues, psi, sm, T), EPS

    def mix_density(self, basis, dF, **info):
        if not _torch_mix():
            out = self._native(basis, dF, info)
            if out is not None:
                return out
        applies = [a for a in (t(basis, **info) for t in self.chi0terms) if a is not None]
        if not applies:
            return dF                                   # "do not bother running GMRES": simple mixing
        poisson = basis.terms.poisson
        count = [0]

        def dielectric_adjoint(x):
            count[0] += 1
            # apply_kernel with RPA = true: the Hartree kernel of the TOTAL density, the same dV for both spin channels
            xt = _total(x).contiguous()
            if poisson is None:
                dV = torch.zeros_like(xt)
            elif _torch_twin():
                dV = basis.irfft(poisson * basis.fft(xt))
            else:
                dV = _filter_array(basis, poisson, xt)
            if x.dim() == 4:
                dV = torch.stack([dV, dV])
            dV = dV - dV.mean()
            out = x
            for a in applies:
                out = a(out, dV, -1.0)                  # eps dF -= chi0 dV
            return out - out.mean()
        dc = dF.mean()
        drho, _ = gmres(dielectric_adjoint, dF - dc, self.reltol)
        self.last_gmres_applies = count[0]
        # (the reference broadcasts d_rho from rank 0, mpi_bcast!: every rank of comm_kpts / comm_pw runs the same
        #  deterministic kernels on identical inputs here, so the replicas are already numerically identical)
        return drho + dc


def LdosMixing(smearing=None, temperature=None, **kw):
    """mixing.jl:221-225."""
    return Chi0Mixing([LdosModel(smearing, temperature)], **kw)


def HybridMixing(eps_r=10.0, kTF=0.8, smearing=None, temperature=None, **kw):
    """mixing.jl:199-205."""
    return Chi0Mixing([DielectricModel(eps_r, kTF), LdosModel(smearing, temperature)], **kw)
