"""Energy terms -> operators, on the device (host mirror of src/terms/*.jl for the hot path).

Set-up (once per basis): kinetic multipliers (kinetic.jl:31-35), V_loc(r) (local.jl:108-138), the
Kleinman-Bylander projector matrix P and coupling D (nonlocal.jl:107-141,166-244), the Poisson
kernel (hartree.jl:29-45), Ewald (ewald.jl:64-168) and pseudopotential-correction
(psp_correction.jl:26-32) energies, Gaussian guess density (density_methods.jl).
Per SCF step: ``energy_hamiltonian`` (Hamiltonian.jl:200-236) builds V = V_loc + V_H + V_xc on
the cube (hand-written cube FFTs of the library + torch elementwise ops) and hands it to each
k-block; LDA exchange-correlation uses closed forms (Slater, VWN5, PW92).
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
from scipy.special import erfc

from . import _lib
from .hamiltonian import DftHamiltonianBlock
from .psp import (eval_psp_energy_correction, eval_psp_local_fourier, eval_psp_projector_fourier,
                  solid_harmonic_real)

TWO_PI = 2 * math.pi


def _structure_factor_cube(basis, r):
    """e^{-2 pi i G.r} on the cube as an outer product of three 1-D phase vectors."""
    from .basis import G_axis
    nx, ny, nz = basis.fft_size
    dev = basis.device
    px = torch.exp(-1j * TWO_PI * r[0] * torch.tensor(G_axis(nx), dtype=torch.float64, device=dev))
    py = torch.exp(-1j * TWO_PI * r[1] * torch.tensor(G_axis(ny), dtype=torch.float64, device=dev))
    pz = torch.exp(-1j * TWO_PI * r[2] * torch.tensor(G_axis(nz), dtype=torch.float64, device=dev))
    return pz[:, None, None] * py[None, :, None] * px[None, None, :]


def _atomic_superposition_abi(basis, kind, params_of, per_atom=False):
    """One library call (``dftk_mi_atomic_superposition``) for sum_s ff_s(|G|) sum_a e^{-2 pi i G.r_a} -> real cube.
    ``per_atom``: ``params_of(element, atom index)`` -- every atom is its own "species" (per-atom coefficients of the
    spin-density guess, density_methods.jl:126-152)."""
    model = basis.model
    species, positions = [], []
    if per_atom:
        par = np.zeros((len(model.atoms), 8))
        for ia, el in enumerate(model.atoms):
            vals = params_of(el, ia)
            par[ia, :len(vals)] = vals
            species.append(ia)
            positions.append(np.asarray(model.positions[ia], dtype=float))
    else:
        par = np.zeros((len(model.atom_groups), 8))
    for s_idx, g in enumerate(model.atom_groups if not per_atom else []):
        vals = params_of(model.atoms[g[0]])
        par[s_idx, :len(vals)] = vals
        for ia in g:
            species.append(s_idx)
            positions.append(np.asarray(model.positions[ia], dtype=float))
    species = np.asarray(species, dtype=np.int32)
    positions = np.ascontiguousarray(np.asarray(positions, dtype=np.float64))
    Bh = np.asfortranarray(model.recip_lattice, dtype=np.float64)
    nx, ny, nz = basis.fft_size
    out = torch.empty((nz, ny, nx), dtype=torch.float64, device=basis.device)
    basis.pre_call()
    _lib.check(basis.lib.dftk_mi_atomic_superposition(basis._cube_handle, kind, Bh.ctypes.data, par.shape[0],
                                                      par.ctypes.data, len(species), species.ctypes.data,
                                                      positions.ctypes.data, out.data_ptr()))
    return out


def _use_setup_abi(basis):
    return basis.handle is not None and os.environ.get("DFTK_MI_TORCH_SETUP") is None


def compute_local_potential(basis):
    """local.jl:108-138."""
    model = basis.model
    if _use_setup_abi(basis):
        return _atomic_superposition_abi(basis, 0, lambda el: [el.psp.rloc, float(el.psp.Zion)] + list(el.psp.cloc)[:4])
    Gnorm = torch.linalg.norm(basis.G_vectors_cart_cube(), dim=-1)
    pot = torch.zeros(Gnorm.shape, dtype=torch.complex128, device=basis.device)
    for group in model.atom_groups:
        ff = eval_psp_local_fourier(model.atoms[group[0]].psp, Gnorm) / math.sqrt(model.unit_cell_volume)
        sf = torch.zeros_like(pot)
        for ia in group:
            sf += _structure_factor_cube(basis, model.positions[ia])
        pot += sf * ff
    pot = pot * basis.enforce_real_mask()
    return basis.irfft(pot)


def build_projector_form_factors(psp, Gpk_cart):
    """nonlocal.jl:205-244; rows ordered (l, m, i): offset_l + n_proj_l (m + l) + i."""
    n_G = Gpk_cart.shape[0]
    pnorm = torch.linalg.norm(Gpk_cart, dim=1)
    out = torch.zeros((psp.count_n_proj(), n_G), dtype=torch.complex128, device=Gpk_cart.device)
    for l in range(psp.lmax + 1):
        n_l = psp.count_n_proj_radial(l)
        off = sum(psp.count_n_proj(ll) for ll in range(l))
        for i in range(1, n_l + 1):
            radial = eval_psp_projector_fourier(psp, i, l, pnorm)
            for m in range(-l, l + 1):
                out[off + n_l * (m + l) + (i - 1)] = radial * ((-1j) ** l) * solid_harmonic_real(l, m, Gpk_cart)
    return out      # (n_proj_psp, n_G)


def build_projection_vectors_abi(basis, kpt):
    """``build_projection_vectors`` (nonlocal.jl:166-244) written by ONE device kernel of the library
    (``dftk_mi_build_projectors_hgh``: HGH radial parts, real solid harmonics, (-i)^l, structure factors), straight
    into P; same column order as the torch construction below (species groups, atoms, (l, m, i))."""
    import ctypes as C
    model = basis.model
    groups = [g for g in model.atom_groups if model.atoms[g[0]].psp.count_n_proj() > 0]
    if not groups:
        return None
    rp = np.zeros((len(groups), 4))
    nproj = np.zeros((len(groups), 4), dtype=np.int32)
    species, positions = [], []
    for s_idx, g in enumerate(groups):
        psp = model.atoms[g[0]].psp
        for l in range(psp.lmax + 1):
            rp[s_idx, l] = psp.rp[l]
            nproj[s_idx, l] = psp.count_n_proj_radial(l)
        for ia in g:
            species.append(s_idx)
            positions.append(np.asarray(model.positions[ia], dtype=float))
    species = np.asarray(species, dtype=np.int32)
    positions = np.ascontiguousarray(np.asarray(positions, dtype=np.float64))
    G32 = kpt.G_vectors[kpt.row0:kpt.row1].to(torch.int32).contiguous()
    Bh = np.asfortranarray(model.recip_lattice, dtype=np.float64)
    kh = np.ascontiguousarray(kpt.coordinate, dtype=np.float64)
    n_p = C.c_int()
    args = (basis.handle, kpt.n_loc, G32.data_ptr(), Bh.ctypes.data, kh.ctypes.data, model.unit_cell_volume,
            len(groups), rp.ctypes.data, nproj.ctypes.data, len(species), species.ctypes.data, positions.ctypes.data)
    _lib.check(basis.lib.dftk_mi_build_projectors_hgh(*args, None, kpt.n_loc, C.byref(n_p)))
    P = torch.empty((n_p.value, kpt.n_loc), dtype=torch.complex128, device=basis.device)
    basis.pre_call()
    _lib.check(basis.lib.dftk_mi_build_projectors_hgh(*args, P.data_ptr(), kpt.n_loc, C.byref(n_p)))
    return P


def build_projection_vectors(basis, kpt):
    """nonlocal.jl:166-199 with torch ops (descriptor-only bases and the parity twin of the kernel above).
    Returns P as a (n_p, n_loc) tensor == column-major n_loc x n_p: the rows of this rank's plane-wave slab (all
    of the sphere without ``comm_pw``)."""
    model = basis.model
    rows = slice(kpt.row0, kpt.row1)
    Gpk = (kpt.G_vectors[rows].to(torch.float64)
           + torch.tensor(kpt.coordinate, dtype=torch.float64, device=basis.device)[None, :])
    Gpk_cart = kpt.Gplusk_cart[rows]
    rows = []
    for group in model.atom_groups:
        psp = model.atoms[group[0]].psp
        if psp.count_n_proj() == 0:
            continue
        ff = build_projector_form_factors(psp, Gpk_cart) / math.sqrt(model.unit_cell_volume)
        for ia in group:
            r = torch.tensor(model.positions[ia], dtype=torch.float64, device=basis.device)
            sf = torch.exp(-1j * TWO_PI * (Gpk[:, 0] * r[0] + Gpk[:, 1] * r[1] + Gpk[:, 2] * r[2]))
            rows.append(ff * sf[None, :])
    if not rows:
        return None
    return torch.cat(rows, dim=0).contiguous()


def build_projection_coefficients(model):
    """Dense block-diagonal D (nonlocal.jl:107-141), numpy on the host."""
    blocks = []
    for group in model.atom_groups:
        psp = model.atoms[group[0]].psp
        n = psp.count_n_proj()
        Dp = np.zeros((n, n))
        c = 0
        for l in range(psp.lmax + 1):
            for _m in range(-l, l + 1):
                nl = psp.count_n_proj_radial(l)
                Dp[c:c + nl, c:c + nl] = psp.h[l]
                c += nl
        blocks += [Dp] * len(group)
    n = sum(b.shape[0] for b in blocks)
    D = np.zeros((n, n))
    c = 0
    for b in blocks:
        D[c:c + b.shape[0], c:c + b.shape[0]] = b
        c += b.shape[0]
    return D


def compute_poisson_green_coeffs(basis):
    """hartree.jl:29-45."""
    Gc = basis.G_vectors_cart_cube()
    G2 = (Gc * Gc).sum(dim=-1)
    G2[0, 0, 0] = 1.0
    coeffs = 4 * math.pi / G2
    coeffs[0, 0, 0] = 0.0
    return coeffs * basis.enforce_real_mask()


def energy_ewald(lattice, charges, positions, eta=None):
    """ewald.jl:64-168, energy only (host, numpy + scipy.erfc)."""
    from .basis import estimate_integer_lattice_bounds
    lattice = np.asarray(lattice, dtype=float)
    q = np.asarray(charges, dtype=float)
    pos = np.asarray(positions, dtype=float).reshape(-1, 3)
    if q.size == 0:
        return 0.0
    recip = TWO_PI * np.linalg.inv(lattice.T)
    if eta is None:   # default_eta, ewald.jl:40-44
        eta = math.sqrt(math.sqrt(1.69 * np.linalg.norm(recip / TWO_PI) / np.linalg.norm(lattice))) / 2
    max_exp = -math.log(np.finfo(float).eps) + 5
    max_erfc = math.sqrt(max_exp)
    Glims = estimate_integer_lattice_bounds(recip, math.sqrt(max_exp) * 2 * eta)
    poslims = [float(np.max(pos[:, i][:, None] - pos[:, i][None, :])) for i in range(3)]
    Rlims = estimate_integer_lattice_bounds(lattice, max_erfc / eta, poslims)
    vol = abs(np.linalg.det(lattice))
    # reciprocal part
    rng = [np.arange(-g, g + 1) for g in Glims]
    G = np.stack(np.meshgrid(*rng, indexing="ij"), axis=-1).reshape(-1, 3)
    G = G[np.any(G != 0, axis=1)]
    Gsq = np.sum((G @ recip.T) ** 2, axis=1)
    sel = Gsq / (4 * eta ** 2) < max_exp + 40
    G, Gsq = G[sel], Gsq[sel]
    s_recip = -(q.sum() ** 2) / (4 * eta ** 2)
    for c0 in range(0, len(G), 32768):
        ph = TWO_PI * (G[c0:c0 + 32768] @ pos.T)
        sf2 = (np.cos(ph) @ q) ** 2 + (np.sin(ph) @ q) ** 2
        s_recip += float(np.sum(sf2 * np.exp(-Gsq[c0:c0 + 32768] / (4 * eta ** 2)) / Gsq[c0:c0 + 32768]))
    s_recip *= 4 * math.pi / vol
    # real-space part
    s_real = -2 * eta / math.sqrt(math.pi) * float(np.sum(q * q))
    rr = [np.arange(-g, g + 1) for g in Rlims]
    R = np.stack(np.meshgrid(*rr, indexing="ij"), axis=-1).reshape(-1, 3).astype(float)
    qq = q[:, None] * q[None, :]
    n = len(q)
    eye = np.eye(n, dtype=bool)
    for Rv in R:
        d = (pos[:, None, :] - pos[None, :, :] - Rv[None, None, :]) @ lattice.T
        dist = np.linalg.norm(d, axis=-1)
        if not Rv.any():
            dist = np.where(eye, np.inf, dist)
        m = dist * eta < max_erfc + 8
        if m.any():
            s_real += float(np.sum(qq[m] * erfc(eta * dist[m]) / dist[m]))
    return (s_recip + s_real) / 2


def energy_psp_correction(model):
    """psp_correction.jl:26-32."""
    corr = sum(len(g) * eval_psp_energy_correction(model.atoms[g[0]].psp) for g in model.atom_groups)
    return corr * sum(a.charge_ionic for a in model.atoms) / model.unit_cell_volume


# ---------------------------------------------------------------------------------- XC, closed forms
def _lda_x(rho):
    cx = -0.75 * (3 / math.pi) ** (1 / 3)
    r13 = rho ** (1 / 3)
    return cx * rho * r13, (4 / 3) * cx * r13


def _lda_c_vwn(rho):
    A, b, c, x0 = 0.0310907, 3.72744, 12.9352, -0.10498
    rs = (3 / (4 * math.pi * rho)) ** (1 / 3)
    x = torch.sqrt(rs)
    X = x * x + b * x + c
    X0 = x0 * x0 + b * x0 + c
    Q = math.sqrt(4 * c - b * b)
    at = torch.atan(Q / (2 * x + b))
    eps = A * (torch.log(x * x / X) + 2 * b / Q * at
               - b * x0 / X0 * (torch.log((x - x0) ** 2 / X) + 2 * (b + 2 * x0) / Q * at))
    dat = -2 * Q / (Q * Q + (2 * x + b) ** 2)
    deps_dx = A * (2 / x - (2 * x + b) / X + 2 * b / Q * dat
                   - b * x0 / X0 * (2 / (x - x0) - (2 * x + b) / X + 2 * (b + 2 * x0) / Q * dat))
    return rho * eps, eps - rs / 3 * deps_dx / (2 * x)


def _lda_c_pw(rho):
    a, a1, b1, b2, b3, b4 = 0.031091, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294
    rs = (3 / (4 * math.pi * rho)) ** (1 / 3)
    sq = torch.sqrt(rs)
    den = 2 * a * (b1 * sq + b2 * rs + b3 * rs * sq + b4 * rs * rs)
    lg = torch.log1p(1 / den)
    eps = -2 * a * (1 + a1 * rs) * lg
    dden = 2 * a * (b1 / (2 * sq) + b2 + 1.5 * b3 * sq + 2 * b4 * rs)
    deps = -2 * a * a1 * lg + 2 * a * (1 + a1 * rs) * dden / (den * den + den)
    return rho * eps, eps - rs / 3 * deps


_FUNCTIONALS = {"lda_x": _lda_x, "lda_c_vwn": _lda_c_vwn, "lda_c_pw": _lda_c_pw}


# GGA energy densities e(rho, sigma) per volume, sigma = |grad rho|^2 (closed forms of libxc's gga_x_pbe /
# gga_c_pbe; the derivatives de/drho, de/dsigma come from torch.autograd, the oracle's from complex steps).
def _gga_x_pbe_e(rho, sigma):
    kappa, mu = 0.8040, 0.2195149727645171
    cx = -0.75 * (3 / math.pi) ** (1 / 3)
    kf = (3 * math.pi ** 2 * rho) ** (1 / 3)
    s2 = sigma / (4 * kf * kf * rho * rho)
    return cx * rho ** (4 / 3) * (1 + kappa - kappa * kappa / (kappa + mu * s2))


def _gga_c_pbe_e(rho, sigma):
    beta, gamma = 0.06672455060314922, (1 - math.log(2)) / math.pi ** 2
    a, a1, b1, b2, b3, b4 = 0.0310907, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294    # lda_c_pw_mod
    rs = (3 / (4 * math.pi * rho)) ** (1 / 3)
    sq = torch.sqrt(rs)
    eps = -2 * a * (1 + a1 * rs) * torch.log1p(1 / (2 * a * (b1 * sq + b2 * rs + b3 * rs * sq + b4 * rs * rs)))
    kf = (3 * math.pi ** 2 * rho) ** (1 / 3)
    t2 = sigma * math.pi / (16 * kf * rho * rho)
    A = beta / gamma / torch.expm1(-eps / gamma)
    f1 = t2 + A * t2 * t2
    H = gamma * torch.log1p(beta / gamma * f1 / (1 + A * f1))
    return rho * (eps + H)


_GGA_FUNCTIONALS = {"gga_x_pbe": _gga_x_pbe_e, "gga_c_pbe": _gga_c_pbe_e}
_DENSITY_THRESHOLD = 1e-12


def xc_energy_potential(basis, rho):
    """xc_potential_real (xc.jl:84-160): E = sum e dvol, V = V_rho - 2 div(V_sigma grad rho); LDA terms
    have V_sigma = 0, for GGAs grad rho and the divergence are taken in Fourier space on the cube with
    the device FFT pipeline (LibxcDensities xc.jl:356-409, divergence_real :576-584)."""
    rc = torch.clamp(rho, min=1e-300)
    e = torch.zeros_like(rho)
    v = torch.zeros_like(rho)
    gga = []
    for name in basis.model.functionals:
        if name in _GGA_FUNCTIONALS:
            gga.append(name)
            continue
        if name not in _FUNCTIONALS:
            raise NotImplementedError(f"XC functional {name}: LDA (lda_x, lda_c_vwn, lda_c_pw) and PBE "
                                      f"(gga_x_pbe, gga_c_pbe) are on this path")
        ei, vi = _FUNCTIONALS[name](rc)
        e += ei
        v += vi
    tiny = rho <= 1e-300
    e = torch.where(tiny, torch.zeros_like(e), e)
    v = torch.where(tiny, torch.zeros_like(v), v)
    if gga:
        G = basis.G_vectors_cart_cube()                                      # (nz, ny, nx, 3)
        rho_f = basis.fft(rho)
        grad = [basis.irfft(1j * G[..., a] * rho_f) for a in range(3)]
        sigma = grad[0] ** 2 + grad[1] ** 2 + grad[2] ** 2
        ok = rho > _DENSITY_THRESHOLD
        zero = torch.zeros_like(rho)
        if os.environ.get("DFTK_MI_TORCH_LOCAL") is None:
            # e, de/drho, de/dsigma by the library (forward-mode derivatives of the closed forms on the device)
            mask = sum({"gga_x_pbe": 8, "gga_c_pbe": 16}[name] for name in gga)
            rho_c, sig_c = rho.contiguous(), sigma.contiguous()
            eg, vr, vsig = torch.empty_like(rho_c), torch.empty_like(rho_c), torch.empty_like(rho_c)
            basis.pre_call()
            _lib.check(basis.lib.dftk_mi_xc_gga(basis.handle, rho_c.numel(), rho_c.data_ptr(), sig_c.data_ptr(), mask,
                                                _DENSITY_THRESHOLD, eg.data_ptr(), vr.data_ptr(), vsig.data_ptr()))
            e = e + eg
            v = v + vr
        else:   # torch formulation (autograd), the parity twin of the kernel
            rho_s = torch.where(ok, rho, torch.ones_like(rho)).detach().requires_grad_(True)
            sig_s = torch.where(ok, sigma, torch.zeros_like(sigma)).detach().requires_grad_(True)
            with torch.enable_grad():
                eg = sum(_GGA_FUNCTIONALS[name](rho_s, sig_s) for name in gga)
                vr, vs = torch.autograd.grad(eg.sum(), (rho_s, sig_s))
            e = e + torch.where(ok, eg.detach(), zero)
            v = v + torch.where(ok, vr, zero)
            vsig = torch.where(ok, vs, zero)
        div = sum(1j * G[..., a] * basis.fft(vsig * grad[a]) for a in range(3))
        v = v - 2.0 * basis.irfft(div)
    return float(e.sum().item() * basis.dvol), v


# ---------------------------------------------------------------------------------- guess density
_DECAY = [(0.5, [0.6, 0.4, 0.3, 0.25, 0.2]),
          (2.5, [1.8, 1.4, 1.0, 0.7, 0.6, 0.5, 0.4, 0.35, 0.3]),
          (10.5, [2.0, 1.6, 1.25, 1.1, 1.0, 0.9, 0.8, 0.7, 0.7, 0.7, 0.6]),
          (12.5, [1.9, 1.5, 1.15, 1.0, 0.9, 0.8, 0.7, 0.6, 0.6, 0.6, 0.5]),
          (18.5, [2.0, 1.8, 1.5, 1.2, 1.0, 0.9, 0.85, 0.8, 0.75, 0.7, 0.65, 0.65, 0.6]),
          (28.5, [1.5, 1.25, 1.15, 1.05, 1.00, 0.95, 0.95, 0.9, 0.9, 0.85, 0.85, 0.80, 0.8, 0.75, 0.7]),
          (36.5, [2.0, 2.00, 1.60, 1.40, 1.25, 1.10, 1.00, 0.95, 0.90, 0.85, 0.80, 0.75, 0.7]),
          (math.inf, [2.0, 2.00, 1.55, 1.25, 1.15, 1.10, 1.05, 1.0, 0.95, 0.9, 0.85, 0.85, 0.8])]


def atom_decay_length(n_core, n_val):
    """ABINIT table (density_methods.jl:286-322)."""
    n_val = int(round(n_val))
    if n_val == 0:
        return 0.0
    for bound, data in _DECAY:
        if n_core < bound:
            return data[min(n_val, len(data)) - 1]
    raise AssertionError


def _gaussian_superposition(basis, coefficients=None):
    """atomic_density_superposition with the Gaussian valence densities (density_methods.jl:158-181, :236-244);
    ``coefficients``: one amplitude per atom (default 1)."""
    model = basis.model
    if _use_setup_abi(basis):
        if coefficients is None:
            return _atomic_superposition_abi(basis, 1, lambda el: [atom_decay_length(el.n_elec_core, el.charge_ionic),
                                                                   float(el.charge_ionic)])
        return _atomic_superposition_abi(basis, 1, lambda el, ia: [atom_decay_length(el.n_elec_core, el.charge_ionic),
                                                                   float(el.charge_ionic) * float(coefficients[ia])],
                                         per_atom=True)
    Gnorm = torch.linalg.norm(basis.G_vectors_cart_cube(), dim=-1)
    rho_G = torch.zeros(Gnorm.shape, dtype=torch.complex128, device=basis.device)
    for group in model.atom_groups:
        el = model.atoms[group[0]]
        ff = el.charge_ionic * torch.exp(-(Gnorm * atom_decay_length(el.n_elec_core, el.charge_ionic)) ** 2)
        sf = torch.zeros_like(rho_G)
        for ia in group:
            sf += _structure_factor_cube(basis, model.positions[ia]) * (1.0 if coefficients is None else float(coefficients[ia]))
        rho_G += sf * ff / math.sqrt(model.unit_cell_volume)
    return basis.irfft(rho_G * basis.enforce_real_mask())


def guess_density(basis, magnetic_moments=()):
    """``guess_density(basis, magnetic_moments)`` (density_methods.jl:35-38, :102-152): Gaussian superposition renormalised
    to n_electrons; for a collinear model the result has shape (2, nz, ny, nx) = ((tot + spin) / 2, (tot - spin) / 2) with
    the spin density = the superposition with coefficients magnetic_moment / n_elec_valence (zero without moments)."""
    model = basis.model
    rho_tot = _gaussian_superposition(basis)
    mm = [float(np.asarray(m, dtype=float).reshape(-1)[-1]) for m in magnetic_moments]
    if model.n_spin_components == 1:
        if any(m != 0 for m in mm):
            raise ValueError("Initial magnetic moments can only be used with collinear models.")
        rho = rho_tot
    else:
        if not mm or all(m == 0 for m in mm):
            import warnings
            # density_methods.jl:135-139 (@warn)
            warnings.warn("Returning zero spin density guess, because no initial magnetization has been specified in any "
                          "of the given elements / atoms. Your SCF will likely not converge to a spin-broken solution.",
                          stacklevel=2)
            rho_spin = torch.zeros_like(rho_tot)
        else:
            if len(mm) != len(model.atoms):
                raise ValueError("one magnetic moment per atom")
            for m, a in zip(mm, model.atoms):
                if m > a.charge_ionic:
                    raise ValueError(f"Magnetic moment {m} too large for {a.symbol} with {a.charge_ionic} valence electrons")
            rho_spin = _gaussian_superposition(basis, [m / a.charge_ionic for m, a in zip(mm, model.atoms)])
        rho = torch.stack([(rho_tot + rho_spin) / 2, (rho_tot - rho_spin) / 2])
    N = float(rho.sum().item()) * model.unit_cell_volume / basis.N
    return rho * (model.n_electrons / N) if N > 0 else rho


def total_density(rho):
    return rho if rho.dim() == 3 else rho.sum(dim=0)                 # densities.jl:149


def spin_density(rho):
    return torch.zeros_like(rho) if rho.dim() == 3 else rho[0] - rho[1]   # densities.jl:150-156


# ---------------------------------------------------------------------------------- containers
class Terms:
    """What ``basis.terms`` holds after PlaneWaveBasis.jl:256-259."""


def instantiate_terms(basis):
    model = basis.model
    T = Terms()
    T.names = list(model.term_types)
    T.kinetic = [k.kinetic_local for k in basis.kpoints] if "Kinetic" in T.names else None
    T.P, T.D = None, None
    if "AtomicNonlocal" in T.names:
        build = build_projection_vectors_abi if _use_setup_abi(basis) else build_projection_vectors
        # (the spin-down block of a k-point applies the SAME projector matrix as its spin-up twin: built once, shared)
        n_k = getattr(basis, "n_kcoords_local", len(basis.kpoints))
        P = [build(basis, k) for k in basis.kpoints[:n_k]]
        P = P + P[:len(basis.kpoints) - n_k]
        if P and P[0] is not None:
            T.P, T.D = P, build_projection_coefficients(model)
    T.E_ewald = (energy_ewald(model.lattice, [a.charge_ionic for a in model.atoms], model.positions)
                 if "Ewald" in T.names else None)
    T.E_pspcorr = energy_psp_correction(model) if "PspCorrection" in T.names else None
    # terms that need cube FFTs can only be instantiated with the device library
    T.V_loc = T.poisson = None
    if basis.handle is not None:
        if "AtomicLocal" in T.names:
            T.V_loc = compute_local_potential(basis)
        if "Hartree" in T.names:
            T.poisson = compute_poisson_green_coeffs(basis)
        for ik, kpt in enumerate(basis.kpoints):
            if T.P is not None:
                Dh = np.asfortranarray(T.D)
                kpt._keep["D"] = Dh
                basis.pre_call()
                _lib.check(basis.lib.dftk_mi_kblock_set_projectors(kpt.handle, T.P[ik].shape[0],
                                                                   T.P[ik].data_ptr(), kpt.n_loc, Dh.ctypes.data))
    return T


def _band_nonlocal_energy(Ppsi, D):
    """Re sum_ij conj(p_i) D_ij p_j per band, using only the non-zero (banded) entries of D."""
    n_p = D.shape[0]
    bw = 0
    nz = np.nonzero(D)
    if len(nz[0]):
        bw = int(np.max(np.abs(nz[0] - nz[1])))
    out = torch.zeros(Ppsi.shape[0], dtype=torch.float64, device=Ppsi.device)
    for off in range(-bw, bw + 1):
        i0, i1 = max(0, -off), min(n_p, n_p - off)
        d = torch.as_tensor(np.ascontiguousarray(np.diagonal(D, offset=off)), dtype=torch.float64, device=Ppsi.device)
        out += ((torch.conj(Ppsi[:, i0:i1]) * Ppsi[:, i0 + off:i1 + off]).real * d[None, :]).sum(dim=1)
    return out


def _PH_psi(basis, Pt, psik):
    """P' psi through the library's f64-MFMA zgemm; returns a (n_bands, n_p) tensor.  With plane-wave
    sharding P and psi are row slabs: the partial projections are summed over ``comm_pw``."""
    n_p, n_G = Pt.shape
    nb = psik.shape[0]
    out = torch.empty((nb, n_p), dtype=torch.complex128, device=basis.device)
    basis.pre_call()
    _lib.check(basis.lib.dftk_mi_zgemm(basis.handle, b"C", n_p, nb, n_G, _lib.cplx(1.0), Pt.data_ptr(), Pt.stride(0),
                                       psik.data_ptr(), psik.stride(0), _lib.cplx(0.0), out.data_ptr(), n_p))
    if basis.comm_pw.size > 1:
        basis.comm_pw.sum_(torch.view_as_real(out).reshape(-1), basis.stream_ptr)
    basis.post_call()
    return out


_LDA_BITS = {"lda_x": 1, "lda_c_vwn": 2, "lda_c_pw": 4, "lda_xc_teter93": 32}
_SPIN_LDA = ("lda_x", "lda_c_pw", "lda_xc_teter93")      # functionals with a spin-polarised closed form in the library
_GGA_BITS = {"gga_x_pbe": 8, "gga_c_pbe": 16}


def local_potential_fused(basis, rho, want_potential=True, want_energies=True):
    """Hartree + XC (LDA point-wise; PBE with its gradient / divergence Fourier passes) + V_loc summed into one
    potential and their three energies by ONE library call (``dftk_mi_local_potential`` / ``_gga``: hartree.jl:50-59,
    xc.jl:84-160,356-409,576-584, local.jl:15-16, operators.jl:213-222).  Returns None for functionals the library does
    not evaluate.  ``want_energies=False`` (potential only): the call does not synchronise, the three energies are NaN."""
    import ctypes as C
    T = basis.terms
    mask = 0
    if "Xc" in T.names:
        for name in basis.model.functionals:
            if name in _LDA_BITS:
                mask |= _LDA_BITS[name]
            elif name in _GGA_BITS:
                mask |= _GGA_BITS[name]
            else:
                return None
    rho = rho.to(torch.float64).contiguous()
    V = torch.empty_like(rho) if want_potential else None
    vloc = T.V_loc if "AtomicLocal" in T.names else None
    green = T.poisson if "Hartree" in T.names else None
    E3 = (C.c_double * 3)() if want_energies or not want_potential else None
    basis.pre_call()
    args = (vloc.data_ptr() if vloc is not None else None, green.data_ptr() if green is not None else None)
    if rho.dim() == 4:
        # collinear spin: (rho_up, rho_down) -> (V_up, V_down); Hartree and the local term see the total density
        if "Xc" in T.names and any(f not in _SPIN_LDA for f in basis.model.functionals):
            raise NotImplementedError(f"collinear spin: spin-polarised forms exist for {_SPIN_LDA} only, got "
                                      f"{basis.model.functionals}")
        _lib.check(basis.lib.dftk_mi_local_potential_collinear(basis._cube_handle, rho.data_ptr(), *args, mask,
                                                               V.data_ptr() if V is not None else None, E3))
    elif mask & 24:
        Bh = np.asfortranarray(basis.model.recip_lattice, dtype=np.float64)
        _lib.check(basis.lib.dftk_mi_local_potential_gga(basis._cube_handle, Bh.ctypes.data, rho.data_ptr(), *args, mask,
                                                         _DENSITY_THRESHOLD, V.data_ptr() if V is not None else None, E3))
    else:
        _lib.check(basis.lib.dftk_mi_local_potential(basis._cube_handle, rho.data_ptr(), *args, mask,
                                                     V.data_ptr() if V is not None else None, E3))
    if E3 is None:
        return dict(Hartree=math.nan, Xc=math.nan, AtomicLocal=math.nan, V=V)
    return dict(Hartree=E3[0], Xc=E3[1], AtomicLocal=E3[2], V=V)


class Energies(dict):
    @property
    def total(self):
        return float(sum(self.values()))


def smearing_entropy(kind, x):
    """Smearing.entropy (Smearing.jl:47,84-93,114) on host arrays: s(x) with s' = x f'."""
    x = np.asarray(x, dtype=float)
    if kind == "none":
        return np.zeros_like(x)
    if kind == "fermi_dirac":
        y = np.exp(-np.abs(x))
        f = np.where(x > 0, y / (1 + y), 1 / (1 + y))
        eps = np.finfo(float).eps
        safe = (np.abs(f) >= eps) & (np.abs(1 - f) >= eps)
        fs = np.where(safe, f, 0.5)
        return np.where(safe, -(fs * np.log(fs) + (1 - fs) * np.log(1 - fs)), 0.0)
    if kind == "gaussian":
        return np.exp(-x * x) / (2 * math.sqrt(math.pi))
    raise NotImplementedError(f"smearing {kind}")


def energy_hamiltonian(basis, psi, occupation, rho=None, only_energies=False, eigenvalues=None, eF=None,
                       ritz_potential=None, ritz_occupation_threshold=0.0, only_hamiltonian=False, ritz_potential_dot=None):
    """``energy_hamiltonian(basis, psi, occupation; rho, eigenvalues, eF)`` (Hamiltonian.jl:200-227); with
    ``only_energies`` it is ``energy(...)`` (:232-236).  Returns (Energies, [DftHamiltonianBlock]).  The entropy
    term -TS (terms/entropy.jl:11-42) needs this rank's eigenvalues and the Fermi level, else it is Inf.
    ``only_hamiltonian`` (the SCF stepper's first call of a step, whose energies nobody reads): the local-potential pipeline
    returns no energies and does not synchronise; Hartree / Xc / AtomicLocal are NaN in the returned Energies."""
    basis._require_gpu()
    T = basis.terms
    E = Energies()
    pot = None
    have_psi = psi is not None and occupation is not None
    reduce_kpts = []       # terms that are sums over this rank's k-points: ONE fused reduction at the end
    ritz_fix = None
    kin_bands = []          # per k-point: kinetic energy of every band (library reduction), or None
    # local-potential pipeline behind the C ABI (LDA and PBE); DFTK_MI_TORCH_LOCAL=1 keeps the torch formulation
    # (the parity twin of tests/test_gpu_scf.py)
    fused = None
    if rho is not None and os.environ.get("DFTK_MI_TORCH_LOCAL") is None and \
            any(n in T.names for n in ("AtomicLocal", "Hartree", "Xc")):
        fused = local_potential_fused(basis, rho, want_potential=not only_energies,
                                      want_energies=not (only_hamiltonian and not only_energies))
        if fused is not None and not only_energies:
            pot = fused["V"]
    if rho is not None and rho.dim() == 4 and fused is None and any(n in T.names for n in ("Hartree", "Xc")):
        raise NotImplementedError("collinear spin needs the library's local-potential pipeline "
                                  "(dftk_mi_local_potential_collinear); the torch twin is spin-unpolarised")
    for name in T.names:
        if fused is not None and name in ("AtomicLocal", "Hartree", "Xc"):
            E[name] = fused[name]
            continue
        if name == "Kinetic":
            if have_psi:
                e = 0.0
                kin_multi = None
                if (getattr(basis, "kbatch", False) and basis.n_lanes == 1 and len(psi) > 1 and basis.comm_pw.size == 1
                        and os.environ.get("DFTK_MI_TORCH_LOCAL") is None and all(p_.stride(1) == 1 for p_ in psi)):
                    # many small k-blocks: the band-wise kinetic energies of all of them in ONE library call
                    import ctypes as C
                    n = len(psi)
                    nbs = [int(p_.shape[0]) for p_ in psi]
                    out = np.zeros(sum(nbs))
                    kbs = (C.c_void_p * n)(*[k_.handle.value for k_ in basis.kpoints])
                    basis.pre_call()
                    _lib.check(basis.lib.dftk_mi_band_kinetic_multi(
                        n, kbs, (C.c_int * n)(*nbs), (C.c_void_p * n)(*[p_.data_ptr() for p_ in psi]),
                        (C.c_int64 * n)(*[p_.stride(0) for p_ in psi]), out.ctypes.data))
                    kin_multi = np.split(out, np.cumsum(nbs)[:-1])
                for ik, psik in enumerate(psi):
                    if kin_multi is not None:
                        mk = kin_multi[ik]
                        e += basis.kweights[ik] * float(np.dot(np.asarray(occupation[ik], dtype=float), mk))
                        kin_bands.append(mk)
                        continue
                    kpt = basis.kpoints[ik]
                    if basis.comm_pw.size == 1 and psik.stride(1) == 1 and os.environ.get("DFTK_MI_TORCH_LOCAL") is None:
                        # sum_G kin_G |psi_Gn|^2 per band: the library's one-pass column reduction (the kernel of
                        # precondprep!, preconditioners.jl:75-77) instead of three cube-sized torch temporaries
                        mk = np.zeros(psik.shape[0])
                        basis.pre_call()
                        _lib.check(basis.lib.dftk_mi_tpa_precondprep(kpt.handle, psik.shape[0], psik.data_ptr(),
                                                                     psik.stride(0), mk.ctypes.data))
                        e += basis.kweights[ik] * float(np.dot(np.asarray(occupation[ik], dtype=float), mk))
                        kin_bands.append(mk)
                        continue
                    kin_bands.append(None)
                    dots = ((psik.real ** 2 + psik.imag ** 2) * T.kinetic[ik][None, :]).sum(dim=1)   # (n_bands,)
                    occ = torch.as_tensor(occupation[ik], dtype=torch.float64, device=basis.device)
                    e += basis.kweights[ik] * float((occ * dots).sum().item())
                E[name] = basis.comm_pw.sum_scalar(e)     # slab partial sums; k-points are summed below
                reduce_kpts.append(name)
            else:
                E[name] = math.inf
        elif name == "AtomicLocal":
            if not only_energies:
                pot = T.V_loc.clone() if pot is None else pot + T.V_loc
            E[name] = float((rho * T.V_loc).sum().item() * basis.dvol) if rho is not None else math.inf
        elif name == "AtomicNonlocal":
            if T.P is None:
                E[name] = 0.0
            elif (have_psi and ritz_potential is not None and eigenvalues is not None and "Kinetic" in E
                  and (basis.model.temperature == 0 or (kin_bands and all(kb_ is not None for kb_ in kin_bands)))):
                # psi are the Ritz vectors of H[V_in] with Ritz values eps_n = <psi_n|H|psi_n> (kinetic + local +
                # nonlocal): sum_n f_n <psi_n|V_nl|psi_n> = sum f eps - sum f kin_n - int V_in rho[psi].  Exact up to the
                # round-off LOBPCG carries in A X (~1e-13 relative); saves the n_p x n_bands x n_G projection GEMM of
                # nonlocal.jl:38-44 on every SCF step.  (finalize() and callers without Ritz data take the GEMM.)
                # rho[psi] was built by compute_density, which DROPS bands with |f| < occupation_threshold
                # (densities.jl:25-33): the two band sums use the same mask, so that the identity holds term by term;
                # what is left out is the nonlocal energy of the dropped bands, < threshold * n_dropped * |<V_nl>| --
                # nothing at T = 0, where occupations are exactly 0 or 2.  With symmetries rho is the SYMMETRISED density of
                # the irreducible k-points: int V_in rho_sym = int V_in rho_irr needs V_in invariant under basis.symmetries,
                # which holds by construction (rho_in is the symmetric guess or a mix of symmetrised outputs, the atomic
                # potentials carry the crystal symmetry; both SCF drivers only ever pass that V_in as ritz_potential).
                # ONE decision for all k-points: either every band-wise kinetic energy is subtracted here, or the total
                # E_kin is subtracted afterwards (ritz_fix[1]) -- a mixed list must not do both
                use_bandwise = bool(kin_bands) and all(kb_ is not None for kb_ in kin_bands)
                e = 0.0
                for ik, psik in enumerate(psi):
                    occ = np.asarray(occupation[ik], dtype=float)
                    occ = np.where(np.abs(occ) >= ritz_occupation_threshold, occ, 0.0)
                    e_k = float(np.dot(occ, np.asarray(eigenvalues[ik], dtype=float)[:len(occ)]))
                    if use_bandwise:
                        e_k -= float(np.dot(occ, kin_bands[ik]))
                    e += basis.kweights[ik] * e_k
                E[name] = e
                reduce_kpts.append(name)
                # (collinear: rho and ritz_potential both carry the spin index -- sum_s int V_s rho_s)
                # (ritz_potential_dot: sum_i V_in[i] rho[i] when the caller has it already -- dftk_mi_step_sums)
                vdot = float((rho * ritz_potential).sum().item()) if ritz_potential_dot is None else float(ritz_potential_dot)
                ritz_fix = (vdot * basis.dvol, use_bandwise)
            elif have_psi:
                e = 0.0
                for ik, psik in enumerate(psi):
                    Ppsi = _PH_psi(basis, T.P[ik], psik)           # (n_bands, n_p) rows = (P' psi)[:, band]
                    band = _band_nonlocal_energy(Ppsi, T.D)
                    occ = torch.as_tensor(occupation[ik], dtype=torch.float64, device=basis.device)
                    e += basis.kweights[ik] * float((band * occ).sum().item())
                E[name] = e
                reduce_kpts.append(name)
            else:
                E[name] = math.inf
        elif name == "Ewald":
            E[name] = T.E_ewald
        elif name == "PspCorrection":
            E[name] = T.E_pspcorr
        elif name == "Hartree":
            rho_G = basis.fft(rho)
            pot_G = T.poisson * rho_G
            if not only_energies:
                vh = basis.irfft(pot_G)
                pot = vh if pot is None else pot + vh
            E[name] = float(torch.vdot(pot_G.reshape(-1), rho_G.reshape(-1)).real.item()) / 2
        elif name == "Xc":
            exc, vxc = xc_energy_potential(basis, rho)
            if not only_energies:
                pot = vxc if pot is None else pot + vxc
            E[name] = exc
        elif name == "Entropy":
            model = basis.model
            if model.temperature == 0:
                E[name] = 0.0
            elif not have_psi or eigenvalues is None or eF is None:
                E[name] = math.inf
            else:
                e = 0.0
                for ik, psik in enumerate(psi):
                    x = (np.asarray(eigenvalues[ik], dtype=float)[:psik.shape[0]] - eF) / model.temperature
                    e -= (model.temperature * basis.kweights[ik] * model.filled_occupation
                          * float(np.sum(smearing_entropy(model.smearing, x))))
                E[name] = e
                reduce_kpts.append(name)
        else:
            raise NotImplementedError(f"term {name} is outside the MI355X hot path")
    if reduce_kpts and basis.comm_kpts.size > 1:
        for name, v in zip(reduce_kpts, basis.comm_kpts.sum_scalars([E[n] for n in reduce_kpts])):
            E[name] = v
    if ritz_fix is not None:
        # (band-wise kinetic energies already subtracted with the occupation mask, else the total E_kin: T = 0 only)
        E["AtomicNonlocal"] = E["AtomicNonlocal"] - (0.0 if ritz_fix[1] else E["Kinetic"]) - ritz_fix[0]
    if only_energies:
        return E, None
    ham = DftHamiltonianBlock.for_all_kpoints(basis, pot)
    return E, ham
