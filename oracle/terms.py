"""Energy terms, Hamiltonian blocks and guess density (oracle restatement).  Test infrastructure only.

Restates: ``src/terms/kinetic.jl:31-57``; ``src/terms/local.jl:9-23,75-138``;
``src/terms/nonlocal.jl:31-47,107-141,166-244``; ``src/terms/hartree.jl:29-59``;
``src/terms/xc.jl:84-160`` with closed-form LDA functionals (Slater exchange, VWN5 and
PW92 correlation -- the arithmetic the reference delegates to Libxc / DftFunctionals.jl);
``src/terms/ewald.jl:40-168``; ``src/terms/psp_correction.jl:26-32``;
``src/density_methods.jl:111-125,158-181,236-244,286-322``;
``src/terms/operators.jl:76-128,213-222``; ``src/terms/Hamiltonian.jl:36-57,137-236``.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import erfc

from .psp import (eval_psp_local_fourier, eval_psp_projector_fourier,
                  eval_psp_energy_correction, solid_harmonic_real)


# ----------------------------------------------------------------------------- setup of terms
def kinetic_energies(basis, kpt):
    """1/2 |k+G|^2 (kinetic.jl:31-35)."""
    p = basis.Gplusk_vectors_cart(kpt)
    return np.sum(p * p, axis=1) / 2


def compute_local_potential(basis):
    """V_loc(r) = irfft( sum_atoms e^{-2 pi i G.r} V_loc(|G|) / sqrt(Omega) ) (local.jl:108-138)."""
    model = basis.model
    gx, gy, gz = basis.G_vectors_cube()
    Gcart = basis.G_vectors_cart_cube()
    Gnorm = np.sqrt(np.sum(Gcart * Gcart, axis=-1))
    pot = np.zeros(Gnorm.shape, dtype=complex)
    for group in model.atom_groups:
        ff = eval_psp_local_fourier(model.atoms[group[0]].psp, Gnorm.ravel()).reshape(Gnorm.shape)
        for ia in group:
            r = model.positions[ia]
            phase = np.exp(-2j * math.pi * (gx * r[0] + gy * r[1] + gz * r[2]))
            pot += phase * ff / math.sqrt(model.unit_cell_volume)
    pot = basis.enforce_real(pot)
    return basis.irfft_cube(pot)


def build_projector_form_factors(psp, Gpk_cart):
    """Form factors of all projectors of one atom at 0 (nonlocal.jl:205-244).
    Column order (l, m, i): offset_l + n_proj_l*(m+l) + i."""
    n_G = Gpk_cart.shape[0]
    pnorm = np.sqrt(np.sum(Gpk_cart * Gpk_cart, axis=1))
    out = np.zeros((n_G, psp.count_n_proj()), dtype=complex)
    for l in range(psp.lmax + 1):
        n_proj_l = psp.count_n_proj_radial(l)
        off_l = sum(psp.count_n_proj(ll) for ll in range(l))
        for i in range(1, n_proj_l + 1):
            radial = eval_psp_projector_fourier(psp, i, l, pnorm)
            for m in range(-l, l + 1):
                col = off_l + n_proj_l * (m + l) + (i - 1)
                out[:, col] = radial * ((-1j) ** l) * solid_harmonic_real(l, m, Gpk_cart)
    return out


def build_projection_vectors(basis, kpt):
    """P[:, (atom, l, m, i)] = e^{-2 pi i (G+k).r} formfactor / sqrt(Omega) (nonlocal.jl:166-199)."""
    model = basis.model
    Gpk = kpt.G_vectors + kpt.coordinate[None, :]
    Gpk_cart = basis.Gplusk_vectors_cart(kpt)
    cols = []
    for group in model.atom_groups:
        psp = model.atoms[group[0]].psp
        ff = build_projector_form_factors(psp, Gpk_cart)
        for ia in group:
            sf = np.exp(-2j * math.pi * (Gpk @ model.positions[ia]))
            cols.append(sf[:, None] * ff / math.sqrt(model.unit_cell_volume))
    if not cols:
        return np.zeros((len(kpt.mapping), 0), dtype=complex)
    return np.concatenate(cols, axis=1)


def build_projection_coefficients_psp(psp):
    """Per-atom block-diagonal D, order (l, m, i) (nonlocal.jl:130-141)."""
    n = psp.count_n_proj()
    D = np.zeros((n, n))
    count = 0
    for l in range(psp.lmax + 1):
        for _m in range(-l, l + 1):
            nl = psp.count_n_proj_radial(l)
            D[count:count + nl, count:count + nl] = psp.h[l]
            count += nl
    return D


def build_projection_coefficients(basis):
    """nonlocal.jl:107-124."""
    model = basis.model
    blocks = []
    for group in model.atom_groups:
        Dp = build_projection_coefficients_psp(model.atoms[group[0]].psp)
        blocks += [Dp] * len(group)
    n = sum(b.shape[0] for b in blocks)
    D = np.zeros((n, n))
    c = 0
    for b in blocks:
        D[c:c + b.shape[0], c:c + b.shape[0]] = b
        c += b.shape[0]
    return D


def compute_poisson_green_coeffs(basis):
    """4 pi / |G|^2, zero DC, enforce_real (hartree.jl:29-45)."""
    Gcart = basis.G_vectors_cart_cube()
    G2 = np.sum(Gcart * Gcart, axis=-1)
    with np.errstate(divide="ignore"):
        coeffs = 4 * math.pi / G2
    coeffs[0, 0, 0] = 0.0
    return np.real(basis.enforce_real(coeffs))


def default_eta(lattice):
    """ewald.jl:40-44."""
    from .basis import compute_recip_lattice
    recip = compute_recip_lattice(lattice)
    return math.sqrt(math.sqrt(1.69 * np.linalg.norm(recip / (2 * math.pi)) / np.linalg.norm(lattice))) / 2


def energy_ewald(lattice, charges, positions, eta=None):
    """Ewald energy per cell with neutralising background (ewald.jl:64-168, energy only)."""
    from .basis import (compute_recip_lattice, compute_unit_cell_volume,
                        estimate_integer_lattice_bounds)
    lattice = np.asarray(lattice, dtype=float)
    charges = np.asarray(charges, dtype=float)
    pos = np.asarray(positions, dtype=float).reshape(-1, 3)
    if len(charges) == 0:
        return 0.0
    if eta is None:
        eta = default_eta(lattice)
    eps = np.finfo(float).eps
    max_exp_arg = -math.log(eps) + 5
    max_erfc_arg = math.sqrt(max_exp_arg)
    recip = compute_recip_lattice(lattice)
    Glims = estimate_integer_lattice_bounds(recip, math.sqrt(max_exp_arg) * 2 * eta)
    poslims = [float(np.max(pos[:, i][:, None] - pos[:, i][None, :])) for i in range(3)]
    Rlims = estimate_integer_lattice_bounds(lattice, max_erfc_arg / eta, poslims)

    # reciprocal sum
    sum_recip = -(np.sum(charges) ** 2 / (4 * eta ** 2))
    g1 = np.arange(-Glims[0], Glims[0] + 1)
    g2 = np.arange(-Glims[1], Glims[1] + 1)
    g3 = np.arange(-Glims[2], Glims[2] + 1)
    G = np.stack(np.meshgrid(g1, g2, g3, indexing="ij"), axis=-1).reshape(-1, 3)
    G = G[np.any(G != 0, axis=1)]
    Gsq = np.sum((G @ recip.T) ** 2, axis=1)
    keep = Gsq / (4 * eta ** 2) < max_exp_arg + 50   # everything else underflows to zero
    G, Gsq = G[keep], Gsq[keep]
    # structure factors in chunks to bound memory for large cells
    for c0 in range(0, len(G), 65536):
        Gc, Gs = G[c0:c0 + 65536], Gsq[c0:c0 + 65536]
        ph = 2 * math.pi * (Gc @ pos.T)
        cs = np.cos(ph) @ charges
        sn = np.sin(ph) @ charges
        sum_recip += np.sum((cs * cs + sn * sn) * np.exp(-Gs / (4 * eta ** 2)) / Gs)
    sum_recip *= 4 * math.pi / compute_unit_cell_volume(lattice)

    # real-space sum
    sum_real = -2 * eta / math.sqrt(math.pi) * np.sum(charges ** 2)
    r1 = np.arange(-Rlims[0], Rlims[0] + 1)
    r2 = np.arange(-Rlims[1], Rlims[1] + 1)
    r3 = np.arange(-Rlims[2], Rlims[2] + 1)
    R = np.stack(np.meshgrid(r1, r2, r3, indexing="ij"), axis=-1).reshape(-1, 3).astype(float)
    qq = charges[:, None] * charges[None, :]
    n = len(charges)
    for Rv in R:
        d = pos[:, None, :] - (pos[None, :, :] + Rv[None, None, :])      # ti - tj
        dist = np.linalg.norm(d @ lattice.T, axis=-1)
        if not np.any(Rv):
            dist = dist + np.where(np.eye(n, dtype=bool), np.inf, 0.0)   # skip self interaction
        mask = dist * eta < max_erfc_arg + 10
        if np.any(mask):
            sum_real += np.sum(qq[mask] * erfc(eta * dist[mask]) / dist[mask])
    return (sum_recip + sum_real) / 2


def energy_psp_correction(model):
    """psp_correction.jl:26-32."""
    corr = sum(len(g) * eval_psp_energy_correction(model.atoms[g[0]].psp) for g in model.atom_groups)
    n_el = sum(a.charge_ionic for a in model.atoms)   # n_electrons_from_atoms
    return corr * n_el / model.unit_cell_volume


# ----------------------------------------------------------------------------- XC (closed forms)
def _lda_x(rho):
    """Slater exchange (libxc ``lda_x``): e = -3/4 (3/pi)^{1/3} rho^{4/3}."""
    cx = -0.75 * (3 / math.pi) ** (1 / 3)
    r13 = np.cbrt(rho)
    return cx * rho * r13, (4 / 3) * cx * r13


def _lda_c_vwn(rho):
    """VWN5 paramagnetic correlation (libxc ``lda_c_vwn``; Vosko, Wilk, Nusair 1980)."""
    A, b, c, x0 = 0.0310907, 3.72744, 12.9352, -0.10498
    rs = np.cbrt(3 / (4 * math.pi * rho))
    x = np.sqrt(rs)
    X = x * x + b * x + c
    X0 = x0 * x0 + b * x0 + c
    Q = math.sqrt(4 * c - b * b)
    at = np.arctan(Q / (2 * x + b))
    eps = A * (np.log(x * x / X) + 2 * b / Q * at
               - b * x0 / X0 * (np.log((x - x0) ** 2 / X) + 2 * (b + 2 * x0) / Q * at))
    datdx = -2 * Q / (Q * Q + (2 * x + b) ** 2)
    deps_dx = A * (2 / x - (2 * x + b) / X + 2 * b / Q * datdx
                   - b * x0 / X0 * (2 / (x - x0) - (2 * x + b) / X + 2 * (b + 2 * x0) / Q * datdx))
    deps_drs = deps_dx / (2 * x)
    v = eps - rs / 3 * deps_drs
    return rho * eps, v


def _lda_c_pw(rho):
    """Perdew-Wang 1992 unpolarised correlation (libxc ``lda_c_pw``, original parameters).
    PARITY UNPINNED: the reference tests hold no numeric value for this functional."""
    a, a1, b1, b2, b3, b4 = 0.031091, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294
    rs = np.cbrt(3 / (4 * math.pi * rho))
    sq = np.sqrt(rs)
    den = 2 * a * (b1 * sq + b2 * rs + b3 * rs * sq + b4 * rs * rs)
    lg = np.log1p(1 / den)
    eps = -2 * a * (1 + a1 * rs) * lg
    dden = 2 * a * (b1 / (2 * sq) + b2 + 1.5 * b3 * sq + 2 * b4 * rs)
    deps = -2 * a * a1 * lg + 2 * a * (1 + a1 * rs) * dden / (den * den + den)
    v = eps - rs / 3 * deps
    return rho * eps, v


_TETER_A = (0.4581652932831429, 2.217058676663745, 0.7405551735357053, 0.01968227878617998)
_TETER_DA = (0.119086804055547, 0.6157402568883345, 0.1574201515892867, 0.003532336663397157)
_TETER_B = (1.0, 4.504130959426697, 1.110667363742916, 0.02359291751427506)
_TETER_DB = (0.0, 0.2673612973836267, 0.2052004607777787, 0.004200005045691381)


def _teter_eps(rs, fz):
    """Goedecker, Teter, Hutter 1996 (PRB 54, 1703) Pade fit of the LDA XC energy per particle (libxc
    ``lda_xc_teter93``): -(sum a_i rs^i) / (sum b_i rs^(i+1)) with a_i, b_i linear in the spin interpolation f(zeta)."""
    a = [_TETER_A[i] + _TETER_DA[i] * fz for i in range(4)]
    b = [_TETER_B[i] + _TETER_DB[i] * fz for i in range(4)]
    num = a[0] + rs * (a[1] + rs * (a[2] + rs * a[3]))
    den = rs * (b[0] + rs * (b[1] + rs * (b[2] + rs * b[3])))
    return -num / den


def _lda_xc_teter93(rho):
    """Unpolarised ``lda_xc_teter93``: (e, v) with v = d(rho eps)/d rho."""
    rs = np.cbrt(3 / (4 * math.pi * rho))
    eps = _teter_eps(rs, 0.0)
    h = 1e-30
    deps = np.imag(_teter_eps(rs + 1j * h, 0.0)) / h
    return rho * eps, eps - rs / 3 * deps


_FUNCTIONALS = {"lda_x": _lda_x, "lda_c_vwn": _lda_c_vwn, "lda_c_pw": _lda_c_pw, "lda_xc_teter93": _lda_xc_teter93}


# ---- collinear spin (LDA): energy densities e(rho_up, rho_down) written with analytic primitives only; the potentials
# v_s = de/d rho_s come from the complex-step method (as for the GGAs below: no hand-derived formulas to get wrong)
def _zeta_terms(ra, rb):
    rt = ra + rb
    xa, xb = 2 * ra / rt, 2 * rb / rt                 # 1 + zeta, 1 - zeta
    p43 = np.exp(4.0 / 3.0 * np.log(xa)) + np.exp(4.0 / 3.0 * np.log(xb))
    fz = (p43 - 2) / (2 ** (4.0 / 3.0) - 2)           # f(zeta), f(0) = 0, f(+-1) = 1
    rs = np.exp(-np.log(4 * math.pi * rt / 3) / 3)
    return rt, (ra - rb) / rt, fz, rs


def _lda_x_spin_e(ra, rb):
    """Spin-scaling relation E_x[ra, rb] = (E_x[2 ra] + E_x[2 rb]) / 2 (libxc ``lda_x`` polarised)."""
    cx = -0.75 * (3 / math.pi) ** (1 / 3)
    return cx * 2 ** (1.0 / 3.0) * (np.exp(4.0 / 3.0 * np.log(ra)) + np.exp(4.0 / 3.0 * np.log(rb)))


def _pw92_G(rs, A, a1, b1, b2, b3, b4):
    sq = np.sqrt(rs)
    return -2 * A * (1 + a1 * rs) * np.log(1 + 1 / (2 * A * (b1 * sq + b2 * rs + b3 * rs * sq + b4 * rs * rs)))


def _lda_c_pw_spin_e(ra, rb):
    """Perdew-Wang 1992 with the spin interpolation of their eq. (8) (libxc ``lda_c_pw``, original parameters,
    f''(0) = 1.709921).  PARITY UNPINNED like the unpolarised form."""
    rt, zeta, fz, rs = _zeta_terms(ra, rb)
    e0 = _pw92_G(rs, 0.031091, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294)
    e1 = _pw92_G(rs, 0.015545, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517)
    mac = _pw92_G(rs, 0.016887, 0.11125, 10.357, 3.6231, 0.88026, 0.49671)     # = -alpha_c
    z4 = zeta ** 4
    return rt * (e0 - mac * fz / 1.709921 * (1 - z4) + (e1 - e0) * fz * z4)


def _lda_xc_teter93_spin_e(ra, rb):
    rt, _, fz, rs = _zeta_terms(ra, rb)
    return rt * _teter_eps(rs, fz)


_SPIN_FUNCTIONALS = {"lda_x": _lda_x_spin_e, "lda_c_pw": _lda_c_pw_spin_e, "lda_xc_teter93": _lda_xc_teter93_spin_e}
_SPIN_FLOOR = 1e-20      # a spin channel is never evaluated below this density (log / fractional powers)


def xc_energy_potential_spin(basis, rho):
    """Collinear LDA: rho has shape (2, nz, ny, nx) = (up, down); returns E_xc and the potential per spin channel
    (xc.jl:84-160 with n_spin = 2).  Gradient-corrected functionals with spin are not restated."""
    missing = [f for f in basis.model.functionals if f not in _SPIN_FUNCTIONALS]
    if missing:
        raise NotImplementedError(f"collinear spin: no spin-polarised restatement of {missing}")
    ra = np.maximum(rho[0], _SPIN_FLOOR)
    rb = np.maximum(rho[1], _SPIN_FLOOR)
    h = 1e-30
    e = np.zeros_like(ra)
    v = np.zeros_like(rho)
    for name in basis.model.functionals:
        fun = _SPIN_FUNCTIONALS[name]
        e += fun(ra, rb)
        v[0] += np.imag(fun(ra + 1j * h, rb.astype(complex))) / h
        v[1] += np.imag(fun(ra.astype(complex), rb + 1j * h)) / h
    empty = (rho[0] + rho[1]) <= 2 * _SPIN_FLOOR
    if np.any(empty):
        e[empty] = 0.0
        v[:, empty] = 0.0
    return float(np.sum(e) * basis.dvol), v


# GGA functionals: energy density per volume e(rho, sigma), sigma = |grad rho|^2.  Written with analytic
# primitives only (no abs / cbrt) so that the derivatives de/drho, de/dsigma can be taken by the
# complex-step method to machine precision (no hand-derived formulas to get wrong).
def _gga_x_pbe_e(rho, sigma):
    """PBE exchange (libxc ``gga_x_pbe``; Perdew, Burke, Ernzerhof 1996): e_x^LDA F_x(s),
    F_x = 1 + kappa - kappa^2 / (kappa + mu s^2), s = |grad rho| / (2 k_F rho)."""
    kappa, mu = 0.8040, 0.2195149727645171
    cx = -0.75 * (3 / math.pi) ** (1 / 3)
    kf = (3 * math.pi ** 2 * rho) ** (1 / 3)
    s2 = sigma / (4 * kf * kf * rho * rho)
    return cx * rho ** (4 / 3) * (1 + kappa - kappa * kappa / (kappa + mu * s2))


def _gga_c_pbe_e(rho, sigma):
    """PBE correlation (libxc ``gga_c_pbe``): rho (eps_c^PW92mod(rs) + H(rs, t)), unpolarised (phi = 1),
    H = gamma ln(1 + beta/gamma t^2 (1 + A t^2) / (1 + A t^2 + A^2 t^4)); libxc builds it on
    ``lda_c_pw_mod`` (a = 0.0310907)."""
    beta, gamma = 0.06672455060314922, (1 - math.log(2)) / math.pi ** 2
    a, a1, b1, b2, b3, b4 = 0.0310907, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294
    rs = (3 / (4 * math.pi * rho)) ** (1 / 3)
    sq = np.sqrt(rs)
    eps = -2 * a * (1 + a1 * rs) * np.log1p(1 / (2 * a * (b1 * sq + b2 * rs + b3 * rs * sq + b4 * rs * rs)))
    kf = (3 * math.pi ** 2 * rho) ** (1 / 3)
    t2 = sigma * math.pi / (16 * kf * rho * rho)
    A = beta / gamma / np.expm1(-eps / gamma)
    f1 = t2 + A * t2 * t2
    H = gamma * np.log1p(beta / gamma * f1 / (1 + A * f1))
    return rho * (eps + H)


_GGA_FUNCTIONALS = {"gga_x_pbe": _gga_x_pbe_e, "gga_c_pbe": _gga_c_pbe_e}
_DENSITY_THRESHOLD = 1e-12   # below it a GGA contributes nothing (libxc-style density threshold)


def _gga_terms(fun, rho, sigma):
    """(e, de/drho, de/dsigma) by complex-step differentiation."""
    h = 1e-30
    e = fun(rho, sigma)
    vrho = np.imag(fun(rho + 1j * h, sigma.astype(complex))) / h
    vsigma = np.imag(fun(rho.astype(complex), sigma + 1j * h)) / h
    return e, vrho, vsigma


def xc_energy_potential(basis, rho):
    """E_xc = sum e dvol and V_xc = V_rho - 2 div(V_sigma grad rho) (xc.jl:84-160): LDA terms have
    V_sigma = 0; for GGAs grad rho and the divergence are taken in Fourier space on the cube
    (LibxcDensities xc.jl:356-409, divergence_real :576-584)."""
    rho_c = np.maximum(rho, 1e-300)   # guard the cube root / log; libxc uses a density threshold
    e = np.zeros_like(rho)
    v = np.zeros_like(rho)
    gga = [f for f in basis.model.functionals if f in _GGA_FUNCTIONALS]
    for name in basis.model.functionals:
        if name in _GGA_FUNCTIONALS:
            continue
        ei, vi = _FUNCTIONALS[name](rho_c)
        e += ei
        v += vi
    tiny = rho <= 1e-300
    if np.any(tiny):
        e[tiny] = 0.0
        v[tiny] = 0.0
    if gga:
        G = basis.G_vectors_cart_cube()                      # (nz, ny, nx, 3)
        rho_f = basis.fft_cube(rho)
        grad = [basis.irfft_cube(1j * G[..., a] * rho_f) for a in range(3)]
        sigma = grad[0] ** 2 + grad[1] ** 2 + grad[2] ** 2
        ok = rho > _DENSITY_THRESHOLD
        rho_s = np.where(ok, rho, 1.0)
        sig_s = np.where(ok, sigma, 0.0)
        vsig = np.zeros_like(rho)
        for name in gga:
            ei, vr, vs = _gga_terms(_GGA_FUNCTIONALS[name], rho_s, sig_s)
            e += np.where(ok, ei, 0.0)
            v += np.where(ok, vr, 0.0)
            vsig += np.where(ok, vs, 0.0)
        div = sum(1j * G[..., a] * basis.fft_cube(vsig * grad[a]) for a in range(3))
        v = v - 2.0 * basis.irfft_cube(div)
    return float(np.sum(e) * basis.dvol), v


# ----------------------------------------------------------------------------- guess density
def atom_decay_length(n_elec_core, n_elec_valence):
    """ABINIT table (density_methods.jl:286-322)."""
    n_elec_valence = int(round(n_elec_valence))
    if n_elec_valence == 0:
        return 0.0
    if n_elec_core < 0.5:
        data = [0.6, 0.4, 0.3, 0.25, 0.2]
    elif n_elec_core < 2.5:
        data = [1.8, 1.4, 1.0, 0.7, 0.6, 0.5, 0.4, 0.35, 0.3]
    elif n_elec_core < 10.5:
        data = [2.0, 1.6, 1.25, 1.1, 1.0, 0.9, 0.8, 0.7, 0.7, 0.7, 0.6]
    elif n_elec_core < 12.5:
        data = [1.9, 1.5, 1.15, 1.0, 0.9, 0.8, 0.7, 0.6, 0.6, 0.6, 0.5]
    elif n_elec_core < 18.5:
        data = [2.0, 1.8, 1.5, 1.2, 1.0, 0.9, 0.85, 0.8, 0.75, 0.7, 0.65, 0.65, 0.6]
    elif n_elec_core < 28.5:
        data = [1.5, 1.25, 1.15, 1.05, 1.00, 0.95, 0.95, 0.9, 0.9, 0.85, 0.85, 0.80, 0.8, 0.75, 0.7]
    elif n_elec_core < 36.5:
        data = [2.0, 2.00, 1.60, 1.40, 1.25, 1.10, 1.00, 0.95, 0.90, 0.85, 0.80, 0.75, 0.7]
    else:
        data = [2.0, 2.00, 1.55, 1.25, 1.15, 1.10, 1.05, 1.0, 0.95, 0.9, 0.85, 0.85, 0.8]
    return data[min(n_elec_valence, len(data)) - 1]


def _gaussian_superposition(basis, coefficients):
    """atomic_density_superposition (density_methods.jl:158-181) with the Gaussian valence densities (:236-244)."""
    model = basis.model
    gx, gy, gz = basis.G_vectors_cube()
    Gcart = basis.G_vectors_cart_cube()
    Gnorm = np.sqrt(np.sum(Gcart * Gcart, axis=-1))
    rho_G = np.zeros(Gnorm.shape, dtype=complex)
    for group in model.atom_groups:
        el = model.atoms[group[0]]
        ff = el.charge_ionic * np.exp(-(Gnorm * atom_decay_length(el.n_elec_core, el.charge_ionic)) ** 2)
        for ia in group:
            r = model.positions[ia]
            rho_G += (coefficients[ia] * np.exp(-2j * math.pi * (gx * r[0] + gy * r[1] + gz * r[2])) * ff
                      / math.sqrt(model.unit_cell_volume))
    rho_G = basis.enforce_real(rho_G)
    return basis.irfft_cube(rho_G)


def guess_density(basis, magnetic_moments=()):
    """Gaussian superposition, renormalised to n_electrons (density_methods.jl:35-38, :102-152).  For a collinear
    model the result has shape (2, nz, ny, nx) = ((tot + spin) / 2, (tot - spin) / 2), the spin density being the same
    superposition with the coefficients magnetic_moment / n_elec_valence per atom (zero without moments)."""
    model = basis.model
    rho_tot = _gaussian_superposition(basis, np.ones(len(model.atoms)))
    if model.n_spin_components == 1:
        if len(magnetic_moments) and any(m != 0 for m in np.asarray(magnetic_moments, dtype=float).reshape(-1)):
            raise ValueError("Initial magnetic moments can only be used with collinear models.")
        rho = rho_tot
    else:
        mm = [float(np.asarray(m, dtype=float).reshape(-1)[-1]) for m in magnetic_moments]
        if not mm or all(m == 0 for m in mm):
            rho_spin = np.zeros_like(rho_tot)
        else:
            if len(mm) != len(model.atoms):
                raise ValueError("one magnetic moment per atom")
            for m, a in zip(mm, model.atoms):
                if m > a.charge_ionic:
                    raise ValueError(f"Magnetic moment {m} too large for {a.symbol} with {a.charge_ionic} valence electrons")
            rho_spin = _gaussian_superposition(basis, [m / a.charge_ionic for m, a in zip(mm, model.atoms)])
        rho = np.stack([(rho_tot + rho_spin) / 2, (rho_tot - rho_spin) / 2])
    N = np.sum(rho) * model.unit_cell_volume / basis.N
    if N > 0:
        rho = rho * (model.n_electrons / N)
    return rho


def total_density(rho):
    return rho if rho.ndim == 3 else rho.sum(axis=0)                 # densities.jl:149


def spin_density(rho):
    return np.zeros_like(rho) if rho.ndim == 3 else rho[0] - rho[1]  # densities.jl:150-156


# ----------------------------------------------------------------------------- terms container
class Terms:
    pass


def instantiate_terms(basis):
    """The ``t(basis)`` loop of PlaneWaveBasis.jl:256-259 for the terms on the hot path."""
    model = basis.model
    T = Terms()
    T.names = list(model.terms)
    T.kinetic = [kinetic_energies(basis, k) for k in basis.kpoints] if "Kinetic" in T.names else None
    T.V_loc = compute_local_potential(basis) if "AtomicLocal" in T.names else None
    if "AtomicNonlocal" in T.names:
        T.P = [build_projection_vectors(basis, k) for k in basis.kpoints]
        T.D = build_projection_coefficients(basis)
        if T.D.shape[0] == 0:
            T.P, T.D = None, None
    else:
        T.P, T.D = None, None
    T.E_ewald = (energy_ewald(model.lattice, [a.charge_ionic for a in model.atoms], model.positions)
                 if "Ewald" in T.names else None)
    T.E_pspcorr = energy_psp_correction(model) if "PspCorrection" in T.names else None
    T.poisson = compute_poisson_green_coeffs(basis) if "Hartree" in T.names else None
    return T


class HamiltonianBlock:
    """DftHamiltonianBlock (Hamiltonian.jl:22-34): 1 Fourier + 1 summed real-space + <=1 nonlocal op."""

    def __init__(self, basis, kpt, kinetic, potential, P, D):
        self.basis, self.kpoint = basis, kpt
        self.kinetic = kinetic          # fourier_op.multiplier (n_G,) or None
        self.potential = potential      # local_op.potential (nz,ny,nx) real or None
        self.P, self.D = P, D           # nonlocal_op

    @property
    def n_G(self):
        return len(self.kpoint.mapping)

    def apply_local(self, psi):
        """FFT[V iFFT[psi]]/N per band (Hamiltonian.jl:152-163)."""
        out = np.zeros_like(psi)
        if self.potential is None:
            return out
        b = self.basis
        pot = self.potential * (b.fft_normalization * b.ifft_normalization)
        for n in range(psi.shape[1]):
            psi_real = b.ifft(self.kpoint, psi[:, n], normalize=False)
            out[:, n] = b.fft(self.kpoint, psi_real * pot, normalize=False)
        return out

    def apply_nonlocal(self, psi):
        """P (D (P' psi)) (operators.jl:126-128)."""
        if self.P is None:
            return np.zeros_like(psi)
        return self.P @ (self.D @ (self.P.conj().T @ psi))

    def mul(self, psi):
        """mul!(Hpsi, H, psi) (Hamiltonian.jl:137-192)."""
        Hpsi = self.apply_local(psi)
        if self.kinetic is not None:
            Hpsi = Hpsi + self.kinetic[:, None] * psi
        if self.P is not None:
            Hpsi = Hpsi + self.apply_nonlocal(psi)
        return Hpsi

    __matmul__ = mul

    def to_dense(self):
        """Matrix(H) via unit vectors (small bases only)."""
        return self.mul(np.eye(self.n_G, dtype=complex))


class Energies(dict):
    @property
    def total(self):
        return float(sum(self.values()))


def smearing_entropy(kind, x):
    """Smearing.entropy (Smearing.jl:47,84-93,114): s(x) with s' = x f'."""
    x = np.asarray(x, dtype=float)
    if kind == "none":
        return np.zeros_like(x)
    if kind == "fermi_dirac":
        f = np.where(x > 0, np.exp(-np.abs(x)) / (1 + np.exp(-np.abs(x))), 1 / (1 + np.exp(-np.abs(x))))
        eps = np.finfo(float).eps
        safe = (np.abs(f) >= eps) & (np.abs(1 - f) >= eps)
        fs = np.where(safe, f, 0.5)
        return np.where(safe, -(fs * np.log(fs) + (1 - fs) * np.log(1 - fs)), 0.0)
    if kind == "gaussian":
        return np.exp(-x * x) / (2 * math.sqrt(math.pi))
    raise NotImplementedError(kind)


def energy_hamiltonian(basis, psi, occupation, rho=None, eigenvalues=None, eF=None):
    """``energy_hamiltonian(basis, psi, occ; rho, eigenvalues, eF)`` (Hamiltonian.jl:200-227): per-term energies and
    one HamiltonianBlock per k-point; ``energy`` (:232-236) gives the same energies.  The entropy term -TS
    (terms/entropy.jl:11-42) needs the eigenvalues and the Fermi level, otherwise it is Inf as in the reference."""
    T = basis.terms
    model = basis.model
    E = Energies()
    pot = None

    def add_pot(v):
        nonlocal pot
        pot = v.copy() if pot is None else pot + v

    have_psi = psi is not None and occupation is not None
    n_spin = model.n_spin_components
    rho_spin_resolved = rho
    if rho is not None and n_spin == 2:
        # density-functional terms of the TOTAL density add the same potential to both spin channels; only Xc
        # distinguishes them (potential has shape (2, nz, ny, nx) then, ops pick potential[kpt.spin], xc.jl:163-175)
        rho = total_density(rho)
    for name in T.names:
        if name == "Kinetic":
            if have_psi:
                e = 0.0
                for ik, psik in enumerate(psi):
                    dots = np.sum(np.abs(psik) ** 2 * T.kinetic[ik][:, None], axis=0)
                    e += basis.kweights[ik] * float(np.sum(np.asarray(occupation[ik]) * dots))
                E[name] = e
            else:
                E[name] = math.inf
        elif name == "AtomicLocal":
            add_pot(T.V_loc)
            E[name] = float(np.sum(rho * T.V_loc) * basis.dvol) if rho is not None else math.inf
        elif name == "AtomicNonlocal":
            if T.P is None:
                E[name] = 0.0
            elif have_psi:
                e = 0.0
                for ik, psik in enumerate(psi):
                    Ppsi = T.P[ik].conj().T @ psik
                    band = np.sum(np.real(np.conj(Ppsi) * (T.D @ Ppsi)), axis=0)
                    e += basis.kweights[ik] * float(np.sum(band * np.asarray(occupation[ik])))
                E[name] = e
            else:
                E[name] = math.inf
        elif name == "Ewald":
            E[name] = T.E_ewald
        elif name == "PspCorrection":
            E[name] = T.E_pspcorr
        elif name == "Hartree":
            rho_G = basis.fft_cube(rho)
            pot_G = T.poisson * rho_G
            add_pot(basis.irfft_cube(pot_G))
            E[name] = float(np.real(np.vdot(pot_G, rho_G)) / 2)
        elif name == "Xc":
            if n_spin == 2:
                exc, vxc = xc_energy_potential_spin(basis, rho_spin_resolved)
                add_pot(vxc)                                     # (2, nz, ny, nx) from here on (numpy broadcasting)
            else:
                exc, vxc = xc_energy_potential(basis, rho)
                add_pot(vxc)
            E[name] = exc
        elif name == "Entropy":
            if model.temperature == 0:
                E[name] = 0.0
            elif not have_psi or eigenvalues is None or eF is None:
                E[name] = math.inf
            else:
                e = 0.0
                for ik, psik in enumerate(psi):
                    x = (np.asarray(eigenvalues[ik], dtype=float)[:psik.shape[1]] - eF) / model.temperature
                    e -= (model.temperature * basis.kweights[ik] * model.filled_occupation
                          * float(np.sum(smearing_entropy(model.smearing, x))))
                E[name] = e
        else:
            raise NotImplementedError(name)
    def pot_of(kpt):
        return pot[kpt.spin - 1] if (pot is not None and pot.ndim == 4) else pot
    ham = [HamiltonianBlock(basis, kpt,
                            T.kinetic[ik] if T.kinetic is not None else None,
                            pot_of(kpt),
                            T.P[ik] if T.P is not None else None, T.D)
           for ik, kpt in enumerate(basis.kpoints)]
    return E, ham
