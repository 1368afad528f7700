"""Real-symmetric orbitals at the Gamma point (oracle restatement of the library's EXTENSION, NumPy).
Test infrastructure only.

The reference has no Gamma special case; this file restates what ``dftk.jl_amd/csrc/gamma_kernels.hip`` does so
that the extension has a CPU oracle of its own:

* pair tables of a k = 0 sphere (one representative per {G, -G}, G = 0 first),
* the half-sphere format (row 0 = x(0), rows j > 0 = sqrt(2) x(G_j)) and its image as 2 n_half - 1 REAL unknowns,
* the Gamma-point Hamiltonian of ``oracle.terms.HamiltonianBlock`` restricted to real-symmetric vectors -- a real
  symmetric operator with the spectrum of the complex one,
* LOBPCG on it: the reference's ``LOBPCG`` restatement (``oracle.lobpcg``, src/eigen/lobpcg_hyper_impl.jl:354-582)
  run on the real unknowns with the TPA preconditioner on the doubled kinetic vector.

Pinned by tests/test_oracle_gamma_real.py: dense spectra, the complex oracle LOBPCG, the library's host pair tables.
"""
from __future__ import annotations

import numpy as np

from .lobpcg import PreconditionerTPA, lobpcg_hyper

SQRT2 = np.sqrt(2.0)


def pair_tables(fft_size, mapping):
    """Rows (g, mg) of G_j and -G_j for one representative per pair, ascending in g, G = 0 first.  Raises for a
    sphere without inversion symmetry or with a Nyquist point (its own partner)."""
    nx, ny, nz = fft_size
    m = np.asarray(mapping, dtype=np.int64)
    if len(m) < 1 or m[0] != 0:
        raise ValueError("the sphere must contain G = 0 as its first entry")
    ix, iy, iz = m % nx, (m // nx) % ny, m // (nx * ny)
    minus = (-ix) % nx + nx * ((-iy) % ny + ny * ((-iz) % nz))
    row = {int(v): i for i, v in enumerate(m)}
    try:
        partner = np.array([row[int(v)] for v in minus])
    except KeyError:
        raise ValueError("the sphere is not inversion symmetric") from None
    idx = np.arange(len(m))
    if np.any((partner == idx) & (idx != 0)):
        raise ValueError("Nyquist frequency inside the sphere")
    first = np.nonzero(partner >= idx)[0]
    return first.astype(np.int64), partner[first].astype(np.int64)


def to_half(x, g, mg):
    """Scaled real-symmetric part of full-sphere vectors (rows = plane waves)."""
    h = (x[g] + np.conj(x[mg])) / 2 * SQRT2
    h[0] = np.real(x[g[0]])
    return h


def from_half(h, g, mg, n_G):
    x = np.zeros((n_G,) + h.shape[1:], dtype=complex)
    x[g] = h / SQRT2
    x[mg] = np.conj(h) / SQRT2
    x[g[0]] = np.real(h[0])
    return x


def half_to_real(h):
    """(n_half, ...) complex half format -> (2 n_half - 1, ...) real unknowns [h0.re, h1.re, h1.im, h2.re, ...]."""
    r = np.empty((2 * h.shape[0] - 1,) + h.shape[1:])
    r[0] = h[0].real
    r[1::2] = h[1:].real
    r[2::2] = h[1:].imag
    return r


def real_to_half(r):
    nh = (r.shape[0] + 1) // 2
    h = np.zeros((nh,) + r.shape[1:], dtype=complex)
    h[0] = r[0]
    h[1:] = r[1::2] + 1j * r[2::2]
    return h


class RealSymmetricBlock:
    """The Gamma-point ``HamiltonianBlock`` acting on the real unknowns of real-symmetric vectors."""

    def __init__(self, ham_block, fft_size):
        self.H = ham_block
        self.n_G = ham_block.n_G
        self.g, self.mg = pair_tables(fft_size, ham_block.kpoint.mapping)
        kin = np.asarray(ham_block.kinetic)
        if np.max(np.abs(kin[self.g] - kin[self.mg])) > 1e-12 * (1 + np.max(np.abs(kin))):
            raise ValueError("kinetic(G) != kinetic(-G): not a Gamma-point block")
        self.n_half = len(self.g)
        self.n_real = 2 * self.n_half - 1
        kh = kin[self.g]
        self.kinetic_real = np.concatenate([[kh[0]], np.repeat(kh[1:], 2)])

    def pack(self, x):
        return half_to_real(to_half(x, self.g, self.mg))

    def unpack(self, r):
        return from_half(real_to_half(r), self.g, self.mg, self.n_G)

    def _apply_real(self, r):
        return self.pack(self.H.mul(self.unpack(r)))

    def mul(self, r):
        """Complex-linear extension (LOBPCG's rare re-randomisation draws complex columns): real parts and imaginary
        parts are mapped separately; real input stays real."""
        r = np.asarray(r)
        if np.iscomplexobj(r) and np.any(r.imag):
            return self._apply_real(r.real) + 1j * self._apply_real(r.imag)
        return self._apply_real(r.real).astype(complex)

    __matmul__ = mul

    def to_dense(self):
        return np.real(self.mul(np.eye(self.n_real)))


def lobpcg_gamma_real(ham_block, fft_size, X0, tol=1e-8, maxiter=100, n_conv_check=None, prec=True, **kw):
    """What ``dftk_mi_lobpcg`` does on a Gamma-real block: project the start block onto the real-symmetric subspace,
    iterate the reference's LOBPCG on the real unknowns, hand back full-sphere vectors."""
    blk = RealSymmetricBlock(ham_block, fft_size)
    Xr = blk.pack(np.asarray(X0))
    P = PreconditionerTPA(blk.kinetic_real) if prec else None
    res = lobpcg_hyper(blk.mul, Xr, maxiter=maxiter, prec=P, tol=tol, n_conv_check=n_conv_check, **kw)
    res["X_real"] = res["X"]
    res["X"] = blk.unpack(np.real(res["X"]))
    return res


# ---- two bands per transform (k_gr_pack / k_gr_unpack / k_gr_pack_full of gamma_kernels.hip) ------------------------
def pack_pair(ha, hb, g, mg, n_G):
    """Full-sphere image of a + i b for two half-format vectors (b = None: a alone)."""
    a = from_half(ha, g, mg, n_G)
    return a if hb is None else a + 1j * from_half(hb, g, mg, n_G)


def unpack_pair(w, g, mg):
    """Inverse on the output of a REAL-linear, conjugation-commuting operator: A = (W(G) + conj W(-G)) / 2 and
    B = (W(G) - conj W(-G)) / (2i), both back in the half format."""
    wg, wm = w[g], w[mg]
    s = np.full(len(g), SQRT2 / 2)
    s[0] = 0.5
    a = s * (wg + np.conj(wm))
    b = s * (wg - np.conj(wm)) / 1j
    a[0], b[0] = np.real(a[0]), np.real(b[0])
    return a, b


def align_phase(x, g, mg):
    """Rotate every column by exp(-i phi), exp(2 i phi) = s / |s|, s = sum_G x(G) x(-G): a real-symmetric vector times
    a global phase becomes +-itself, so that taking the real-symmetric part afterwards loses nothing (planned entry
    step of the device LOBPCG)."""
    x = np.asarray(x, dtype=complex)
    w = np.ones(len(g))
    w[1:] = 2.0
    s = np.sum(w[:, None] * x[g] * x[mg], axis=0)
    phi = np.where(np.abs(s) > 0, np.angle(s) / 2, 0.0)
    return x * np.exp(-1j * phi)[None, :]
