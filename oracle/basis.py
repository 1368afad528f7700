"""Model, k-grids, FFT grid and plane-wave basis (oracle restatement).  Test infrastructure only.

Restates ``src/structure.jl:24-61`` (reciprocal lattice, integer bounds),
``src/fft.jl:24-31,76-98,106-172,231-337`` (G vectors, normalisations, size rule,
sphere<->cube transforms), ``src/Kpoint.jl:20-41`` (sphere enumeration and mapping),
``src/bzmesh.jl:41-48`` (Monkhorst-Pack mesh), ``src/supercell.jl:5-20``,
``src/PlaneWaveBasis.jl:129-261`` (basis assembly; no MPI split, no symmetries).

Array conventions (shared with the device library):
cubes are NumPy arrays of shape (nz, ny, nx), C-order, so that ``cube.ravel()[i]`` with
``i = ix + nx*(iy + ny*iz)`` is Julia's column-major linear index minus one;
``Kpoint.mapping`` is that 0-based linear index, ascending.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import scipy.fft as sfft

from .psp import PspHgh, load_psp_hgh, ATOMIC_NUMBER


# ----------------------------------------------------------------------------- lattice helpers
def compute_recip_lattice(lattice):
    """2 pi inv(lattice') (structure.jl:24-26); lattice vectors are columns."""
    return 2 * math.pi * np.linalg.inv(np.asarray(lattice, dtype=float).T)


def compute_unit_cell_volume(lattice):
    return abs(np.linalg.det(np.asarray(lattice, dtype=float)))


def estimate_integer_lattice_bounds(M, delta, shift=(0, 0, 0), tol=math.sqrt(np.finfo(float).eps)):
    """structure.jl:50-61."""
    inv_lattice_t = np.linalg.inv(np.asarray(M, dtype=float).T)
    xlims = [np.linalg.norm(inv_lattice_t[:, i]) * delta + shift[i] for i in range(3)]
    return [0 if x == 0 else int(math.ceil(x - tol)) for x in xlims]


def _is_smooth(n, primes=(2, 3, 5)):
    for p in primes:
        while n % p == 0:
            n //= p
    return n == 1


def next_compatible_fft_size(size, smallprimes=(2, 3, 5), factors=(1,)):
    """fft.jl:277-287."""
    f = int(np.prod(factors))
    while not (size % f == 0 and (not smallprimes or _is_smooth(size, smallprimes))):
        size += 1
    return size


def compute_fft_size(lattice, Ecut, supersampling=2, factors=(1,)):
    """``compute_fft_size(...; algorithm=:fast)`` (fft.jl:231-267, 331-337)."""
    Gmax = supersampling * math.sqrt(2 * Ecut)
    Glims = estimate_integer_lattice_bounds(compute_recip_lattice(lattice), Gmax)
    return tuple(next_compatible_fft_size(2 * g + 1, factors=factors) for g in Glims)


def G_axis(n):
    """FFT-ordered integer frequencies [0..floor((n-1)/2), -ceil((n-1)/2)..-1] (fft.jl:24-31)."""
    stop = (n - 1) // 2
    start = -((n - 1) - (n - 1) // 2)
    return np.array(list(range(0, stop + 1)) + list(range(start, 0)), dtype=np.int64)


# ----------------------------------------------------------------------------- model
@dataclass
class ElementPsp:
    symbol: str
    psp: PspHgh

    @property
    def Z(self):
        return ATOMIC_NUMBER[self.symbol]

    @property
    def charge_ionic(self):
        return self.psp.Zion

    @property
    def n_elec_core(self):
        return self.Z - self.psp.Zion


@dataclass
class Model:
    """Subset of ``struct Model`` (src/Model.jl) used on the hot path: ``spin_polarization`` none or collinear
    (Model.jl:29-39, :190-196; inferred from ``magnetic_moments`` as ``determine_spin_polarization`` does)."""
    lattice: np.ndarray
    atoms: list
    positions: list
    terms: tuple = ("Kinetic", "AtomicLocal", "AtomicNonlocal", "Ewald", "PspCorrection",
                    "Hartree", "Xc")
    functionals: tuple = ("lda_x", "lda_c_pw")
    temperature: float = 0.0
    smearing: str = "none"
    n_electrons: int | None = None
    # Model.jl:104-110: true = automatic detection, false = identity only, or an explicit list of SymOp.
    # (Default false here: the pinned reference tests of this oracle run explicit / unreduced k-lists.)
    symmetries: object = False
    # per-atom initial magnetic moments in units of mu_B (z components; Model.jl:91-118): used to infer the spin
    # polarisation and to break the symmetries, NOT stored as the state of the calculation
    magnetic_moments: tuple = ()
    spin_polarization: str | None = None      # "none" | "collinear"; None = infer from magnetic_moments

    def __post_init__(self):
        self.lattice = np.asarray(self.lattice, dtype=float)
        self.positions = [np.asarray(p, dtype=float) for p in self.positions]
        self.recip_lattice = compute_recip_lattice(self.lattice)
        self.unit_cell_volume = compute_unit_cell_volume(self.lattice)
        if self.n_electrons is None:
            self.n_electrons = int(sum(a.charge_ionic for a in self.atoms))
        # atom_groups: indices of identical elements (Model.jl:169)
        groups, seen = [], []
        for i, a in enumerate(self.atoms):
            for g, s in zip(groups, seen):
                if s is a or (s.symbol == a.symbol and s.psp.identifier == a.psp.identifier):
                    g.append(i)
                    break
            else:
                groups.append([i])
                seen.append(a)
        self.atom_groups = groups
        self.magnetic_moments = tuple(float(np.asarray(m, dtype=float).reshape(-1)[-1]) for m in self.magnetic_moments)
        if self.magnetic_moments and len(self.magnetic_moments) != len(self.atoms):
            raise ValueError("Length of atoms and magnetic_moments vectors need to agree.")
        if self.spin_polarization is None:            # determine_spin_polarization (Model.jl)
            self.spin_polarization = "collinear" if any(m != 0 for m in self.magnetic_moments) else "none"
        if self.spin_polarization not in ("none", "collinear"):
            raise NotImplementedError(f"spin_polarization = {self.spin_polarization}")
        self.n_spin_components = 2 if self.spin_polarization == "collinear" else 1        # Model.jl:196, :366-372
        self.filled_occupation = 1 if self.spin_polarization == "collinear" else 2        # Model.jl:352-360
        if self.n_electrons % (self.n_spin_components * self.filled_occupation) != 0 and self.temperature == 0:
            raise ValueError("Odd number of electrons without smearing (occupation.jl:163-172)")
        if self.temperature > 0 and self.smearing == "none":
            self.smearing = "fermi_dirac"
        from . import symmetry as _sym
        if self.symmetries is True:
            if self.spin_polarization == "collinear" and not self.magnetic_moments:
                self.symmetries = [_sym.identity()]      # default_symmetries (Model.jl:330-332): breaking unknown
            else:
                # atoms carrying different moments are different "species" for the symmetry search (symmetry.jl:66-125
                # hands the moments to Spglib)
                groups = self.atom_groups
                if self.magnetic_moments:
                    groups = []
                    for g in self.atom_groups:
                        by_m = {}
                        for i in g:
                            by_m.setdefault(round(self.magnetic_moments[i], 10), []).append(i)
                        groups.extend(by_m.values())
                self.symmetries = _sym.symmetry_operations(self.lattice, groups, self.positions)
        elif self.symmetries is False or self.symmetries is None:
            self.symmetries = [_sym.identity()]


def model_atomic(lattice, atoms, positions, extra_terms=(), **kw):
    """``model_atomic`` (standard_models.jl:45-61): Kinetic + AtomicLocal + AtomicNonlocal + ..."""
    terms = ("Kinetic", "AtomicLocal", "AtomicNonlocal", "Ewald", "PspCorrection") + tuple(extra_terms)
    if kw.get("temperature", 0) != 0:          # standard_models.jl:56-58: the total becomes the free energy E - TS
        terms = terms + ("Entropy",)
    return Model(lattice, atoms, positions, terms=terms, **kw)


def model_DFT(lattice, atoms, positions, functionals=("lda_x", "lda_c_pw"), **kw):
    """``model_DFT`` (standard_models.jl:116-131): atomic model + Hartree + Xc."""
    return model_atomic(lattice, atoms, positions, extra_terms=("Hartree", "Xc"),
                        functionals=tuple(functionals), **kw)


def create_supercell(lattice, atoms, positions, supercell_size):
    """supercell.jl:5-20: atoms species-major, then (i,j,k) with i fastest."""
    nx, ny, nz = supercell_size
    lat = np.asarray(lattice, dtype=float) * np.array([nx, ny, nz])[None, :]
    new_atoms, new_pos = [], []
    for atom, pos in zip(atoms, positions):
        for k in range(nz):
            for j in range(ny):
                for i in range(nx):
                    new_pos.append((np.asarray(pos, dtype=float) + np.array([i, j, k]))
                                   / np.array([nx, ny, nz]))
                    new_atoms.append(atom)
    return lat, new_atoms, new_pos


def silicon_primitive(a=10.26, functional="lda"):
    """fcc Si as in examples/silicon.jl:5-11 (HGH pseudopotential instead of PseudoDojo)."""
    lattice = a / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
    Si = ElementPsp("Si", load_psp_hgh("Si", functional))
    return lattice, [Si, Si], [np.ones(3) / 8, -np.ones(3) / 8]


# ----------------------------------------------------------------------------- k grids
@dataclass
class ExplicitKpoints:
    kcoords: list
    kweights: list


@dataclass
class MonkhorstPack:
    kgrid_size: tuple
    kshift: tuple = (0, 0, 0)

    def reducible(self):
        """bzmesh.jl:41-48 (+ normalisation :4-9); weights uniform (symmetries=false)."""
        size = np.array(self.kgrid_size)
        start = -np.floor((size - 1) / 2).astype(int)
        stop = np.ceil((size - 1) / 2).astype(int)
        out = []
        # Julia comprehension: i fastest, then j, then k
        for k in range(start[2], stop[2] + 1):
            for j in range(start[1], stop[1] + 1):
                for i in range(start[0], stop[0] + 1):
                    kc = (np.array(self.kshift, dtype=float) + np.array([i, j, k])) / size
                    kc = kc - np.floor(kc + 0.5)   # into [-0.5, 0.5), ties up
                    out.append(kc)
        n = len(out)
        return ExplicitKpoints(out, [1.0 / n] * n)


# ----------------------------------------------------------------------------- basis
@dataclass
class Kpoint:
    """Kpoint.jl:6-18."""
    spin: int
    coordinate: np.ndarray
    G_vectors: np.ndarray     # (n_G, 3) int
    mapping: np.ndarray       # (n_G,) int64, 0-based x-fastest linear cube index, ascending


class PlaneWaveBasis:
    """Subset of PlaneWaveBasis.jl:25-97 / :129-261 (single process, no symmetries)."""

    def __init__(self, model: Model, Ecut: float, kgrid=None, fft_size=None, build_terms=True,
                 use_symmetries_for_kpoint_reduction=True):
        from . import symmetry as _sym
        self.model = model
        self.Ecut = float(Ecut)
        symmetries_respect_rgrid = fft_size is None                      # PlaneWaveBasis.jl:330
        if fft_size is None:
            # FFT size compatible with the fractional translations of the symmetries (PlaneWaveBasis.jl:349-361)
            factors = (1,)
            if any(not s.isone() for s in model.symmetries):
                from fractions import Fraction
                den = {Fraction(float(x)).limit_denominator(1000).denominator for s in model.symmetries for x in s.w}
                factors = tuple(sorted(den & {2, 3, 4, 6})) or (1,)
            fft_size = compute_fft_size(model.lattice, Ecut, factors=factors)
        self.fft_size = tuple(int(n) for n in fft_size)
        nx, ny, nz = self.fft_size
        self.N = nx * ny * nz
        if kgrid is None:
            kgrid = MonkhorstPack((1, 1, 1))
        # symmetries that survive the discretisation (PlaneWaveBasis.jl:161-173)
        symmetries = list(model.symmetries)
        if symmetries_respect_rgrid:
            symmetries = _sym.symmetries_preserving_rgrid(symmetries, self.fft_size)
        if isinstance(kgrid, MonkhorstPack):
            symmetries = _sym.symmetries_preserving_kgrid(symmetries, kgrid.kgrid_size, kgrid.kshift)
            if use_symmetries_for_kpoint_reduction and any(not s.isone() for s in symmetries):
                kc, kw = _sym.irreducible_kcoords(kgrid.kgrid_size, symmetries, kgrid.kshift)
                kgrid = ExplicitKpoints(kc, kw)
            else:
                kgrid = kgrid.reducible()        # (full mesh; the density is still symmetrised, as the reference)
        else:
            symmetries = _sym.symmetries_preserving_kcoords(symmetries, kgrid.kcoords)   # symmetry.jl:163-174
        self.symmetries = symmetries
        self.kcoords = [np.asarray(k, dtype=float) for k in kgrid.kcoords]
        self.kweights = [float(w) for w in kgrid.kweights]
        self.dvol = model.unit_cell_volume / self.N
        # fft.jl:81-88
        self.ifft_normalization = 1 / math.sqrt(model.unit_cell_volume)
        self.fft_normalization = math.sqrt(model.unit_cell_volume) / self.N
        self.Gx, self.Gy, self.Gz = G_axis(nx), G_axis(ny), G_axis(nz)
        self.kpoints = [self._build_kpoint(k) for k in self.kcoords]
        self.n_kcoords = len(self.kcoords)
        if model.n_spin_components == 2:
            # collinear spin: the k-point list is doubled -- all spin-up blocks, then all spin-down blocks -- and so are
            # the weights, which then sum to n_spin_components (build_kpoints Kpoint.jl:58-74, PlaneWaveBasis.jl:218-232)
            self.kpoints = self.kpoints + [Kpoint(2, k.coordinate, k.G_vectors, k.mapping) for k in self.kpoints]
            self.kweights = self.kweights + self.kweights
        self.terms = None
        if build_terms:
            from .terms import instantiate_terms
            self.terms = instantiate_terms(self)

    # cube G vectors in reduced coordinates, arrays of shape (nz, ny, nx)
    def G_vectors_cube(self):
        gz, gy, gx = np.meshgrid(self.Gz, self.Gy, self.Gx, indexing="ij")
        return gx, gy, gz

    def G_vectors_cart_cube(self):
        gx, gy, gz = self.G_vectors_cube()
        B = self.model.recip_lattice
        G = np.stack([gx, gy, gz], axis=-1).astype(float)
        return G @ B.T   # (..., 3) cartesian

    def _build_kpoint(self, kcoord):
        """Kpoint.jl:20-41: enumerate the whole cube in column-major order, keep |k+G|^2/2 <= Ecut."""
        gx, gy, gz = self.G_vectors_cube()
        G = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1)
        Gk = (G + kcoord[None, :]) @ self.model.recip_lattice.T
        keep = np.sum(Gk * Gk, axis=1) / 2 <= self.Ecut
        mapping = np.nonzero(keep)[0].astype(np.int64)
        return Kpoint(1, np.asarray(kcoord, dtype=float), G[mapping].astype(np.int64), mapping)

    def Gplusk_vectors_cart(self, kpt: Kpoint):
        return (kpt.G_vectors + kpt.coordinate[None, :]) @ self.model.recip_lattice.T

    # ---- FFTs.  Cube transforms are normalised (fft.jl:106-109,155-161); sphere transforms
    #      take normalize= like the reference (fft.jl:110-122,162-172).
    def ifft_cube(self, f_fourier):
        return sfft.ifftn(f_fourier, norm="forward", workers=-1) * self.ifft_normalization

    def irfft_cube(self, f_fourier):
        return np.real(self.ifft_cube(f_fourier))

    def fft_cube(self, f_real):
        return sfft.fftn(np.asarray(f_real, dtype=complex), norm="backward",
                         workers=-1) * self.fft_normalization

    def ifft(self, kpt: Kpoint, f_fourier, normalize=True):
        cube = np.zeros(self.N, dtype=complex)
        cube[kpt.mapping] = f_fourier
        nx, ny, nz = self.fft_size
        out = sfft.ifftn(cube.reshape(nz, ny, nx), norm="forward", workers=-1)  # unnormalised BFFT
        if normalize:
            out = out * self.ifft_normalization
        return out

    def fft(self, kpt: Kpoint, f_real, normalize=True):
        out = sfft.fftn(f_real, norm="backward", workers=-1).ravel()[kpt.mapping]
        if normalize:
            out = out * self.fft_normalization
        return out

    def enforce_real(self, f_fourier):
        """Zero the coefficients whose -G partner is not on the grid (symmetry.jl:318-337,550-552)."""
        out = np.array(f_fourier, dtype=complex, copy=True)
        nx, ny, nz = self.fft_size
        for axis, (n, G) in enumerate(((nz, self.Gz), (ny, self.Gy), (nx, self.Gx))):
            if n % 2 == 0:
                bad = np.nonzero(G == -(n // 2))[0]
                sl = [slice(None)] * 3
                sl[axis] = bad
                out[tuple(sl)] = 0
        return out

    def r_vectors_frac(self):
        nx, ny, nz = self.fft_size
        rz, ry, rx = np.meshgrid(np.arange(nz) / nz, np.arange(ny) / ny, np.arange(nx) / nx,
                                 indexing="ij")
        return rx, ry, rz
