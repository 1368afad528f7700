"""Physics specification: ``Model``, ``ElementPsp``, standard models, supercells, k-grids.

Mirrors the parts of src/Model.jl, src/elements.jl, src/standard_models.jl:45-61,116-131,220,
src/supercell.jl:5-20 and src/bzmesh.jl:4-48 that the SCF hot path consumes.  Symmetry
detection, k-mesh reduction and density symmetrisation live in symmetry.py (``symmetries=True``); the default
is ``symmetries=False`` (explicit lists / unreduced Monkhorst-Pack meshes).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from .psp import ATOMIC_NUMBER, PspHgh, load_psp


@dataclass(eq=False)
class ElementPsp:
    symbol: str
    psp: PspHgh

    @property
    def charge_nuclear(self):
        return ATOMIC_NUMBER[self.symbol]

    @property
    def charge_ionic(self):
        return self.psp.Zion

    n_elec_valence = charge_ionic

    @property
    def n_elec_core(self):
        return self.charge_nuclear - self.psp.Zion


def compute_recip_lattice(lattice):
    """structure.jl:24-26 (lattice vectors are columns)."""
    return 2 * math.pi * np.linalg.inv(np.asarray(lattice, dtype=float).T)


class Model:
    """``Model`` (src/Model.jl): lattice (columns), atoms, fractional positions, term list, XC functionals,
    temperature / smearing, and the spin polarisation -- ``:none`` or ``:collinear`` (Model.jl:29-39), inferred from the
    per-atom initial ``magnetic_moments`` (mu_B, z components) as ``determine_spin_polarization`` does."""

    def __init__(self, lattice, atoms, positions, terms, functionals=("lda_x", "lda_c_pw"),
                 temperature=0.0, smearing=None, n_electrons=None, symmetries=False, magnetic_moments=(),
                 spin_polarization=None):
        self.lattice = np.asarray(lattice, dtype=float)
        self.atoms = list(atoms)
        self.positions = [np.asarray(p, dtype=float) for p in positions]
        if len(self.atoms) != len(self.positions):
            raise ValueError("Length of atoms and positions vectors need to agree.")
        self.term_types = tuple(terms)
        if not self.term_types:
            raise ValueError("Model without terms not supported.")
        self.functionals = tuple(functionals)
        self.temperature = float(temperature)
        self.smearing = smearing if smearing is not None else ("fermi_dirac" if temperature > 0 else "none")
        self.recip_lattice = compute_recip_lattice(self.lattice)
        self.unit_cell_volume = abs(np.linalg.det(self.lattice))
        self.n_electrons = int(sum(a.charge_ionic for a in self.atoms)) if n_electrons is None else n_electrons
        self.magnetic_moments = tuple(float(np.asarray(m, dtype=float).reshape(-1)[-1]) for m in magnetic_moments)
        if self.magnetic_moments and len(self.magnetic_moments) != len(self.atoms):
            raise ValueError("Length of atoms and magnetic_moments vectors need to agree.")
        if spin_polarization is None:
            spin_polarization = "collinear" if any(m != 0 for m in self.magnetic_moments) else "none"
        if spin_polarization not in ("none", "collinear"):
            raise NotImplementedError(f"spin_polarization = {spin_polarization!r} (Model.jl:190-192: :full is not "
                                      "supported by the reference either; :spinless is outside the hot path)")
        self.spin_polarization = spin_polarization
        self.n_spin_components = 2 if spin_polarization == "collinear" else 1          # Model.jl:196, :366-372
        # atom_groups (Model.jl:169): indices of identical elements, in order of first appearance
        self.atom_groups = []
        reps = []
        for i, a in enumerate(self.atoms):
            for g, r in zip(self.atom_groups, reps):
                if r is a or (r.symbol == a.symbol and r.psp.identifier == a.psp.identifier):
                    g.append(i)
                    break
            else:
                self.atom_groups.append([i])
                reps.append(a)

        # Model.jl:104-110: True = automatic detection (symmetry.py; the reference asks Spglib), False = identity
        # only, or an explicit list of SymOp.  Default False: explicit / unreduced k-lists unless asked for.
        from . import symmetry as _sym
        if symmetries is True:
            if self.spin_polarization == "collinear" and not self.magnetic_moments:
                symmetries = [_sym.identity()]          # default_symmetries (Model.jl:330-332): the breaking is unknown
            else:
                # atoms that carry different moments are different species for the search (symmetry.jl:66-125 hands
                # the moments to Spglib)
                groups = self.atom_groups
                if self.magnetic_moments:
                    groups = []
                    for g in self.atom_groups:
                        by_m = {}
                        for i in g:
                            by_m.setdefault(round(self.magnetic_moments[i], 10), []).append(i)
                        groups.extend(by_m.values())
                symmetries = _sym.symmetry_operations(self.lattice, groups, self.positions)
        elif symmetries is False or symmetries is None:
            symmetries = [_sym.identity()]
        self.symmetries = list(symmetries)

    @property
    def filled_occupation(self):   # Model.jl:352-360
        return 1 if self.spin_polarization == "collinear" else 2


def model_atomic(lattice, atoms, positions, extra_terms=(), **kw):
    """standard_models.jl:45-61."""
    terms = ("Kinetic", "AtomicLocal", "AtomicNonlocal", "Ewald", "PspCorrection") + tuple(extra_terms)
    if kw.get("temperature", 0) != 0:          # :56-58: the total becomes the free energy E - TS
        terms = terms + ("Entropy",)
    return Model(lattice, atoms, positions, terms, **kw)


def model_DFT(lattice, atoms, positions, functionals=("lda_x", "lda_c_pw"), **kw):
    """standard_models.jl:116-131; default functionals = ``LDA()`` (:220)."""
    return model_atomic(lattice, atoms, positions, extra_terms=("Hartree", "Xc"),
                        functionals=tuple(functionals), **kw)


def create_supercell(lattice, atoms, positions, supercell_size):
    """supercell.jl:5-20 (species-major, then (i,j,k) with i fastest)."""
    nx, ny, nz = supercell_size
    size = np.array([nx, ny, nz], dtype=float)
    lat = np.asarray(lattice, dtype=float) * size[None, :]
    new_atoms, new_pos = [], []
    for atom, pos in zip(atoms, positions):
        for k in range(nz):
            for j in range(ny):
                for i in range(nx):
                    new_pos.append((np.asarray(pos, dtype=float) + np.array([i, j, k])) / size)
                    new_atoms.append(atom)
    return lat, new_atoms, new_pos


def silicon_cell(supercell=(1, 1, 1), a=10.26, functional="lda"):
    """fcc silicon of examples/silicon.jl:5-11 with the vendored HGH pseudopotential."""
    lattice = a / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
    Si = ElementPsp("Si", load_psp("Si", functional))
    atoms, positions = [Si, Si], [np.ones(3) / 8, -np.ones(3) / 8]
    if tuple(supercell) != (1, 1, 1):
        lattice, atoms, positions = create_supercell(lattice, atoms, positions, supercell)
    return lattice, atoms, positions


@dataclass
class ExplicitKpoints:
    kcoords: list
    kweights: list


@dataclass
class MonkhorstPack:
    kgrid_size: tuple
    kshift: tuple = (0, 0, 0)

    def reducible(self) -> ExplicitKpoints:
        """bzmesh.jl:41-48; uniform weights (no symmetry reduction)."""
        size = np.array(self.kgrid_size)
        start = -np.floor((size - 1) / 2).astype(int)
        stop = np.ceil((size - 1) / 2).astype(int)
        ks = []
        for k in range(start[2], stop[2] + 1):
            for j in range(start[1], stop[1] + 1):
                for i in range(start[0], stop[0] + 1):
                    kc = (np.array(self.kshift, dtype=float) + np.array([i, j, k])) / size
                    ks.append(kc - np.floor(kc + 0.5))
        return ExplicitKpoints(ks, [1.0 / len(ks)] * len(ks))
