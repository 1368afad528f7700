"""Crystal symmetries on the host + density symmetrisation on the device (mirror of src/SymOp.jl, src/symmetry.jl,
src/bzmesh.jl:41-101).

* ``symmetry_operations`` (symmetry.jl:66-125): the reference asks Spglib; here the same group is found the way
  Spglib finds it -- reduce the cell, scan the 3^9 integer matrices with entries in {-1, 0, 1} for those preserving
  the metric, then the translations w that map every atom onto an atom of its species ((U u)(x) = u(W x + w);
  S = W', tau = -W^-1 w, SymOp.jl:1-60).  (The oracle uses a different search -- shells of integer vectors in the
  unreduced cell -- so the two are independent restatements; both are pinned by the reference's Spglib counts.)
* ``symmetries_preserving_kgrid / _rgrid`` (:163-211), ``irreducible_kcoords`` (bzmesh.jl:54-101, no time reversal
  as the reference): the k-points a ``PlaneWaveBasis`` really computes.
* ``symmetrize_rho`` (:282-357): rho(G) <- 1/|S| sum_s e^{-2 pi i G.tau_s} rho(S_s^-1 G) on the device -- per
  symmetry one gather of the Fourier cube through a cached index table and one phase multiply, between the
  library's cube FFTs; called by ``compute_density`` after the all-reduce (densities.jl:47).
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass

import numpy as np

SYMMETRY_TOLERANCE = 1e-5


@dataclass
class SymOp:
    W: np.ndarray      # 3x3 int, real-space rotation in reduced coordinates
    w: np.ndarray      # translation in [0, 1)^3
    S: np.ndarray      # W'
    tau: np.ndarray    # -W^-1 w

    @staticmethod
    def make(W, w):
        W = np.asarray(W, dtype=int)
        w = np.mod(np.asarray(w, dtype=float), 1.0)
        w = np.where(np.abs(w - 1.0) < 1e-12, 0.0, w)
        return SymOp(W, w, W.T.copy(), -np.linalg.solve(W.astype(float), w))

    def isone(self):
        return np.array_equal(self.W, np.eye(3, dtype=int)) and not self.w.any()


def identity():
    return SymOp.make(np.eye(3, dtype=int), np.zeros(3))


def _approx_integer(x, tol):
    return np.all(np.abs(x - np.round(x)) <= tol)


def _reduced_cell(lattice):
    """Minkowski-style reduction of the cell: returns (A, T) with A = lattice @ T, T unimodular integer, such that no
    column of A gets shorter by adding integer combinations (coefficients in {-1, 0, 1}) of the other two.  In such a
    basis every lattice automorphism has entries in {-1, 0, 1} -- the property Spglib's own point-group search relies
    on after its Delaunay reduction (the reference: symmetry.jl:88-93 -> Spglib)."""
    A = np.array(lattice, dtype=float)
    T = np.eye(3, dtype=int)
    combos = [c for c in itertools.product((-1, 0, 1), repeat=2) if any(c)]
    for _ in range(200):
        improved = False
        for i in range(3):
            j, k = [a for a in range(3) if a != i]
            # Gauss step against each neighbour, then the mixed combinations
            for other in (j, k):
                mu = int(np.rint(A[:, i] @ A[:, other] / (A[:, other] @ A[:, other])))
                if mu and (A[:, i] - mu * A[:, other]) @ (A[:, i] - mu * A[:, other]) < (A[:, i] @ A[:, i]) * (1 - 1e-12):
                    A[:, i] -= mu * A[:, other]
                    T[:, i] -= mu * T[:, other]
                    improved = True
            for x, y in combos:
                cand = A[:, i] + x * A[:, j] + y * A[:, k]
                if cand @ cand < (A[:, i] @ A[:, i]) * (1 - 1e-12):
                    A[:, i] = cand
                    T[:, i] += x * T[:, j] + y * T[:, k]
                    improved = True
        if not improved:
            break
    return A, T


def _lattice_point_group(lattice, tol):
    """All integer W (in the basis of ``lattice``) with W' G W = G, G = lattice' lattice: exhaustive scan of the
    3^9 matrices with entries in {-1, 0, 1} in the reduced cell, mapped back with the reduction's unimodular T."""
    A, T = _reduced_cell(lattice)
    G = A.T @ A
    entries = np.array(list(itertools.product((-1, 0, 1), repeat=9)), dtype=np.int64).reshape(-1, 3, 3)
    Gw = np.einsum("nki,kl,nlj->nij", entries, G, entries)
    keep = np.max(np.abs(Gw - G), axis=(1, 2)) <= tol * np.max(np.abs(G))
    Tinv = np.rint(np.linalg.inv(T.astype(float))).astype(np.int64)
    return [T @ Wr @ Tinv for Wr in entries[keep]]


def symmetry_operations(lattice, atom_groups, positions, tol=SYMMETRY_TOLERANCE):
    """All (W, w) with W an integer matrix preserving the metric lattice' lattice and W a + w an atom of the same
    species for every atom a (symmetry.jl:66-125: the reference delegates to Spglib; this is Spglib's scheme -- point
    group of the REDUCED cell by exhaustive scan, then the translations that carry the sparsest species onto itself).
    Pinned by the reference's own Spglib-derived counts in tests/test_host_side.py."""
    lattice = np.asarray(lattice, dtype=float)
    positions = [np.asarray(p, dtype=float) for p in positions]
    if not positions:
        return [identity()]
    species_pos = [np.array([positions[i] for i in group]) for group in atom_groups if len(group)]
    pivot = min(species_pos, key=len)

    def maps_crystal(W, w):
        for P in species_pos:
            delta = (P @ W.T + w)[:, None, :] - P[None, :, :]
            delta -= np.rint(delta)
            if not np.all((np.abs(delta) <= tol).all(axis=2).any(axis=1)):
                return False
        return True

    ops = []
    for W in _lattice_point_group(lattice, tol):
        image0 = W @ pivot[0]
        for target in pivot:                      # W x_0 + w must be an atom of the pivot species
            w = target - image0
            if not maps_crystal(W, w):
                continue
            op = SymOp.make(W, w)
            if not any(np.array_equal(op.W, o.W) and _approx_integer(op.w - o.w, tol) for o in ops):
                ops.append(op)
    ops.sort(key=lambda o: (not o.isone(),))
    # a set that is not a GROUP would silently give wrong irreducible weights: the closure is verified
    # (SymOp.jl check_group) and anything else falls back to the identity
    if 1 < len(ops) <= 192:
        try:
            check_group(ops, tol)
        except ValueError:
            import warnings
            warnings.warn("symmetry_operations: the detected operations do not form a group; using the identity only")
            return [identity()]
    return ops


def normalize_kpoint_coordinate(k):
    """Bring into [-0.5, 0.5) (bzmesh.jl:5-10: round half up)."""
    k = np.asarray(k, dtype=float)
    return k - np.floor(k + 0.5)


def reducible_kcoords(kgrid_size, kshift=(0, 0, 0)):
    """bzmesh.jl:41-48 (first index fastest, as Julia's comprehension)."""
    size = np.asarray(kgrid_size, dtype=int)
    start = -np.floor((size - 1) / 2).astype(int)
    stop = np.ceil((size - 1) / 2).astype(int)
    ks = []
    for k in range(start[2], stop[2] + 1):
        for j in range(start[1], stop[1] + 1):
            for i in range(start[0], stop[0] + 1):
                ks.append(normalize_kpoint_coordinate((np.asarray(kshift, dtype=float) + np.array([i, j, k])) / size))
    return ks


def _grid_key(k, size, kshift):
    """Integer address of a grid point modulo the reciprocal lattice."""
    return tuple(np.mod(np.round(np.asarray(k) * size - np.asarray(kshift, dtype=float)).astype(int), size))


def _in_grid(k, size, kshift, tol=1e-8):
    x = np.asarray(k) * size - np.asarray(kshift, dtype=float)
    return bool(np.all(np.abs(x - np.round(x)) < tol))


def symmetries_preserving_kgrid(symmetries, kgrid_size, kshift=(0, 0, 0)):
    """symmetry.jl:176-193 (Monkhorst-Pack: by linearity the origin and the three unit steps suffice)."""
    size = np.asarray(kgrid_size, dtype=int)
    shift = np.asarray(kshift, dtype=float)
    probes = [(shift + d) / size for d in (np.zeros(3), np.eye(3)[0], np.eye(3)[1], np.eye(3)[2])]
    return [s for s in symmetries if all(_in_grid(s.S @ k, size, shift) for k in probes)]


def symmetries_preserving_kcoords(symmetries, kcoords, tol=1e-8):
    """symmetry.jl:163-174 (generic k-lists): unfold the list with all symmetries, keep the operations that map the
    unfolded set onto itself."""
    if all(s.isone() for s in symmetries):
        return list(symmetries)
    allk = []
    for k in kcoords:
        for s in symmetries:
            q = normalize_kpoint_coordinate(s.S @ np.asarray(k, dtype=float))
            if not any(np.all(np.abs(normalize_kpoint_coordinate(q - p)) < tol) for p in allk):
                allk.append(q)

    def inside(q):
        return any(np.all(np.abs(normalize_kpoint_coordinate(q - p)) < tol) for p in allk)
    return [s for s in symmetries if all(inside(s.S @ k) for k in allk)]


def symmetries_preserving_rgrid(symmetries, fft_size, tol=SYMMETRY_TOLERANCE):
    """symmetry.jl:198-211."""
    fft_size = np.asarray(fft_size, dtype=int)
    out = []
    for s in symmetries:
        ok = True
        for i in range(3):
            e = np.zeros(3)
            e[i] = 1.0 / fft_size[i]
            r = s.W @ e + s.w
            if np.any(np.abs(r * fft_size - np.round(r * fft_size)) / fft_size > tol):
                ok = False
        if ok:
            out.append(s)
    return out


def irreducible_kcoords(kgrid_size, symmetries, kshift=(0, 0, 0)):
    """bzmesh.jl:54-101: orbits of the mesh under the S = W' of the symmetries (no time reversal); the first point
    of an orbit (in reducible order) represents it, weight = orbit size / mesh size."""
    size = np.asarray(kgrid_size, dtype=int)
    if np.all(size == 1):
        return [np.asarray(kshift, dtype=float)], [1.0]
    red = reducible_kcoords(size, kshift)
    index = {_grid_key(k, size, kshift): i for i, k in enumerate(red)}
    rep = [-1] * len(red)
    for i, k in enumerate(red):
        if rep[i] >= 0:
            continue
        rep[i] = i
        for s in symmetries:
            j = index.get(_grid_key(s.S @ k, size, kshift))
            if j is None:
                raise ValueError("symmetry does not preserve the k-grid")
            if rep[j] < 0:
                rep[j] = i
    # orbits may be joined through chains only if the operations form a group; verify
    irr = sorted(set(rep))
    counts = {i: rep.count(i) for i in irr}
    return [red[i] for i in irr], [counts[i] / len(red) for i in irr]


def _G_cube(fft_size):
    """Integer G vectors of the cube in storage order, shape (nz, ny, nx, 3)."""
    from .basis import G_axis
    nx, ny, nz = fft_size
    GZ, GY, GX = np.meshgrid(G_axis(nz), G_axis(ny), G_axis(nx), indexing="ij")
    return np.stack([GX, GY, GZ], axis=-1)


def _index_of(G, fft_size):
    """index_G_vectors(fft_size, G): linear cube index of integer G, -1 outside the cube's frequency range."""
    nx, ny, nz = fft_size
    ok = np.ones(G.shape[:-1], dtype=bool)
    idx = []
    for a, n in enumerate((nx, ny, nz)):
        g = G[..., a]
        lo, hi = -((n - 1) - (n - 1) // 2), (n - 1) // 2
        ok &= (g >= lo) & (g <= hi)
        idx.append(np.mod(g, n))
    lin = idx[0] + nx * (idx[1] + ny * idx[2])
    return np.where(ok, lin, -1)


def _tables(basis):
    """Per symmetry: gather index into the flattened Fourier cube (clamped), validity mask folded into the phase
    e^{-2 pi i G.tau} (zero where S^-1 G leaves the cube).  Cached on the basis (device tensors)."""
    import torch
    cache = getattr(basis, "_symm_tables", None)
    if cache is not None:
        return cache
    G = _G_cube(basis.fft_size)
    tabs = []
    for s in basis.symmetries:
        invS = np.round(np.linalg.inv(s.S.astype(float))).astype(int)
        idx = _index_of(G @ invS.T, basis.fft_size).reshape(-1)
        phase = np.exp(-2j * np.pi * (G @ s.tau)).reshape(-1) if s.tau.any() else np.ones(idx.size, dtype=complex)
        phase = np.where(idx >= 0, phase, 0.0)
        tabs.append((torch.from_numpy(np.maximum(idx, 0).astype(np.int32)).to(basis.device),
                     torch.from_numpy(phase).to(basis.device) if (s.tau.any() or np.any(idx < 0))
                     else None))
    basis._symm_tables = tabs
    return tabs


def symmetrize_rho(basis, rho, do_lowpass=True):
    """``symmetrize_rho(basis, rho; do_lowpass)`` (symmetry.jl:346-357): ONE library call (``dftk_mi_symmetrize_rho``:
    cube FFT, all symmetries accumulated per G in one kernel, low-pass, inverse FFT).  ``DFTK_MI_TORCH_LOCAL=1`` keeps
    the torch formulation (one gather per symmetry), the parity twin of tests/test_gpu_symmetry.py."""
    import os
    import torch
    syms = basis.symmetries
    if all(s.isone() for s in syms):
        return rho
    if os.environ.get("DFTK_MI_TORCH_LOCAL") is None:
        from . import _lib
        S_h = np.ascontiguousarray(np.stack([np.asfortranarray(s.S).ravel(order="F") for s in syms]), dtype=np.int32)
        tau_h = np.ascontiguousarray(np.stack([s.tau for s in syms]), dtype=np.float64)
        rin = rho.to(torch.float64).contiguous()
        out = torch.empty_like(rin)
        basis.pre_call()
        _lib.check(basis.lib.dftk_mi_symmetrize_rho(basis._cube_handle, len(syms), S_h.ctypes.data, tau_h.ctypes.data,
                                                    1 if do_lowpass else 0, rin.data_ptr(), out.data_ptr()))
        basis.post_call()
        return out
    rf = basis.fft(rho).reshape(-1)
    acc = torch.zeros_like(rf)
    for idx, phase in _tables(basis):                  # accumulate_over_symmetries! (:282-319)
        acc += rf[idx.long()] * phase if phase is not None else rf[idx.long()]
    if do_lowpass:                                      # lowpass_for_symmetry! (:323-343)
        G = _G_cube(basis.fft_size)
        keep = np.ones(G.shape[:-1], dtype=bool)
        for s in syms:
            keep &= _index_of(G @ s.S.T, basis.fft_size) >= 0
        acc = acc * torch.from_numpy(keep.reshape(-1).astype(np.float64)).to(basis.device)
    return basis.irfft((acc / len(syms)).reshape(rho.shape))


def check_group(symmetries, tol=SYMMETRY_TOLERANCE):
    """SymOp.jl ``check_group``: identity, inverses and products are in the set.  Raises ``ValueError`` (not an
    ``assert``: the check must survive ``python -O``)."""
    def member(W, w):
        return any(np.array_equal(W, o.W) and _approx_integer(w - o.w, tol) for o in symmetries)
    if not member(np.eye(3, dtype=int), np.zeros(3)):
        raise ValueError("check_group: the identity is missing")
    for s in symmetries:
        Wi = np.round(np.linalg.inv(s.W.astype(float))).astype(int)
        if not member(Wi, -Wi @ s.w):
            raise ValueError("check_group: an inverse is missing")
        for t in symmetries:
            if not member(s.W @ t.W, s.w + s.W @ t.w):
                raise ValueError("check_group: a product is missing")
    return symmetries
