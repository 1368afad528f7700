"""``PlaneWaveBasis`` / ``Kpoint`` host mirror bound to the device library.

Reference: src/PlaneWaveBasis.jl:25-97,129-261,323-369 (struct, constructor), src/Kpoint.jl:6-41,
src/fft.jl:24-31,76-98,231-287 (G vectors, normalisations, FFT size rule), structure.jl:50-61.

Array layout: cubes are torch tensors of shape (nz, ny, nx) -- x fastest, i.e.
Julia's (nx, ny, nz) column-major array; orbital blocks are tensors of shape (n_bands, n_G)
(= column-major n_G x n_bands, one column per band); everything fp64 / complex128 in HBM.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib
from .comm import KptComm, distribute_kpoints, split_evenly
from .model import ExplicitKpoints, Model, MonkhorstPack


def estimate_integer_lattice_bounds(M, delta, shift=(0, 0, 0), tol=math.sqrt(np.finfo(float).eps)):
    """structure.jl:50-61."""
    inv_t = np.linalg.inv(np.asarray(M, dtype=float).T)
    xlims = [np.linalg.norm(inv_t[:, i]) * delta + shift[i] for i in range(3)]
    return [0 if x == 0 else int(math.ceil(x - tol)) for x in xlims]


def next_compatible_fft_size(size, smallprimes=(2, 3, 5), factors=(1,)):
    """fft.jl:277-287."""
    def smooth(n):
        for p in smallprimes:
            while n % p == 0:
                n //= p
        return n == 1
    f = int(np.prod(factors))
    while not (size % f == 0 and (not smallprimes or smooth(size))):
        size += 1
    return size


def compute_fft_size(model_or_lattice, Ecut, supersampling=2, factors=(1,)):
    """``compute_fft_size(model, Ecut; algorithm=:fast)`` (fft.jl:231-267, 331-337)."""
    lattice = getattr(model_or_lattice, "lattice", model_or_lattice)
    recip = 2 * math.pi * np.linalg.inv(np.asarray(lattice, dtype=float).T)
    Glims = estimate_integer_lattice_bounds(recip, supersampling * math.sqrt(2 * Ecut))
    return tuple(next_compatible_fft_size(2 * g + 1, factors=factors) for g in Glims)


def G_axis(n):
    """[0..floor((n-1)/2), -ceil((n-1)/2)..-1] (fft.jl:24-31)."""
    stop = (n - 1) // 2
    return np.array(list(range(0, stop + 1)) + list(range(stop - (n - 1), 0)), dtype=np.int64)


class Kpoint:
    """Kpoint.jl:6-18 + the device-side k-block handle (sphere tables, kinetic vector)."""

    def __init__(self, basis, coordinate, spin=1, lane=0):
        self.basis = basis
        self.spin = spin
        self.lane = lane            # which of the basis' library handles (HIP streams) owns this k-block
        self.coordinate = np.asarray(coordinate, dtype=float)
        dev = basis.device
        B = torch.tensor(basis.model.recip_lattice, dtype=torch.float64, device=dev)
        k = torch.tensor(self.coordinate, dtype=torch.float64, device=dev)
        if basis.lib is not None:
            # sphere enumeration behind the ABI (dftk_mi_kpoint_sphere_host, Kpoint.jl:28-35): one native pass over
            # the cube in index order, no cube-sized temporaries
            nx, ny, nz = basis.fft_size
            Bh = np.asfortranarray(basis.model.recip_lattice, dtype=np.float64)
            kh = np.ascontiguousarray(self.coordinate, dtype=np.float64)
            n = C.c_int64()
            _lib.check(basis.lib.dftk_mi_kpoint_sphere_host(nx, ny, nz, Bh.ctypes.data, kh.ctypes.data, basis.Ecut, 0,
                                                            C.byref(n), None, None, None))
            map_h = np.zeros(n.value, dtype=np.int64)
            kin_h = np.zeros(n.value, dtype=np.float64)
            G_h = np.zeros((n.value, 3), dtype=np.int32)
            _lib.check(basis.lib.dftk_mi_kpoint_sphere_host(nx, ny, nz, Bh.ctypes.data, kh.ctypes.data, basis.Ecut,
                                                            n.value, C.byref(n), map_h.ctypes.data, kin_h.ctypes.data,
                                                            G_h.ctypes.data))
            mapping = torch.from_numpy(map_h).to(dev)
            self.mapping = map_h
            self.mapping_device = mapping
            self.G_vectors = torch.from_numpy(G_h).to(dev).to(torch.int64)            # (n_G, 3) on device
            Gp = self.G_vectors.to(torch.float64) + k[None, :]
            self.Gplusk_cart = Gp[:, 0:1] * B[:, 0][None, :] + Gp[:, 1:2] * B[:, 1][None, :] + Gp[:, 2:3] * B[:, 2][None, :]
            self.kinetic = torch.from_numpy(kin_h).to(dev)                            # 1/2 |k+G|^2 (kinetic.jl:31-35)
        else:
            # descriptor-only basis (device="cpu"): the same enumeration with torch
            # enumerate the cube in column-major order, keep |B (G + k)|^2 / 2 <= Ecut (Kpoint.jl:28-35)
            gx, gy, gz = basis.G_vectors_cube()
            G = torch.stack([gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)], dim=1).to(torch.float64)
            Gp = G + k[None, :]
            # B (G + k) spelled out (keeps the set-up free of BLAS calls with degenerate 3-wide shapes)
            Gk = Gp[:, 0:1] * B[:, 0][None, :] + Gp[:, 1:2] * B[:, 1][None, :] + Gp[:, 2:3] * B[:, 2][None, :]
            kin_all = (Gk * Gk).sum(dim=1) / 2
            mapping = torch.nonzero(kin_all <= basis.Ecut).reshape(-1)
            self.mapping = mapping.cpu().numpy().astype(np.int64)            # 0-based, ascending
            self.mapping_device = mapping
            self.G_vectors = G[mapping].to(torch.int64)                     # (n_G, 3) on device
            self.Gplusk_cart = Gk[mapping]                                  # (n_G, 3) cartesian, device
            self.kinetic = kin_all[mapping].contiguous()                    # 1/2 |k+G|^2 (kinetic.jl:31-35)
        self.n_G = int(mapping.numel())
        # plane-wave sharding (comm_pw): this rank holds the rows [row0, row1) of every orbital block of the
        # k-point (split_evenly over the sphere index); without it the slab is the whole sphere
        pw = basis.comm_pw
        self.row_starts = np.array([r.start for r in split_evenly(self.n_G, pw.size)] + [self.n_G], dtype=np.int64)
        self.row0, self.row1 = int(self.row_starts[pw.rank]), int(self.row_starts[pw.rank + 1])
        self.n_loc = self.row1 - self.row0
        self.kinetic_local = self.kinetic[self.row0:self.row1]
        self.handle = C.c_void_p()
        self._keep = {}
        self._pot_owner = None      # the DftHamiltonianBlock whose potential currently sits in the device handle
        if basis.handle is not None:
            kin_h = np.ascontiguousarray(self.kinetic.cpu().numpy())
            _lib.check(basis.lib.dftk_mi_kblock_create(basis.lane_handles[lane], self.n_G, self.mapping.ctypes.data,
                                                       kin_h.ctypes.data, C.byref(self.handle)))
            if pw.size > 1:
                if self.n_loc < 1:
                    raise ValueError("plane-wave sharding: more ranks than plane waves")
                _lib.check(basis.lib.dftk_mi_kblock_set_shard(self.handle, pw.abi_handle(basis.device.index),
                                                              self.row_starts.ctypes.data))
        # Gamma point: LOBPCG iterates real-symmetric orbitals psi(-G) = conj(psi(G)) in the library's half-sphere
        # format (dftk_mi_kblock_set_gamma_real: real matrix products over half the rows, two bands per FFT pass).
        # An extension over the reference (no Gamma special case there); eigenvalues / density / energies unchanged.
        # Automatic (basis.gamma_real None) only where it pays: a Gamma point inside a lock-step batched k-mesh (kbatch)
        # stays a general complex block and iterates WITH the other k-blocks -- on its own it would be a separate,
        # launch-latency-bound LOBPCG call per SCF step (Al 12^3 mesh: half of the diagonalisation time for 1 of 72 k-points).
        self.gamma_real = False
        auto_ok = basis.gamma_real is True or (basis.gamma_real is None and not getattr(basis, "kbatch", False))
        if (auto_ok and self.handle and not self.coordinate.any()):
            st = basis.lib.dftk_mi_kblock_set_gamma_real(self.handle, 1)
            if st == 0:
                self.gamma_real = True
            elif basis.gamma_real is True:          # explicitly requested: report why it is impossible
                _lib.check(st)

    def __del__(self):
        try:
            if self.handle:
                self.basis.lib.dftk_mi_kblock_destroy(self.handle)
        except Exception:
            pass


COARSE_ECUT_RATIO = 0.25      # companion basis of the two-level start: Ecut / 4 = half the sphere radius, 1 / 8 of the plane waves


class PlaneWaveBasis:
    """``PlaneWaveBasis(model; Ecut, kgrid, fft_size, architecture=GPU, comm_kpts)``.

    ``comm_kpts`` shards the k-points over ranks (as the reference); ``comm_pw`` shards the PLANE WAVES of
    every k-block over ranks (row slabs; what a Gamma-only supercell needs to strong-scale).
    ``device="cuda"`` binds the basis to the MI355X library (required for the hot path);
    ``device="cpu"`` builds the descriptors only (set-up / sharding tests): any hot-path call
    then raises, there is no CPU fallback.
    """

    def __init__(self, model: Model, Ecut: float, kgrid=None, fft_size=None, device="cuda",
                 comm_kpts: KptComm | None = None, build_terms=True, comm_pw: KptComm | None = None,
                 use_symmetries_for_kpoint_reduction=True, n_lanes: int | None = None, gamma_real: bool | None = None,
                 coarse_start: bool | None = None):
        from . import symmetry as _sym
        # gamma_real: None = automatic at k = 0 (env DFTK_MI_GAMMA_REAL=0 switches it off), True = required, False = off
        if gamma_real is None and os.environ.get("DFTK_MI_GAMMA_REAL", "1") == "0":
            gamma_real = False
        self.gamma_real = gamma_real
        self.model = model
        self.Ecut = float(Ecut)
        self.device = torch.device(device)
        self.comm_kpts = comm_kpts if comm_kpts is not None else KptComm.single()
        # comm_pw: ranks that share every k-block of this basis by plane-wave row slabs (Gamma-only cells)
        self.comm_pw = comm_pw if comm_pw is not None else KptComm.single()
        if self.comm_pw.size > 1 and self.comm_kpts.size > 1:
            raise NotImplementedError("k-point and plane-wave sharding cannot be combined yet")
        symmetries_respect_rgrid = fft_size is None                          # PlaneWaveBasis.jl:330
        if fft_size is None:
            # FFT size compatible with the fractional translations of the symmetries (PlaneWaveBasis.jl:349-361)
            factors = (1,)
            if any(not s.isone() for s in model.symmetries):
                from fractions import Fraction
                den = {Fraction(float(x)).limit_denominator(1000).denominator for s in model.symmetries for x in s.w}
                factors = tuple(sorted(den & {2, 3, 4, 6})) or (1,)
            fft_size = compute_fft_size(model, Ecut, factors=factors)
        self.fft_size = tuple(int(n) for n in fft_size)
        nx, ny, nz = self.fft_size
        self.N = nx * ny * nz
        self.dvol = model.unit_cell_volume / self.N
        self.ifft_normalization = 1 / math.sqrt(model.unit_cell_volume)      # fft.jl:87
        self.fft_normalization = math.sqrt(model.unit_cell_volume) / self.N  # fft.jl:88
        self.lib = None
        self.handle = None
        if self.device.type == "cuda":
            self.lib = _lib.load()
            self.handle = C.c_void_p()
            idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
            self.device = torch.device("cuda", idx)
            _lib.check(self.lib.dftk_mi_basis_create(nx, ny, nz, model.unit_cell_volume, idx,
                                                     C.byref(self.handle)))
        # k-points: explicit list or unreduced Monkhorst-Pack mesh, split over comm_kpts
        if kgrid is None:
            kgrid = MonkhorstPack((1, 1, 1))
        # symmetries that survive the discretisation, irreducible k-points (PlaneWaveBasis.jl:161-173)
        symmetries = list(model.symmetries)
        # what todict(basis) reports (input_output.jl:184-196): the grid AS GIVEN and the two constructor flags
        self.symmetries_respect_rgrid = bool(symmetries_respect_rgrid)
        self.use_symmetries_for_kpoint_reduction = bool(use_symmetries_for_kpoint_reduction)
        if isinstance(kgrid, MonkhorstPack):                                   # Base.show, bzmesh.jl:31-37 / :120-122
            shift = [float(x) for x in kgrid.kshift]
            self.kgrid_description = (f"MonkhorstPack([{', '.join(str(int(n)) for n in kgrid.kgrid_size)}]"
                                      + (f", [{', '.join(repr(x) for x in shift)}]" if any(shift) else "") + ")")
        else:
            self.kgrid_description = f"ExplicitKpoints with {len(kgrid.kcoords)} k-points"
        if symmetries_respect_rgrid:
            symmetries = _sym.symmetries_preserving_rgrid(symmetries, self.fft_size)
        if isinstance(kgrid, MonkhorstPack):
            symmetries = _sym.symmetries_preserving_kgrid(symmetries, kgrid.kgrid_size, kgrid.kshift)
            if use_symmetries_for_kpoint_reduction and any(not s.isone() for s in symmetries):
                kc, kw = _sym.irreducible_kcoords(kgrid.kgrid_size, symmetries, kgrid.kshift)
                kgrid = ExplicitKpoints(kc, kw)
            else:
                kgrid = kgrid.reducible()
        else:
            symmetries = _sym.symmetries_preserving_kcoords(symmetries, kgrid.kcoords)
        self.symmetries = symmetries
        (kc, kw, self.kcoords_global, self.kweights_global,
         self.krange_allprocs) = distribute_kpoints(kgrid.kcoords, kgrid.kweights, self.comm_kpts)
        self.krange_thisproc = self.krange_allprocs[self.comm_kpts.rank]
        self.kweights = kw
        # Lanes: k-blocks are tiny on k-point workloads (n_G ~ 1e3, 6-8 bands: launch-latency bound), and the
        # reference loops over them sequentially (diag.jl:24).  Here the local k-points are dealt round-robin onto
        # n_lanes library handles = HIP streams with their own scratch; host threads drive the lanes concurrently
        # (diagonalize_all_kblocks, compute_density), so the small kernels of different k-points overlap on the GPU.
        # kbatch: many small k-blocks iterate in lock-step inside ONE library call (dftk_mi_lobpcg_multi) -- batched
        # launches over all k-points instead of per-k launches overlapped by host threads; needs one handle (lane).
        # DFTK_MI_KBATCH=0 (or an explicit n_lanes) keeps the lane pool.
        # Measured (one MI355X, whole SCF): Al 72 k-points 17 vs 11 SCF it/s batched vs lanes, but Si 8 k-points 65 vs 80 and
        # graphene 12 k-points 20 vs 33 -- a scheduling round costs ~150 us whatever the number of k-blocks in it, so a
        # handful of k-points is better served by as many concurrent streams; DFTK_MI_KBATCH=1 forces the batched loop.
        # Round 6: k-blocks small enough for the library's small-block LOBPCG driver (M <= 8 bands, n_G * M <= 65536: ONE
        # host synchronisation per iteration, dftk_mi_lobpcg_small_stats) batch from two k-points on -- a scheduling round
        # then carries a whole LOBPCG iteration of every k-point: Si 8 k-points 120 -> 207, graphene 12 k-points 33 -> 58 SCF
        # it/s against the lanes (which run the same driver, one fiber per call).
        env_kb = os.environ.get("DFTK_MI_KBATCH")
        kb_min = int(os.environ.get("DFTK_MI_KBATCH_MIN", "32"))
        n_occ = -(-model.n_electrons // (model.n_spin_components * model.filled_occupation))
        m_est = max(n_occ + 3, int(math.ceil(1.2 * n_occ))) if model.temperature > 0 else n_occ + 3
        n_g_est = model.unit_cell_volume * (2.0 * Ecut) ** 1.5 / (6.0 * math.pi ** 2)
        if m_est <= 8 and 1.15 * n_g_est * m_est <= 65536 and os.environ.get("DFTK_MI_LOBPCG_SMALL") != "0":
            kb_min = min(kb_min, 2)
        # The decision is taken from the GLOBAL k-point list: every rank of a k-parallel run must walk the same k-loop
        # algorithm and the same chain of start vectors whatever its local share (9 of 72 k-points per rank at 8 GPUs
        # stay batched, exactly as the one-rank run of the same workload).
        n_k_global = len(self.kcoords_global)
        self.kbatch = (n_lanes is None and env_kb != "0" and "DFTK_MI_LANES" not in os.environ and len(kc) > 1
                       and self.comm_pw.size == 1 and (env_kb == "1" or n_k_global >= kb_min))
        if self.kbatch:
            n_lanes = 1
        if n_lanes is None:
            n_lanes = int(os.environ.get("DFTK_MI_LANES", "16"))
        n_lanes = max(1, min(n_lanes, len(kc))) if (self.handle is not None and self.comm_pw.size == 1) else 1
        self.lane_handles = [self.handle]
        for _ in range(1, n_lanes):
            h = C.c_void_p()
            _lib.check(self.lib.dftk_mi_basis_create(nx, ny, nz, model.unit_cell_volume, self.device.index, C.byref(h)))
            self.lane_handles.append(h)
        self.n_lanes = n_lanes
        self._pool = None
        self.kpoints = [Kpoint(self, k, lane=i % n_lanes) for i, k in enumerate(kc)]
        self.n_kcoords_local = len(kc)
        if model.n_spin_components == 2:
            # collinear spin: all spin-up k-blocks, then all spin-down ones, weights repeated -- they sum to
            # n_spin_components over the ranks (build_kpoints Kpoint.jl:58-74, PlaneWaveBasis.jl:50-53, :218-232)
            self.kpoints = self.kpoints + [Kpoint(self, k, spin=2, lane=(len(kc) + i) % n_lanes) for i, k in enumerate(kc)]
            self.kweights = list(self.kweights) + list(self.kweights)
        # the full cube as a degenerate "sphere": gives hand-written cube FFTs for Hartree etc.
        self._cube_handle = C.c_void_p()
        if self.handle is not None:
            full = np.arange(self.N, dtype=np.int64)
            _lib.check(self.lib.dftk_mi_kblock_create(self.handle, self.N, full.ctypes.data, None,
                                                      C.byref(self._cube_handle)))
        self.terms = None
        if build_terms:
            from .terms import instantiate_terms
            self.terms = instantiate_terms(self)
        # Two-level start of the FIRST diagonalisation (an extension; DFTK_MI_COARSE_START=0 or coarse_start=False switch it
        # off): a companion basis at Ecut / 4 (half the cube) on which the first Hamiltonian is solved from random orbitals,
        # its eigenvectors zero-padded into this basis as start vectors (eigen.py: diagonalize_all_kblocks).  For the large
        # k-blocks only -- the 1000-electron cell: 15-16 LOBPCG iterations from random orbitals (1.2 s, 27 % of a 20-step
        # SCF) become 12 coarse ones (0.25 s) + 3 fine ones (0.35 s); small k-blocks are latency-bound and batch instead.
        self.coarse = None
        if coarse_start is None:
            coarse_start = (os.environ.get("DFTK_MI_COARSE_START", "1") != "0" and build_terms and self.handle is not None
                            and not self.kbatch and min(self.fft_size) >= 96 and model.n_spin_components == 1)
        if coarse_start:
            self.coarse = PlaneWaveBasis(model, self.Ecut * COARSE_ECUT_RATIO, ExplicitKpoints([list(k) for k in kc], list(kw)),
                                         device=self.device, build_terms=True, n_lanes=self.n_lanes,
                                         gamma_real=self.gamma_real, coarse_start=False,
                                         comm_pw=self.comm_pw if self.comm_pw.size > 1 else None)

    # ---- grids ---------------------------------------------------------------------------
    def G_vectors_cube(self):
        nx, ny, nz = self.fft_size
        ax = [torch.tensor(G_axis(n), device=self.device) for n in (nx, ny, nz)]
        gz, gy, gx = torch.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
        return gx, gy, gz

    def G_vectors_cart_cube(self):
        gx, gy, gz = self.G_vectors_cube()
        B = torch.tensor(self.model.recip_lattice, dtype=torch.float64, device=self.device)
        gx, gy, gz = gx.to(torch.float64), gy.to(torch.float64), gz.to(torch.float64)
        return gx[..., None] * B[:, 0] + gy[..., None] * B[:, 1] + gz[..., None] * B[:, 2]

    def enforce_real_mask(self):
        """1 where the -G partner exists on the grid (symmetry.jl:318-337,550-552), else 0."""
        nx, ny, nz = self.fft_size
        mask = torch.ones((nz, ny, nx), dtype=torch.float64, device=self.device)
        for axis, n in ((0, nz), (1, ny), (2, nx)):
            if n % 2 == 0:
                idx = [slice(None)] * 3
                idx[axis] = n // 2          # G = -n/2 sits at FFT index n/2
                mask[tuple(idx)] = 0
        return mask

    # ---- cube FFTs through the library (normalised as fft.jl:106-109,155-161) -----------------
    def _require_gpu(self):
        if self.handle is None:
            raise RuntimeError("PlaneWaveBasis was built with device='cpu': the MI355X hot path is unavailable "
                               "(no CPU fallback)")

    def sync(self, lane=None):
        """Block until the library's stream(s) are idle: one lane, or all of them."""
        self._require_gpu()
        for h in (self.lane_handles if lane is None else [self.lane_handles[lane]]):
            _lib.check(self.lib.dftk_mi_basis_sync(h))

    # ---- stream discipline of the host mirror ----------------------------------------------------------------------
    # The library works on ITS stream(s); torch on its current stream.  Round 6: ``on_library_stream()`` makes the
    # library's stream torch's current stream (an ExternalStream) for a block of host code -- the SCF drivers run inside it
    # -- so that torch kernels and library kernels are ordered by ONE stream and the mirror's calls need no host
    # synchronisation around them (they were ~30 hipStreamSynchronize per SCF step of the k-point workloads).  Outside such
    # a block, or when several lanes (streams) are in use, the two hooks below synchronise as before.
    def on_library_stream(self):
        self._require_gpu()
        if getattr(self, "_ext_stream", None) is None:
            self._ext_stream = torch.cuda.ExternalStream(int(self.stream_ptr), device=self.device)
        return torch.cuda.stream(self._ext_stream)

    def _same_stream(self):
        ext = getattr(self, "_ext_stream", None)
        return (ext is not None and self.n_lanes == 1
                and torch.cuda.current_stream(self.device).cuda_stream == ext.cuda_stream)

    def pre_call(self):
        """Before a library call: its inputs, produced by torch, must be complete on the library's stream."""
        if not self._same_stream():
            torch.cuda.current_stream(self.device).synchronize()

    def post_call(self, lane=None):
        """After a library call whose outputs torch consumes next."""
        if not self._same_stream():
            self.sync(lane)

    def run_on_lanes(self, fn, items):
        """``[fn(i, item) for i, item in enumerate(items)]`` with item i executed by the host thread of lane
        ``kpoints[i].lane`` (items of one lane in order, lanes concurrently).  ctypes releases the GIL inside the
        library calls, which is where the time goes."""
        items = list(items)
        if self.n_lanes == 1 or len(items) <= 1:
            return [fn(i, it) for i, it in enumerate(items)]
        from concurrent.futures import ThreadPoolExecutor
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=self.n_lanes, thread_name_prefix="dftk-lane")
        out = [None] * len(items)

        def work(lane):
            torch.cuda.set_device(self.device)
            for i, it in enumerate(items):
                if self.kpoints[i].lane == lane:
                    out[i] = fn(i, it)
        futures = [self._pool.submit(work, lane) for lane in range(self.n_lanes)]
        for f in futures:
            f.result()
        return out

    @property
    def stream_ptr(self):
        """hipStream_t of the library's stream (collectives of the host mirror are enqueued on it)."""
        self._require_gpu()
        return self.lib.dftk_mi_basis_stream(self.handle)

    def fft(self, f_real: torch.Tensor) -> torch.Tensor:
        """cube -> Fourier coefficients (c_G = sqrt(Omega)/N sum_r f(r) e^{-iG.r})."""
        self._require_gpu()
        f = f_real.to(torch.complex128).contiguous()
        out = torch.empty_like(f)
        self.pre_call()
        _lib.check(self.lib.dftk_mi_fft_sphere(self._cube_handle, f.data_ptr(), out.data_ptr()))
        self.post_call()
        return out * self.fft_normalization

    def ifft(self, f_fourier: torch.Tensor) -> torch.Tensor:
        self._require_gpu()
        f = f_fourier.to(torch.complex128).contiguous()
        out = torch.empty_like(f)
        self.pre_call()
        _lib.check(self.lib.dftk_mi_ifft_sphere(self._cube_handle, f.data_ptr(), out.data_ptr()))
        self.post_call()
        return out * self.ifft_normalization

    def irfft(self, f_fourier: torch.Tensor) -> torch.Tensor:
        return self.ifft(f_fourier).real.contiguous()

    def __del__(self):
        try:
            if self.handle is not None:
                self.kpoints = []
                if self._cube_handle:
                    self.lib.dftk_mi_kblock_destroy(self._cube_handle)
                if self._pool is not None:
                    self._pool.shutdown(wait=True)
                for h in self.lane_handles[::-1]:
                    self.lib.dftk_mi_basis_destroy(h)
        except Exception:
            pass
