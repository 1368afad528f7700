"""Memory statistics and a per-GPU plan check (host only: no GPU needed).

* ``estimate_memory_usage`` mirrors the reference's ``estimate_memory_usage`` / ``MemoryStatistics``
  (src/memory_usage.jl:11-87): same fields, same "scf_peak" rule of thumb (1x projectors, 2x psi + 6 psi_k, 12x rho),
  for the REFERENCE's layout (dense complex orbitals, dense projector matrix, one process).
* ``plan_planewave_sharded`` answers the question SURVEY appendix B leaves open for the literal BASELINE configs[4]
  string (Si 8x8x8 = 1024 atoms, 4096 electrons; 463 GB of LOBPCG work arrays in the reference's layout): what THIS
  library allocates per GPU when the plane waves of the single Gamma k-block are sharded as row slabs over ``n_ranks``
  GPUs, item by item from the allocation formulas of the C++ side, and whether it fits the HBM
  of an MI355X.  Nothing is allocated; the sphere is counted exactly by the library's host routine
  (``dftk_mi_kpoint_sphere_host``, Kpoint.jl:20-41).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import asdict, dataclass

import numpy as np

HBM_BYTES_MI355X = 288e9
CPLX, F64 = 16, 8


@dataclass
class MemoryStatistics:
    """src/memory_usage.jl:11-22 (same field names; psi -> ψ, rho -> ρ)."""
    n_kpoints: int
    n_Gk: int
    n_bands: int
    n_nonlocal_projectors: int
    psik_bytes: int
    nonlocal_Pk_bytes: int
    psi_bytes: int
    rho_bytes: int
    nonlocal_P_bytes: int
    scf_peak_bytes: int


def _sphere_size(fft_size, recip_lattice, kcoord, Ecut) -> int:
    from . import _lib
    lib = _lib.load()
    n = C.c_int64(0)
    B = np.asfortranarray(np.asarray(recip_lattice, dtype=np.float64))
    k = np.ascontiguousarray(np.asarray(kcoord, dtype=np.float64))
    _lib.check(lib.dftk_mi_kpoint_sphere_host(int(fft_size[0]), int(fft_size[1]), int(fft_size[2]),
                                              B.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p), float(Ecut), 0,
                                              C.byref(n), None, None, None))
    return int(n.value)


def _n_projectors(model) -> int:
    """count_n_proj summed over the atoms (memory_usage.jl:61-71; NormConservingPsp.jl:187-234)."""
    n = 0
    for atom in model.atoms:
        psp = getattr(atom, "psp", None)
        if psp is not None:
            n += sum((2 * l + 1) * int(np.asarray(psp.h[l]).shape[0]) for l in range(len(psp.h)))
    return n


def _fft_size(model, Ecut, fft_size):
    from .basis import compute_fft_size
    return tuple(int(x) for x in (fft_size if fft_size is not None else compute_fft_size(model, Ecut)))


def estimate_memory_usage(model, Ecut, kcoords=((0.0, 0.0, 0.0),), fft_size=None, n_ranks=1) -> MemoryStatistics:
    """``estimate_memory_usage(model; kgrid, Ecut)`` (src/memory_usage.jl:35-87) for an explicit k-list; ``n_ranks``
    plays the role of the MPI size (k-points per process = ceil(n_k / n_ranks), as ``krange_allprocs``)."""
    from .scf import AdaptiveBands
    fft = _fft_size(model, Ecut, fft_size)
    n_kpoints = -(-len(kcoords) // n_ranks)
    n_Gk = _sphere_size(fft, model.recip_lattice, kcoords[0], Ecut)
    n_bands = AdaptiveBands(model).n_bands_compute
    psik = CPLX * n_Gk * n_bands
    psi = psik * n_kpoints
    rho = F64 * int(np.prod(fft)) * getattr(model, "n_spin_components", 1)
    n_p = _n_projectors(model)
    Pk = CPLX * n_Gk * n_p
    P = Pk * n_kpoints
    return MemoryStatistics(n_kpoints, n_Gk, n_bands, n_p, psik, Pk, psi, rho, P, P + 2 * psi + 6 * psik + 12 * rho)


def plan_planewave_sharded(model, Ecut, n_ranks, fft_size=None, gamma_real=True, fft_batch=32,
                           hbm_bytes=HBM_BYTES_MI355X) -> dict:
    """Per-GPU bytes of a Gamma-only SCF whose single k-block is plane-wave sharded over ``n_ranks`` GPUs.

    Items (formulas = the allocations of the library; M = n_bands_compute, m3 = 3 M, rows = this rank's rows of
    every n_G-sized array: n_half / p in the real-symmetric half-sphere format, n_G / p otherwise):
      lobpcg_blocks     14 blocks of rows x M complex        (lobpcg.cpp: 2 x Y(3), 2 x AY(3), newR, tmp)
      lobpcg_small      Gram / Ritz / Cholesky matrices       (lobpcg.cpp ``small_elems``: replicated on every rank)
      heev_workspace    3 padded (3M)^2 complex-sized work matrices + 2 rotation buffers (dense_kernels.hip heev_impl)
      projectors        slab of P in the format the products use (+ the caller's full-format slab it is built from)
      orbitals_caller   psi handed in / out by the host mirror (full-sphere rows x M complex) + LOBPCG's copy
      transposer        3 x n_G x ceil(M / p) complex: slab <-> band all-to-all buffers (api.cpp shard_buffers)
      gamma_pack        2 x n_G x ceil(M / p / 2) complex: pair-packed whole bands (gamma_kernels.hip)
      fft_scratch       fft_batch x (T1 + T2)                 (fft_kernels.hip fft_ensure_scratch)
      gemm_split_k      <= 1 GiB of split-K slabs             (gemm_kernels.hip)
      cubes             rho, V and its parts, Anderson history of 10 (x, r) pairs: 26 real cubes
    Also returned: the per-step communication volumes of one H psi sweep and one Rayleigh-Ritz Gram all-reduce."""
    from .scf import AdaptiveBands
    p = int(n_ranks)
    fft = _fft_size(model, Ecut, fft_size)
    N = int(np.prod(fft))
    nxp = -(-fft[0] // 8) * 8
    n_G = _sphere_size(fft, model.recip_lattice, (0.0, 0.0, 0.0), Ecut)
    M = AdaptiveBands(model).n_bands_compute
    n_p = _n_projectors(model)
    n_half = (n_G + 1) // 2
    rows_fmt = -(-(n_half if gamma_real else n_G) // p)          # rows of the iteration's format on the fullest rank
    rows_full = -(-n_G // p)
    m3 = 3 * M
    # sphere geometry of the pruned pipeline: ~pi/4 of the (y, z) lines inside the bounding square, half of the z planes
    n_lines = int(math.ceil(math.pi / 4 * (fft[1] / 2 + 1) * (fft[2] / 2 + 1)))
    nzx = fft[2] // 2 + 2
    bands_rank = -(-M // p)
    nb_heev = -(-m3 // 16)
    np_heev = 16 * (nb_heev + nb_heev % 2)          # heev_impl pads to an even number of 16-wide blocks
    items = {
        "lobpcg_blocks": 14 * rows_fmt * M * CPLX,
        "lobpcg_small": (m3 * m3 * 2 + m3 * M * 2 + M * M * 4 + (2 * M + m3) * (M + 1)) * CPLX,
        "heev_workspace": 3 * np_heev ** 2 * CPLX + 2 * (np_heev // 32) * 32 * 32 * CPLX,
        "projectors": rows_fmt * n_p * CPLX + rows_full * n_p * CPLX,
        "orbitals_caller": 2 * rows_full * M * CPLX,
        "transposer": (3 * n_G * bands_rank * CPLX) if p > 1 else 0,
        "gamma_pack": (2 * n_G * -(-bands_rank // 2) * CPLX) if gamma_real else 0,
        "fft_scratch": fft_batch * (n_lines * nxp + nzx * fft[1] * nxp) * CPLX,
        "gemm_split_k": 1 << 30,
        "cubes": 26 * N * F64,
    }
    total = int(sum(items.values()))
    band_block = CPLX * (n_half if gamma_real else n_G) * M          # one block of all bands, iteration format
    comm = {
        "alltoall_bytes_sent_per_rank_per_Hpsi_sweep": int(2 * band_block / p * (p - 1) / p) if p > 1 else 0,
        "gram_allreduce_bytes": int(m3 * m3 * CPLX),
        "rho_allreduce_bytes": N * F64,
    }
    ref = estimate_memory_usage(model, Ecut, fft_size=fft)
    return {"n_ranks": p, "fft_size": fft, "n_G": n_G, "n_half": n_half, "n_bands": M, "n_projectors": n_p,
            "rows_per_rank": rows_fmt, "bytes_per_rank": {k: int(v) for k, v in items.items()}, "total_bytes_per_rank": total,
            "hbm_bytes": int(hbm_bytes), "fits": bool(total <= 0.92 * hbm_bytes), "headroom_fraction": 1.0 - total / hbm_bytes,
            "communication": comm,
            "reference_layout": {**asdict(ref), "note": "src/memory_usage.jl:74-81 rule of thumb, one process, dense complex "
                                                        "orbitals: what the reference itself would need"},
            "limits": {"fft_axis_lds_ok": bool(max(fft) * 9 * CPLX <= 160 * 1024),
                       "register_resident_z_kernels": bool(64 <= fft[2] <= 256),
                       "note": "axes longer than 256 take the LDS-pass z kernels (slower, same results); the replicated "
                               "Rayleigh-Ritz eigensolver works on a (3M)^2 matrix on every rank"}}


def format_plan(plan: dict) -> str:
    lines = [f"plane-wave sharded Gamma block on {plan['n_ranks']} GPU(s): fft {plan['fft_size']}, n_G {plan['n_G']}, "
             f"{plan['n_bands']} bands, {plan['n_projectors']} projectors, {plan['rows_per_rank']} rows per rank"]
    for k, v in plan["bytes_per_rank"].items():
        lines.append(f"  {k:18s} {v / 1e9:9.2f} GB")
    lines.append(f"  {'TOTAL':18s} {plan['total_bytes_per_rank'] / 1e9:9.2f} GB of {plan['hbm_bytes'] / 1e9:.0f} GB -> "
                 f"{'fits' if plan['fits'] else 'DOES NOT FIT'}")
    return "\n".join(lines)
