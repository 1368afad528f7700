"""``compute_density`` (src/densities.jl:13-57): per-band pruned iFFT + |psi|^2 accumulation on the
device, one RCCL all-reduce over ``comm_kpts``, then the symmetrisation over ``basis.symmetries``
(symmetry.jl:346-357; a no-op for ``symmetries=False`` / Gamma-only supercells)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def compute_density(basis, psi, occupation, occupation_threshold: float = 0.0, real_symmetric=None, extra_weights=None,
                    extra_threshold: float = 0.0):
    """``real_symmetric`` (list of bools per k-point, from ``diagonalize_all_kblocks``): the orbitals of that k-point
    satisfy psi(-G) = conj(psi(G)) -- two bands then share one transform (dftk_mi_density_accumulate_real).
    ``extra_weights`` (per k-point band weights, e.g. those of ``compute_ldos``, dos.jl:43-62 -- "compute_density with modified
    weights"): a SECOND cube accumulated with them; returns ``(rho, rho_extra)``.  For many small k-blocks both come out of
    ONE pass over the bands (``dftk_mi_density_accumulate_multi2``), otherwise of two."""
    if extra_weights is not None:
        two = (getattr(basis, "kbatch", False) and basis.n_lanes == 1 and len(basis.kpoints) > 1
               and basis.model.n_spin_components == 1 and not (real_symmetric is not None and any(real_symmetric))
               and basis.comm_pw.size == 1)
        if not two:
            return (compute_density(basis, psi, occupation, occupation_threshold, real_symmetric),
                    compute_density(basis, psi, extra_weights, extra_threshold, real_symmetric))
    basis._require_gpu()
    nx, ny, nz = basis.fft_size
    n_spin = basis.model.n_spin_components
    # one accumulator per lane (the lanes run concurrently on their own streams), summed in lane order afterwards;
    # rho[kpt.spin - 1] takes the bands of a k-block (rho[:, :, :, kpt.spin], densities.jl:29, :39)
    rhos = [torch.zeros((n_spin, nz, ny, nx), dtype=torch.float64, device=basis.device) for _ in range(basis.n_lanes)]
    rho2 = torch.zeros((nz, ny, nx), dtype=torch.float64, device=basis.device) if extra_weights is not None else None
    basis.pre_call()

    if getattr(basis, "kbatch", False) and basis.n_lanes == 1 and len(basis.kpoints) > 1:
        # many small k-blocks: the bands of all of them go through ONE pipeline (dftk_mi_density_accumulate_multi);
        # k-points whose orbitals are real-symmetric keep their paired call
        import ctypes as C
        multi_all = [ik for ik in range(len(basis.kpoints)) if not (real_symmetric is not None and real_symmetric[ik])]
        done_multi = set()
        for spin in range(1, n_spin + 1):          # one batched pipeline per spin channel (its own density cube)
            multi = [ik for ik in multi_all if basis.kpoints[ik].spin == spin]
            if len(multi) < 2:
                continue
            ws, ws2, keep = [], [], []
            for ik in multi:
                occ = np.asarray(occupation[ik], dtype=np.float64)
                ws.append(np.where(np.abs(occ) >= occupation_threshold, occ, 0.0) * basis.kweights[ik]
                          * basis.ifft_normalization ** 2)
                if rho2 is not None:
                    w2 = np.asarray(extra_weights[ik], dtype=np.float64)
                    if len(w2) != len(occ):
                        raise ValueError("compute_density: extra_weights must have one entry per band")
                    ws2.append(np.where(np.abs(w2) >= extra_threshold, w2, 0.0) * basis.kweights[ik]
                               * basis.ifft_normalization ** 2)
                psik = psi[ik]
                if not (psik.is_cuda and psik.dtype == torch.complex128 and psik.stride(1) == 1):
                    raise TypeError("compute_density: complex128 CUDA band-major blocks required")
                keep.append(psik)
            n = len(multi)
            w_all = np.ascontiguousarray(np.concatenate(ws))
            kbs = (C.c_void_p * n)(*[basis.kpoints[ik].handle.value for ik in multi])
            nbs = (C.c_int * n)(*[len(w_) for w_ in ws])
            pp = (C.c_void_p * n)(*[p_.data_ptr() for p_ in keep])
            ld = (C.c_int64 * n)(*[p_.stride(0) for p_ in keep])
            if rho2 is not None:
                w2_all = np.ascontiguousarray(np.concatenate(ws2))
                _lib.check(basis.lib.dftk_mi_density_accumulate_multi2(n, kbs, nbs, pp, ld, w_all.ctypes.data,
                                                                       rhos[0][spin - 1].data_ptr(), w2_all.ctypes.data,
                                                                       rho2.data_ptr()))
            else:
                _lib.check(basis.lib.dftk_mi_density_accumulate_multi(n, kbs, nbs, pp, ld, w_all.ctypes.data,
                                                                      rhos[0][spin - 1].data_ptr()))
            done_multi |= set(multi)
        if rho2 is not None and done_multi != set(range(len(basis.kpoints))):
            raise RuntimeError("compute_density: the one-pass two-weight form needs every k-point in the batched pipeline")
    else:
        done_multi = set()

    def accumulate(ik, kpt):
        if ik in done_multi:
            return
        occ = np.asarray(occupation[ik], dtype=np.float64)          # occupations live on the host (:16)
        w = np.where(np.abs(occ) >= occupation_threshold, occ, 0.0) * basis.kweights[ik] * basis.ifft_normalization ** 2
        w = np.ascontiguousarray(w)
        psik = psi[ik]
        if not (psik.is_cuda and psik.dtype == torch.complex128 and psik.stride(1) == 1):
            raise TypeError("compute_density: complex128 CUDA band-major blocks required")
        paired = real_symmetric is not None and real_symmetric[ik]
        if paired:
            _lib.check(basis.lib.dftk_mi_density_accumulate_real(kpt.handle, len(w), psik.data_ptr(), psik.stride(0),
                                                                 w.ctypes.data, rhos[kpt.lane][kpt.spin - 1].data_ptr()))
        else:
            _lib.check(basis.lib.dftk_mi_density_accumulate_spin(kpt.handle, len(w), psik.data_ptr(), psik.stride(0),
                                                                 w.ctypes.data, rhos[kpt.lane].data_ptr(), kpt.spin - 1,
                                                                 n_spin))
    basis.run_on_lanes(accumulate, basis.kpoints)
    rho = rhos[0]
    if basis.n_lanes > 1:
        basis.post_call()
        for r in rhos[1:]:
            rho += r
        basis.pre_call()
    # mpi_sum!(rho, comm_kpts) (:46), enqueued on the library's stream behind the accumulation kernels; with
    # plane-wave sharding every rank has accumulated its share of the BANDS: the same all-reduce over comm_pw
    for comm in (basis.comm_pw, basis.comm_kpts):
        if comm.size > 1:
            comm.sum_(rho, basis.stream_ptr)
            if rho2 is not None:
                comm.sum_(rho2, basis.stream_ptr)
    basis.post_call()
    if any(not s.isone() for s in basis.symmetries):
        from .symmetry import symmetrize_rho
        rho = torch.stack([symmetrize_rho(basis, r, do_lowpass=False) for r in rho])   # densities.jl:47 (per spin)
        if rho2 is not None:
            rho2 = symmetrize_rho(basis, rho2, do_lowpass=False)
    out = rho if n_spin == 2 else rho[0]
    return out if rho2 is None else (out, rho2)
