"""ctypes binding of the C ABI (include/dftk_mi355x.h).

This is the Python twin of the Julia ``ccall`` shim in INTEGRATION.md.  There is no CPU
fallback: if the shared library is missing it is built with hipcc, and if that is impossible
(or no GPU is visible at call time) the error is raised to the caller.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _build

c_cplx_p = C.c_void_p          # device/host pointers travel as void*
_i64 = C.c_int64


class dftk_mi_cplx(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


class DftkMiError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"dftk_mi355x status {status}: {msg}")
        self.status = status


_SIGNATURES = {
    # name: (restype, argtypes)
    "dftk_mi_last_error": (C.c_char_p, []),
    "dftk_mi_version": (C.c_char_p, []),
    "dftk_mi_basis_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_void_p)]),
    "dftk_mi_basis_destroy": (C.c_int, [C.c_void_p]),
    "dftk_mi_basis_sync": (C.c_int, [C.c_void_p]),
    "dftk_mi_basis_set_fft_batch": (C.c_int, [C.c_void_p, C.c_int]),
    "dftk_mi_basis_stream": (C.c_void_p, [C.c_void_p]),
    "dftk_mi_kblock_create": (C.c_int, [C.c_void_p, _i64, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "dftk_mi_kblock_destroy": (C.c_int, [C.c_void_p]),
    "dftk_mi_kblock_set_projectors": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p]),
    "dftk_mi_kblock_set_potential": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dftk_mi_apply_H": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p, _i64]),
    "dftk_mi_apply_H_parts": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, _i64, C.c_void_p, _i64]),
    "dftk_mi_local_potential": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p]),
    "dftk_mi_local_potential_collinear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                    C.c_void_p]),
    "dftk_mi_density_accumulate_spin": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p, C.c_void_p, C.c_int,
                                                  C.c_int]),
    "dftk_mi_kpoint_sphere_host": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, _i64,
                                             C.POINTER(_i64), C.c_void_p, C.c_void_p, C.c_void_p]),
    "dftk_mi_build_projectors_hgh": (C.c_int, [C.c_void_p, _i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _i64,
                                               C.POINTER(C.c_int)]),
    "dftk_mi_atomic_superposition": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_void_p]),
    "dftk_mi_xc_gga": (C.c_int, [C.c_void_p, _i64, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "dftk_mi_ifft_sphere": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dftk_mi_fft_sphere": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dftk_mi_density_accumulate": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p, C.c_void_p]),
    "dftk_mi_lobpcg": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_double, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(C.c_int),
                                 C.POINTER(C.c_int), C.POINTER(_i64)]),
    "dftk_mi_local_potential_gga": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_double, C.c_void_p, C.c_void_p]),
    "dftk_mi_symmetrize_rho": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "dftk_mi_mix_kerker": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]),
    "dftk_mi_mix_dielectric": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "dftk_mi_chi0_dielectric_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "dftk_mi_cube_fourier_filter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dftk_mi_kblocks_set_potential": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "dftk_mi_lobpcg_multi": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "dftk_mi_density_accumulate_multi": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p]),
    "dftk_mi_density_accumulate_multi2": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "dftk_mi_band_kinetic_multi": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dftk_mi_batch_stats": (C.c_int, [C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "dftk_mi_lobpcg_small_stats": (C.c_int, [C.POINTER(_i64), C.POINTER(_i64)]),
    "dftk_mi_ortho_small": (C.c_int, [C.c_void_p, _i64, C.c_int, C.c_void_p, _i64, C.c_int, C.c_void_p, _i64, C.c_void_p,
                                      C.c_double, C.c_void_p]),
    "dftk_mi_lobpcg_last_AX": (C.c_void_p, [C.c_void_p]),
    "dftk_mi_lobpcg_history": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_int)]),
    "dftk_mi_columnwise_norms": (C.c_int, [C.c_void_p, _i64, C.c_int, C.c_void_p, _i64, C.c_void_p]),
    "dftk_mi_columnwise_dots": (C.c_int, [C.c_void_p, _i64, C.c_int, C.c_void_p, _i64, C.c_void_p, _i64, C.c_void_p]),
    "dftk_mi_ortho_qr": (C.c_int, [C.c_void_p, _i64, C.c_int, C.c_void_p, _i64, C.c_int, C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]),
    "dftk_mi_tpa_precondprep": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p]),
    "dftk_mi_tpa_ldiv": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p, C.c_double, C.c_void_p, _i64]),
    "dftk_mi_block_residual": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p, _i64, C.c_void_p,
                                         C.c_void_p, _i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dftk_mi_kblock_set_gamma_real": (C.c_int, [C.c_void_p, C.c_int]),
    "dftk_mi_gamma_half_size": (C.c_int, [C.c_void_p, C.POINTER(_i64)]),
    "dftk_mi_gamma_tables_host": (C.c_int, [C.c_int, C.c_int, C.c_int, _i64, C.c_void_p, C.POINTER(_i64), C.c_void_p,
                                            C.c_void_p]),
    "dftk_mi_gamma_compress": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p, _i64]),
    "dftk_mi_gamma_compress_aligned": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p, _i64]),
    "dftk_mi_gamma_expand": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p, _i64]),
    "dftk_mi_gamma_apply_H": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, _i64, C.c_void_p, _i64]),
    "dftk_mi_density_accumulate_real": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p, C.c_void_p]),
    "dftk_mi_shard_plan_host": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "dftk_mi_kblock_set_shard": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dftk_mi_zgemm": (C.c_int, [C.c_void_p, C.c_char, _i64, _i64, _i64, dftk_mi_cplx, C.c_void_p, _i64,
                                C.c_void_p, _i64, dftk_mi_cplx, C.c_void_p, _i64]),
    "dftk_mi_zgemm_ex": (C.c_int, [C.c_void_p, C.c_char, _i64, _i64, _i64, dftk_mi_cplx, C.c_void_p, _i64,
                                   C.c_void_p, _i64, dftk_mi_cplx, C.c_void_p, _i64, C.c_int]),
    "dftk_mi_zgemm_plan_host": (C.c_int, [C.c_char, _i64, _i64, _i64, C.c_int, C.POINTER(C.c_int)]),
    "dftk_mi_heev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p, C.c_void_p, _i64]),
    "dftk_mi_prof_zgemm_shapes": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "dftk_mi_heev_lowest": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, _i64, C.c_void_p, C.c_void_p, _i64]),
    "dftk_mi_heev_sigma_host": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_int), C.c_double]),
    "dftk_mi_potrf_trtri": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p, _i64]),
    "dftk_mi_potrf_trtri_real": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p, _i64]),
    "dftk_mi_comm_get_unique_id": (C.c_int, [C.c_char_p]),
    "dftk_mi_comm_init_rank": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dftk_mi_comm_create_host": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_void_p)]),
    "dftk_mi_comm_destroy": (C.c_int, [C.c_void_p]),
    "dftk_mi_comm_describe": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dftk_mi_comm_rank": (C.c_int, [C.c_void_p]),
    "dftk_mi_comm_size": (C.c_int, [C.c_void_p]),
    "dftk_mi_allreduce_sum_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dftk_mi_prof_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "dftk_mi_prof_get": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(_i64)]),
    "dftk_mi_anderson_create": (C.c_int, [C.c_void_p, _i64, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]),
    "dftk_mi_anderson_destroy": (C.c_int, [C.c_void_p]),
    "dftk_mi_anderson_reset": (C.c_int, [C.c_void_p]),
    "dftk_mi_anderson_history": (C.c_int, [C.c_void_p]),
    "dftk_mi_anderson_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "dftk_mi_chi0_mix": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_double,
                                   C.c_double, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]),
    "dftk_mi_launch_count": (C.c_int, [C.POINTER(_i64), C.POINTER(_i64)]),
    "dftk_mi_kblock_reuse_AX": (C.c_int, [C.c_void_p, C.c_int]),
    "dftk_mi_ax_reuse_count": (C.c_int, [C.POINTER(_i64)]),
    "dftk_mi_step_sums": (C.c_int, [C.c_void_p, _i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dftk_mi_fermi_bisection": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                          C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double)]),
    "dftk_mi_diag_mfma_peak": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "dftk_mi_jacobi_schedule_host": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                               C.POINTER(C.c_int)]),
    "dftk_mi_fft_plan_host": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dftk_mi_sphere_tables_host": (C.c_int, [C.c_int, C.c_int, C.c_int, _i64, C.c_void_p, C.POINTER(_i64),
                                             C.POINTER(C.c_int), C.c_void_p, C.c_void_p]),
}

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                           C.POINTER(C.c_double), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t))

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
_lib = None


def load(build_if_missing: bool = True):
    """Load libdftk_mi355x.so (building it with hipcc when absent or stale)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIBPATH
    if build_if_missing and _build.needs_build():
        _build.build()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing and could not be built; the MI355X hot path has no fallback")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)    # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    version = lib.dftk_mi_version().decode()
    if not version.endswith("src=" + _build.source_hash()):
        raise RuntimeError(f"{path} was built from other sources ({version!r}, expected src={_build.source_hash()}): "
                           "stale binary, rebuild with python -m dftk_jl_amd._build")
    _lib = lib
    return lib


def check(status: int):
    if status != 0:
        msg = load().dftk_mi_last_error().decode(errors="replace")
        raise DftkMiError(status, msg)


def cplx(z) -> dftk_mi_cplx:
    z = complex(z)
    return dftk_mi_cplx(z.real, z.imag)
