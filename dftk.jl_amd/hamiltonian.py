"""``DftHamiltonianBlock`` and ``mul_`` -- host mirror of src/terms/Hamiltonian.jl:22-57,137-192.

The block owns no arrays: kinetic multiplier, sphere tables and projectors live in the k-block
handle of the device library; constructing a block uploads the summed local potential
(operators.jl:213-222) into it, exactly like the reference rebuilds its blocks every SCF step.
"""
from __future__ import annotations

import torch

from . import _lib


class DftHamiltonianBlock:
    def __init__(self, basis, kpoint, potential):
        basis._require_gpu()
        self.basis, self.kpoint = basis, kpoint
        self.potential = None
        if potential is not None:
            self.potential = potential.to(torch.float64).contiguous()
            torch.cuda.current_stream(basis.device).synchronize()
            _lib.check(basis.lib.dftk_mi_kblock_set_potential(kpoint.handle, self.potential.data_ptr()))
        else:
            _lib.check(basis.lib.dftk_mi_kblock_set_potential(kpoint.handle, None))

    @property
    def n_G(self):
        return self.kpoint.n_G

    def size(self):
        return (self.kpoint.n_G, self.kpoint.n_G)

    def mul_(self, Hpsi: torch.Tensor, psi: torch.Tensor, which: int = 7) -> torch.Tensor:
        """``mul!(Hpsi, H, psi)``: psi, Hpsi are (n_bands, ld >= n_G) complex128 CUDA tensors whose rows
        are bands (column-major n_G x n_bands)."""
        if not (psi.is_cuda and Hpsi.is_cuda and psi.dtype == torch.complex128 and Hpsi.dtype == torch.complex128):
            raise TypeError("mul_: complex128 CUDA tensors required (the hot path has no CPU fallback)")
        if psi.dim() != 2 or psi.stride(1) != 1 or Hpsi.stride(1) != 1:
            raise ValueError("mul_: band-major contiguous blocks required")
        nb = psi.shape[0]
        torch.cuda.current_stream(self.basis.device).synchronize()
        _lib.check(self.basis.lib.dftk_mi_apply_H_parts(self.kpoint.handle, which, nb, psi.data_ptr(), psi.stride(0),
                                                        Hpsi.data_ptr(), Hpsi.stride(0)))
        self.basis.sync()
        return Hpsi

    def __matmul__(self, psi):
        return self.mul_(torch.empty_like(psi), psi)


def mul_(Hpsi, H: DftHamiltonianBlock, psi):
    """Free-function spelling of ``mul!(Hpsi, H, psi)``."""
    return H.mul_(Hpsi, psi)
