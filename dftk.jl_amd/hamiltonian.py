"""``DftHamiltonianBlock`` and ``mul_`` -- host mirror of src/terms/Hamiltonian.jl:22-57,137-192.

Kinetic multiplier, sphere tables and projectors live in the k-block handle of the device library; the
block OWNS its summed local potential (operators.jl:213-222) like the reference's block owns its operators.
The device handle holds one padded copy of a potential at a time: every entry point that applies H
(``mul_``, ``lobpcg_hyper``) calls ``bind()``, which re-uploads this block's potential when another block
of the same k-point (an older / newer Hamiltonian that is still alive) was the last one bound.
"""
from __future__ import annotations

import torch

from . import _lib


class DftHamiltonianBlock:
    def __init__(self, basis, kpoint, potential, bind: bool = True):
        basis._require_gpu()
        self.basis, self.kpoint = basis, kpoint
        self.potential = potential.to(torch.float64).contiguous() if potential is not None else None
        if bind:
            self.bind(force=True)

    @staticmethod
    def for_all_kpoints(basis, potential):
        """The blocks of every local k-point for ONE summed potential (Hamiltonian.jl:36-57), bound with a single library
        call (``dftk_mi_kblocks_set_potential``) instead of one host round trip per k-point."""
        import ctypes as C
        if potential is not None and potential.dim() == 4:
            # collinear spin: potential[s] belongs to the k-blocks of spin s + 1 (xc.jl:163-175: Vxc[:, :, :, kpt.spin])
            pots = [potential[s_].to(torch.float64).contiguous() for s_ in range(potential.shape[0])]
            blocks = [DftHamiltonianBlock(basis, kpt, pots[kpt.spin - 1], bind=False) for kpt in basis.kpoints]
            basis.pre_call()
            for s_, pot_s in enumerate(pots):
                mine = [b_ for b_ in blocks if b_.kpoint.spin == s_ + 1]
                kbs = (C.c_void_p * len(mine))(*[b_.kpoint.handle.value for b_ in mine])
                _lib.check(basis.lib.dftk_mi_kblocks_set_potential(len(mine), kbs, pot_s.data_ptr()))
            for b_ in blocks:
                b_.kpoint._pot_owner = b_
            return blocks
        pot = potential.to(torch.float64).contiguous() if potential is not None else None
        blocks = [DftHamiltonianBlock(basis, kpt, pot, bind=False) for kpt in basis.kpoints]
        if pot is None or len(blocks) < 2:
            for b_ in blocks:
                b_.bind(force=True)
            return blocks
        n = len(blocks)
        kbs = (C.c_void_p * n)(*[b_.kpoint.handle.value for b_ in blocks])
        basis.pre_call()
        _lib.check(basis.lib.dftk_mi_kblocks_set_potential(n, kbs, pot.data_ptr()))
        for b_ in blocks:
            b_.kpoint._pot_owner = b_
        return blocks

    def bind(self, force: bool = False):
        """Make the device handle of the k-point apply THIS block's potential."""
        kpt = self.kpoint
        if not force and kpt._pot_owner is self:
            return
        if self.potential is not None:
            self.basis.pre_call()
            _lib.check(self.basis.lib.dftk_mi_kblock_set_potential(kpt.handle, self.potential.data_ptr()))
        else:
            _lib.check(self.basis.lib.dftk_mi_kblock_set_potential(kpt.handle, None))
        kpt._pot_owner = self

    @property
    def n_G(self):
        return self.kpoint.n_G

    @property
    def n_loc(self):
        """Rows of an orbital block on this rank (== n_G unless the basis shards plane waves)."""
        return self.kpoint.n_loc

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
        if psi.shape[1] != self.n_loc or Hpsi.shape != psi.shape:
            raise ValueError(f"mul_: blocks must be (n_bands, {self.n_loc})")
        self.bind()
        self.basis.pre_call()
        _lib.check(self.basis.lib.dftk_mi_apply_H_parts(self.kpoint.handle, which, nb, psi.data_ptr(), psi.stride(0),
                                                        Hpsi.data_ptr(), Hpsi.stride(0)))
        self.basis.post_call(self.kpoint.lane)
        return Hpsi

    def __matmul__(self, psi):
        return self.mul_(torch.empty_like(psi), psi)


def mul_(Hpsi, H: DftHamiltonianBlock, psi):
    """Free-function spelling of ``mul!(Hpsi, H, psi)``."""
    return H.mul_(Hpsi, psi)
