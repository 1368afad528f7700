"""``comm_kpts``: the k-point communicator of PlaneWaveBasis (src/PlaneWaveBasis.jl:183-235,
src/common/mpi.jl:19-53) re-designed for one process per GPU.

* ``KptComm.single()``   -- one rank, no communication.
* ``KptComm.from_torch()`` -- uses an initialised ``torch.distributed`` group for the rendezvous
  (rank / size).  The bulk density all-reduce is ``torch.distributed.all_reduce`` -- RCCL over xGMI
  with the ``nccl`` backend on GPUs, gloo on CPU tensors (tests of the sharding logic).  With
  ``DFTK_MI_COMM=abi`` GPU reductions go through the library's own RCCL communicator instead
  (``dftk_mi_allreduce_sum_f64``, the entry point a Julia shim binds; the unique id is shipped
  through the torch store).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch


def split_evenly(n_items: int, n_parts: int):
    """Contiguous ranges as src/common/split_evenly.jl:4-19 (first parts get the remainder)."""
    base, rem = divmod(n_items, n_parts)
    out, start = [], 0
    for p in range(n_parts):
        size = base + (1 if p < rem else 0)
        out.append(range(start, start + size))
        start += size
    return out


class KptComm:
    def __init__(self, rank=0, size=1, group=None):
        self.rank, self.size, self.group = rank, size, group
        self._rccl = None

    @staticmethod
    def single():
        return KptComm()

    @staticmethod
    def from_torch(group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            return KptComm()
        return KptComm(dist.get_rank(group), dist.get_world_size(group), group if group is not None else True)

    # -- RCCL communicator of the C ABI, created lazily on first GPU reduction
    def _ensure_rccl(self, device_index: int):
        if self._rccl is not None or self.size == 1:
            return
        import torch.distributed as dist
        from ._lib import load, check
        lib = load()
        buf = C.create_string_buffer(128)
        if self.rank == 0:
            check(lib.dftk_mi_comm_get_unique_id(buf))
        obj = [bytes(buf.raw)]
        dist.broadcast_object_list(obj, src=0, group=None if self.group is True else self.group)
        handle = C.c_void_p()
        check(lib.dftk_mi_comm_init_rank(obj[0], self.size, self.rank, device_index, C.byref(handle)))
        self._rccl = handle

    def sum_(self, t: torch.Tensor, stream_ptr=None) -> torch.Tensor:
        """mpi_sum!(arr, comm) (common/mpi.jl:19-21), in place."""
        if self.size == 1:
            return t
        if t.is_cuda and os.environ.get("DFTK_MI_COMM", "torch") == "abi":
            # the C-ABI communicator (what a Julia shim uses): dftk_mi_allreduce_sum_f64 -> ncclAllReduce
            if t.dtype != torch.float64 or not t.is_contiguous():
                raise ValueError("RCCL density all-reduce expects a contiguous float64 tensor")
            from ._lib import load, check
            self._ensure_rccl(t.device.index or 0)
            check(load().dftk_mi_allreduce_sum_f64(self._rccl, t.data_ptr(), t.numel(), stream_ptr))
            return t
        import torch.distributed as dist
        dist.all_reduce(t, group=None if self.group is True else self.group)
        return t

    def sum_scalar(self, x: float) -> float:
        if self.size == 1:
            return x
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, group=None if self.group is True else self.group)
        return float(t.item())

    def max_scalar(self, x: float) -> float:
        if self.size == 1:
            return x
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=None if self.group is True else self.group)
        return float(t.item())

    def gather_lists(self, local):
        """All-gather a small picklable object (eigenvalues for the Fermi level)."""
        if self.size == 1:
            return [local]
        import torch.distributed as dist
        out = [None] * self.size
        dist.all_gather_object(out, local, group=None if self.group is True else self.group)
        return out


def distribute_kpoints(kcoords, kweights, comm: KptComm):
    """k-point split of PlaneWaveBasis.jl:183-235, including the duplication (with halved
    weights) of the heaviest k-points when there are more ranks than k-points."""
    kcoords = [np.asarray(k, dtype=float) for k in kcoords]
    kweights = [float(w) for w in kweights]
    n_kpt = len(kcoords)
    if comm.size > n_kpt:
        for _ in range(n_kpt, comm.size):
            idx = int(np.argmax(kweights))
            kweights[idx] *= 0.5
            kweights.append(kweights[idx])
            kcoords.append(kcoords[idx])
    ranges = split_evenly(len(kcoords), comm.size)
    mine = ranges[comm.rank]
    return [kcoords[i] for i in mine], [kweights[i] for i in mine], kcoords, kweights, ranges
