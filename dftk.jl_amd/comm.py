"""Communicators of the hot path, one process per GPU.

``comm_kpts`` -- the k-point communicator of ``PlaneWaveBasis`` (src/PlaneWaveBasis.jl:183-235,
src/common/mpi.jl:19-53): k-points shard over the ranks, one density all-reduce per SCF step.
``comm_pw``   -- NEW (the reference can only duplicate a k-point on surplus ranks,
PlaneWaveBasis.jl:190-203): the plane waves of ONE k-block shard over the ranks as row slabs
(``dftk_mi_kblock_set_shard``), so that a Gamma-only supercell strong-scales over the GPUs of a node.

Both are ``KptComm`` objects.  ``torch.distributed`` provides the rendezvous (rank / size / store);
the DATA path on GPUs is the library's own communicator (``dftk_mi_comm``):

* ``nccl`` process group  -> RCCL over xGMI, bound by the library itself (``dftk_mi_comm_init_rank``; the
  unique id travels through the torch store).  This is the default on GPUs.
* any other process group (gloo) -> the library's host-staged communicator (``dftk_mi_comm_create_host``)
  with callbacks that run ``torch.distributed`` collectives on the pinned host buffers: the path for
  MPI-like hosts, and how the test-suite runs two ranks on ONE GPU.

Scalars (energies, counts, eigenvalue lists) are gathered ONCE per use through the host group
(``gather_lists``) instead of one all-reduce per scalar.
"""
from __future__ import annotations

import ctypes as C

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
        self.host_group = None    # gloo twin of an nccl group: host-side scalar / object collectives
        self._abi = None          # dftk_mi_comm handle (created lazily)
        self._abi_kind = None
        self._callbacks = None    # keeps the ctypes callbacks alive

    @staticmethod
    def single():
        return KptComm()

    @staticmethod
    def from_torch(group=None, host_group=None):
        """Communicator over a ``torch.distributed`` group (default: the world).  For an nccl group a gloo twin carries
        the small host-side collectives; ``dist.new_group`` is COLLECTIVE OVER THE WHOLE WORLD, so for a sub-group
        either every world rank calls ``from_torch(group=...)`` (non-members pass the group too and get a single-rank
        communicator back) or the caller creates the gloo twin itself, collectively, and passes it as ``host_group``."""
        import torch.distributed as dist
        if not dist.is_initialized():
            return KptComm()
        member = group is None or dist.get_rank(group) >= 0
        need_twin = dist.get_backend(group if member else None) == "nccl" and host_group is None
        twin = None
        if need_twin and dist.get_world_size(group if member else None) > 1:
            # small host-side collectives (eigenvalue / energy gathers) stay off the GPU and off RCCL
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            twin = dist.new_group(ranks=ranks, backend="gloo")            # entered by every world rank
        if not member:
            return KptComm()
        comm = KptComm(dist.get_rank(group), dist.get_world_size(group), group if group is not None else True)
        if comm.size > 1:
            comm.host_group = host_group if host_group is not None else twin
        return comm

    def describe(self, device_index=None):
        """What the data path of this communicator is, as the library itself reports it
        (``dftk_mi_comm_describe``): backend, the rank count RCCL saw (``ncclCommCount``), the RCCL version."""
        if self.size == 1:
            return {"backend": "none", "n_ranks": 1}
        import torch
        from ._lib import check, load
        if device_index is None:
            device_index = torch.cuda.current_device() if torch.cuda.is_available() else 0
        h = self.abi_handle(device_index)
        be, n, ver = C.c_int(), C.c_int(), C.c_int()
        check(load().dftk_mi_comm_describe(h, C.byref(be), C.byref(n), C.byref(ver)))
        v = ver.value
        return {"backend": "rccl" if be.value == 0 else "host-staged callbacks", "n_ranks": n.value,
                "version": f"{v // 10000}.{(v // 100) % 100}.{v % 100}" if v else None, "version_code": v}

    def _group(self):
        return None if self.group is True else self.group

    # -- the C-ABI communicator --------------------------------------------------------------------
    def abi_handle(self, device_index: int):
        """``dftk_mi_comm*`` for this group: RCCL when the process group is nccl, host-staged otherwise."""
        if self.size == 1:
            return None
        if self._abi is not None:
            return self._abi
        import torch.distributed as dist
        from ._lib import ALLREDUCE_FN, ALLTOALLV_FN, check, load
        lib = load()
        handle = C.c_void_p()
        if dist.get_backend(self._group()) == "nccl":
            buf = C.create_string_buffer(128)
            if self.rank == 0:
                check(lib.dftk_mi_comm_get_unique_id(buf))
            obj = [bytes(buf.raw)]
            dist.broadcast_object_list(obj, src=dist.get_global_rank(self._group(), 0) if self._group() else 0,
                                       group=self.host_group if self.host_group is not None else self._group())
            check(lib.dftk_mi_comm_init_rank(obj[0], self.size, self.rank, device_index, C.byref(handle)))
            self._abi_kind = "rccl"
        else:
            grp, size = self._group(), self.size

            def allreduce(_user, buf, n):
                try:
                    t = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(n,)))
                    dist.all_reduce(t, group=grp)
                    return 0
                except Exception:      # never unwind through the C frames
                    import traceback
                    traceback.print_exc()
                    return 1

            def alltoallv(_user, send, scnt, soff, recv, rcnt, roff):
                try:
                    sc = [scnt[i] for i in range(size)]
                    so = [soff[i] for i in range(size)]
                    rc = [rcnt[i] for i in range(size)]
                    ro = [roff[i] for i in range(size)]
                    stot = max((o + c for o, c in zip(so, sc)), default=0)
                    rtot = max((o + c for o, c in zip(ro, rc)), default=0)
                    s = torch.from_numpy(np.ctypeslib.as_array(send, shape=(max(stot, 1),)))
                    r = torch.from_numpy(np.ctypeslib.as_array(recv, shape=(max(rtot, 1),)))
                    # gloo has no all_to_all on every build: one broadcast-free exchange with all_gather of
                    # the padded pieces would waste traffic, so use point-to-point pairs ordered by rank
                    me = dist.get_rank(grp)
                    ranks = [dist.get_global_rank(grp, i) if grp else i for i in range(size)]
                    r[ro[me]:ro[me] + rc[me]] = s[so[me]:so[me] + sc[me]]
                    reqs = []
                    for i in range(size):
                        if i == me:
                            continue
                        if sc[i]:
                            reqs.append(dist.isend(s[so[i]:so[i] + sc[i]], dst=ranks[i], group=grp))
                        if rc[i]:
                            reqs.append(dist.irecv(r[ro[i]:ro[i] + rc[i]], src=ranks[i], group=grp))
                    for q in reqs:
                        q.wait()
                    return 0
                except Exception:
                    import traceback
                    traceback.print_exc()
                    return 1

            self._callbacks = (ALLREDUCE_FN(allreduce), ALLTOALLV_FN(alltoallv))
            check(lib.dftk_mi_comm_create_host(self.size, self.rank, device_index,
                                               C.cast(self._callbacks[0], C.c_void_p),
                                               C.cast(self._callbacks[1], C.c_void_p), None, C.byref(handle)))
            self._abi_kind = "host"
        self._abi = handle
        return handle

    # -- reductions --------------------------------------------------------------------------------
    def sum_(self, t: torch.Tensor, stream_ptr=None) -> torch.Tensor:
        """mpi_sum!(arr, comm) (common/mpi.jl:19-21), in place.  CUDA float64 tensors go through the C-ABI
        communicator on ``stream_ptr`` (the basis' stream); anything else through torch.distributed."""
        if self.size == 1:
            return t
        if t.is_cuda:
            if t.dtype != torch.float64 or not t.is_contiguous():
                raise ValueError("density all-reduce expects a contiguous float64 tensor")
            from ._lib import check, load
            handle = self.abi_handle(t.device.index or 0)
            check(load().dftk_mi_allreduce_sum_f64(handle, t.data_ptr(), t.numel(), stream_ptr))
            return t
        import torch.distributed as dist
        dist.all_reduce(t, group=self._group())
        return t

    def gather_lists(self, local):
        """All-gather a small picklable object (eigenvalues, partial energies): ONE collective."""
        if self.size == 1:
            return [local]
        import torch.distributed as dist
        out = [None] * self.size
        dist.all_gather_object(out, local, group=self.host_group if self.host_group is not None else self._group())
        return out

    def sum_scalars(self, xs):
        """Sum a short list of floats over the ranks with one collective; every rank gets the same floats
        (summed in rank order on the host, so the result is identical everywhere)."""
        if self.size == 1:
            return [float(x) for x in xs]
        parts = self.gather_lists([float(x) for x in xs])
        return [float(sum(p[i] for p in parts)) for i in range(len(xs))]

    def sum_scalar(self, x: float) -> float:
        return self.sum_scalars([x])[0]

    def max_scalar(self, x: float) -> float:
        if self.size == 1:
            return x
        return float(max(self.gather_lists(float(x))))


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
