#!/usr/bin/env python
"""A Gamma-only supercell over several GPUs: the plane waves of the single k-block are sharded as row slabs
(``comm_pw``) -- what the reference cannot do (it can only duplicate the k-point on surplus ranks).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        examples/silicon_supercell_sharded.py [n]        (n x n x n supercell, default 4 = 128 atoms)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
comm = dftk.KptComm.single()
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    comm = dftk.KptComm.from_torch()
lattice, atoms, positions = dftk.silicon_cell((n, n, n))
model = dftk.model_DFT(lattice, atoms, positions)
basis = dftk.PlaneWaveBasis(model, 30, dftk.MonkhorstPack((1, 1, 1)), device=f"cuda:{local_rank}", comm_pw=comm)
kpt = basis.kpoints[0]
if comm.rank == 0:
    print(f"{len(atoms)} atoms, fft {basis.fft_size}, n_G {kpt.n_G}: rows [{kpt.row0}, {kpt.row1}) on rank 0 of {comm.size}")
scfres = dftk.self_consistent_field(basis, tol=1e-6, callback=dftk.ScfDefaultCallback() if comm.rank == 0 else None)
if comm.rank == 0:
    print(f"E = {scfres['energies'].total:.10f} Ha in {scfres['n_iter']} SCF steps, {scfres['runtime']:.1f} s")
