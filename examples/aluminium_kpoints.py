#!/usr/bin/env python
"""A metal on a dense k-mesh (BASELINE configs[2]): fcc aluminium, PBE, Ecut 40, 12x12x12 mesh with symmetries
(72 irreducible k-points), Gaussian smearing, LDOS mixing (the default, as in the reference).

One GPU:    python examples/aluminium_kpoints.py
N GPUs:     python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                examples/aluminium_kpoints.py          (k-points split over the ranks, one density all-reduce per step)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402

comm = dftk.KptComm.single()
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    comm = dftk.KptComm.from_torch()

a = 7.6324708938577865
lattice = a / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
Al = dftk.ElementPsp("Al", dftk.load_psp("Al", "pbe"))
model = dftk.model_DFT(lattice, [Al], [np.zeros(3)], functionals=("gga_x_pbe", "gga_c_pbe"), temperature=1e-3,
                       smearing="gaussian", symmetries=True)
basis = dftk.PlaneWaveBasis(model, 40, dftk.MonkhorstPack((12, 12, 12)), device=f"cuda:{local_rank}", comm_kpts=comm)
cb = dftk.ScfDefaultCallback() if comm.rank == 0 else None
scfres = dftk.self_consistent_field(basis, tol=1e-8, callback=cb)
if comm.rank == 0:
    print(f"E = {scfres['energies'].total:.10f} Ha, eF = {scfres['eF']:.8f} Ha, {len(basis.kcoords_global)} k-points "
          f"({len(basis.kpoints)} on this rank), {scfres['n_iter']} SCF steps")
