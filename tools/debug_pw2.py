"""Debug driver 2 (2 ranks on one GPU over gloo): long LOBPCG runs with locking, then a whole SCF."""
import os, sys, traceback
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch, torch.distributed as dist
import dftk_jl_amd as dftk
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["PORT"],
                        rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
comm = dftk.KptComm.from_torch()
def say(*a):
    if comm.rank == 0:
        print(*a, flush=True)
for sc, ecut, fft in (((2, 2, 2), 8, (40, 40, 40)), ((1, 1, 1), 12, None)):
    lat, atoms, pos = dftk.silicon_cell(sc)
    model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
    kw = dict(fft_size=fft, device="cuda:0")
    basis = dftk.PlaneWaveBasis(model, ecut, dftk.MonkhorstPack((1, 1, 1)), comm_pw=comm, **kw)
    full = dftk.PlaneWaveBasis(model, ecut, dftk.MonkhorstPack((1, 1, 1)), **kw)
    kpt, fk = basis.kpoints[0], full.kpoints[0]
    rows = slice(kpt.row0, kpt.row1)
    rho0 = dftk.guess_density(full)
    _, ham = dftk.energy_hamiltonian(basis, None, None, rho=rho0)
    _, fham = dftk.energy_hamiltonian(full, None, None, rho=rho0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(11)
    M = model.n_electrons // 2 + 3
    psi_full = dftk.random_orbitals(full, fk, M, gen)
    psi = psi_full[:, rows].contiguous()
    say("== cell", sc, "n_G", kpt.n_G, "n_loc", kpt.n_loc, "M", M)
    for tol, ncc in ((1e-6, M - 3), (2e-10, M - 3), (2e-10, M)):
        try:
            rf = dftk.lobpcg_hyper(fham[0], psi_full, prec=dftk.PreconditionerTPA(fham[0]), tol=tol, n_conv_check=ncc)
            hf, _ = dftk.eigen.lobpcg_residual_history(fham[0])
            say("full   tol", tol, "ncc", ncc, "n_iter", rf.n_iter, "conv", rf.converged)
            rs = dftk.lobpcg_hyper(ham[0], psi, prec=dftk.PreconditionerTPA(ham[0]), tol=tol, n_conv_check=ncc)
            hs, _ = dftk.eigen.lobpcg_residual_history(ham[0])
            n = min(hs.shape[1], hf.shape[1])
            say("sharded tol", tol, "ncc", ncc, "n_iter", rs.n_iter, "conv", rs.converged, "lam diff",
                float(np.abs(rs.λ - rf.λ)[:ncc].max()), "hist dev first 8",
                np.array2string((np.abs(hs[:, :n] - hf[:, :n]) / np.maximum(hf[:, :n], 1e-300)).max(axis=0)[:8], precision=1))
        except Exception as e:
            say("FAILED tol", tol, "ncc", ncc, repr(e))
    try:
        res = dftk.self_consistent_field(basis, tol=1e-8, callback=lambda i: say("  scf", i["n_iter"], i["energies"].total,
                                         i["history_drho"][-1], i["diagonalization"]["n_iter"]))
        ref = dftk.self_consistent_field(full, tol=1e-8)
        say("SCF ok: dE", res["energies"].total - ref["energies"].total, "n_iter", res["n_iter"], ref["n_iter"])
    except Exception as e:
        say("SCF FAILED", repr(e))
dist.barrier(); dist.destroy_process_group()
