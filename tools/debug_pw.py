"""Debug driver (2 ranks on one GPU over gloo): localise a plane-wave sharding bug step by step."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch, torch.distributed as dist
import dftk_jl_amd as dftk
from dftk_jl_amd._lib import check
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["PORT"],
                        rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
comm = dftk.KptComm.from_torch()
lat, atoms, pos = dftk.silicon_cell((2, 2, 2))
model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
kw = dict(fft_size=(40, 40, 40), device="cuda:0")
basis = dftk.PlaneWaveBasis(model, 8, dftk.MonkhorstPack((1, 1, 1)), comm_pw=comm, **kw)
full = dftk.PlaneWaveBasis(model, 8, dftk.MonkhorstPack((1, 1, 1)), **kw)       # unsharded twin on every rank
kpt, fk = basis.kpoints[0], full.kpoints[0]
rows = slice(kpt.row0, kpt.row1)
def say(*a):
    if comm.rank == 0:
        print(*a, flush=True)
rho0 = dftk.guess_density(full)
_, ham = dftk.energy_hamiltonian(basis, None, None, rho=rho0)
_, fham = dftk.energy_hamiltonian(full, None, None, rho=rho0)
gen = torch.Generator(device="cuda"); gen.manual_seed(11)
M = 35
psi_full = dftk.random_orbitals(full, fk, M, gen)
psi = psi_full[:, rows].contiguous()
for which in (2, 1, 3, 4, 7):
    got = ham[0].mul_(torch.empty_like(psi), psi, which)
    ref = fham[0].mul_(torch.empty_like(psi_full), psi_full, which)[:, rows]
    say("H parts", which, "relerr", float((got - ref).norm() / ref.norm()))
occ = [np.concatenate([np.full(20, 2.0), np.full(5, 0.7), np.zeros(M - 25)])]
r1 = dftk.compute_density(basis, [psi], occ); r2 = dftk.compute_density(full, [psi_full], occ)
say("density relerr", float((r1 - r2).norm() / r2.norm()))
lib = basis.lib
def hist_of(H):
    h, ns = dftk.eigen.lobpcg_residual_history(H)
    return h
for maxit in (0, 1, 2, 3, 6):
    try:
        rs = dftk.lobpcg_hyper(ham[0], psi, prec=dftk.PreconditionerTPA(ham[0]), tol=1e-9, maxiter=maxit, n_conv_check=28)
        rf = dftk.lobpcg_hyper(fham[0], psi_full, prec=dftk.PreconditionerTPA(fham[0]), tol=1e-9, maxiter=maxit, n_conv_check=28)
    except Exception as e:
        say("maxiter", maxit, "FAILED", repr(e)); break
    parts = comm.gather_lists((kpt.row0, rs.X.cpu().numpy()))
    Xg = np.concatenate([p[1] for p in sorted(parts, key=lambda t: t[0])], axis=1)
    G = Xg.conj() @ Xg.T
    hs, hf = hist_of(ham[0]), hist_of(fham[0])
    say("maxiter", maxit, "|X'X-I|", float(np.abs(G - np.eye(M)).max()), "lam diff", float(np.abs(rs.λ - rf.λ).max()),
        "hist rel diff", float((np.abs(hs - hf) / np.maximum(hf, 1e-300)).max()),
        "X vs full", float(np.abs(np.abs(Xg) - np.abs(rf.X.cpu().numpy())).max()))
dist.barrier(); dist.destroy_process_group()
