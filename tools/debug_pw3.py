"""Debug driver 3: first-call vs second-call behaviour, with DFTK_MI_POISON=1 (NaN-filled scratch)."""
import os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch, torch.distributed as dist
import dftk_jl_amd as dftk
torch.cuda.set_device(0)
world = int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["PORT"],
                            rank=int(os.environ["RANK"]), world_size=world)
    comm = dftk.KptComm.from_torch()
else:
    comm = dftk.KptComm.single()
def say(*a):
    if comm.rank == 0:
        print(*a, flush=True)
lat, atoms, pos = dftk.silicon_cell((2, 2, 2))
model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
kw = dict(fft_size=(40, 40, 40), device="cuda:0")
order = os.environ.get("ORDER", "sf")
bases = {}
for tag in order:
    bases[tag] = dftk.PlaneWaveBasis(model, 8, dftk.MonkhorstPack((1, 1, 1)), comm_pw=comm if tag == "s" else None, **kw)
full_for_rho = bases[order[0]]
M = 35
hists = {}
for tag in order:
    basis = bases[tag]
    kpt = basis.kpoints[0]
    rho0 = dftk.guess_density(basis)
    _, ham = dftk.energy_hamiltonian(basis, None, None, rho=rho0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(11)
    psi = dftk.random_orbitals(basis, kpt, M, gen)
    for call in range(3):
        try:
            r = dftk.lobpcg_hyper(ham[0], psi, prec=dftk.PreconditionerTPA(ham[0]), tol=1e-6, n_conv_check=32)
            h, _ = dftk.eigen.lobpcg_residual_history(ham[0])
            hists[(tag, call)] = h
            say(tag, "call", call, "n_iter", r.n_iter, "resid it2 head", np.array2string(h[:4, 2], precision=6))
        except Exception as e:
            say(tag, "call", call, "FAILED", repr(e))
keys = sorted(hists)
ref = hists[keys[-1]]
for k in keys:
    h = hists[k]; n = min(h.shape[1], ref.shape[1], 8)
    say(k, "dev vs last", np.array2string((np.abs(h[:, :n] - ref[:, :n]) / np.maximum(ref[:, :n], 1e-300)).max(axis=0), precision=1))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
