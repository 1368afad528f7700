"""Debug driver 4: the failing second SCF step, with per-iteration orthogonality checks."""
import os, sys
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
lat, atoms, pos = dftk.silicon_cell((1, 1, 1))
model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
kw = dict(device="cuda:0")
which = os.environ.get("WHICH", "s")
basis = dftk.PlaneWaveBasis(model, 12, dftk.MonkhorstPack((1, 1, 1)), comm_pw=comm if which == "s" else None, **kw)
st = dftk.ScfStepper(basis, tol=1e-8)
try:
    for i in range(3):
        info = st.step()
        say("step", i + 1, info["energies"].total, info["history_drho"][-1], info["diagonalization"]["n_iter"])
except Exception as e:
    say("FAILED", repr(e))
dist.barrier(); dist.destroy_process_group()
