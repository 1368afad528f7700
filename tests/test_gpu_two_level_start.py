"""Two-level start of the first diagonalisation (basis.py: companion basis at Ecut / 4; eigen.py: lowpass_to_coarse,
zero_pad_to_fine, diagonalize_all_kblocks) -- an extension over the reference, which starts from random orbitals
(src/eigen/diag.jl:39-48).  The transfer operators against NumPy, the start vectors against the random start: fewer fine LOBPCG
iterations, the SAME converged SCF."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd.eigen import lowpass_to_coarse, zero_pad_to_fine  # noqa: E402


@pytest.fixture(scope="module")
def bases():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    lat, atoms, pos = dftk.silicon_cell((2, 2, 2))
    model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
    # (the automatic rule asks for cubes >= 96; here the companion basis is requested explicitly on a 16-atom cell)
    b2 = dftk.PlaneWaveBasis(model, 16.0, dftk.ExplicitKpoints([[0.0, 0.25, -0.5], [0.25, 0.25, 0.0]], [0.5, 0.5]), coarse_start=True)
    b1 = dftk.PlaneWaveBasis(model, 16.0, dftk.ExplicitKpoints([[0.0, 0.25, -0.5], [0.25, 0.25, 0.0]], [0.5, 0.5]), coarse_start=False)
    return b2, b1


def test_companion_basis_and_transfer_operators(bases):
    b2, b1 = bases
    c = b2.coarse
    assert c is not None and b1.coarse is None and c.coarse is None
    assert abs(c.Ecut - b2.Ecut / 4) < 1e-12 and all(nc < nf for nc, nf in zip(c.fft_size, b2.fft_size))
    assert len(c.kpoints) == len(b2.kpoints)
    for kc, kf in zip(c.kpoints, b2.kpoints):
        assert np.allclose(kc.coordinate, kf.coordinate) and kc.n_G < kf.n_G / 4
    # Fourier truncation of a real cube: against numpy.fft on the host
    nx, ny, nz = b2.fft_size
    cx, cy, cz = c.fft_size
    rng = np.random.default_rng(0)
    f = rng.standard_normal((nz, ny, nx))
    got = lowpass_to_coarse(b2, c, torch.tensor(f, device="cuda")).cpu().numpy()
    F = np.fft.fftn(f) / f.size

    def sel(nc, nf):
        g = np.arange(nc)
        g = np.where(g < (nc + 1) // 2, g, g - nc)
        keep = (2 * np.abs(g) < nc) if nc % 2 == 0 else np.ones(nc, bool)
        return g % nf, keep
    (iz, kz), (iy, ky), (ix, kx) = sel(cz, nz), sel(cy, ny), sel(cx, nx)
    Fc = F[np.ix_(iz, iy, ix)] * (kz[:, None, None] & ky[None, :, None] & kx[None, None, :])
    ref = np.fft.ifftn(Fc) * Fc.size
    assert np.abs(ref.imag).max() < 1e-12 * np.abs(ref.real).max()
    assert np.abs(got - ref.real).max() < 1e-12 * np.abs(ref.real).max()
    # zero padding by integer G: the coarse coefficients sit at their G vectors of the fine sphere, zeros elsewhere
    kc, kf = c.kpoints[1], b2.kpoints[1]
    Xc = torch.randn((3, kc.n_G), dtype=torch.complex128, device="cuda")
    Xf = zero_pad_to_fine(Xc, kc, kf, b2)
    assert Xf.shape == (3, kf.n_G) and abs(float(torch.linalg.norm(Xf)) - float(torch.linalg.norm(Xc))) < 1e-12
    Gf = {tuple(g): i for i, g in enumerate(kf.G_vectors.cpu().numpy().tolist())}
    pos = np.array([Gf[tuple(g)] for g in kc.G_vectors.cpu().numpy().tolist()])
    assert torch.equal(Xf[:, torch.as_tensor(pos, device="cuda")], Xc)


def test_two_level_start_needs_fewer_fine_iterations_and_converges_to_the_same_scf(bases):
    b2, b1 = bases
    res = {}
    for name, basis, flag in (("two-level", b2, True), ("random", b2, False), ("no companion", b1, True)):
        st = dftk.ScfStepper(basis, tol=1e-9, seed=5, coarse_start=flag)
        first = st.step()
        it1 = float(np.mean(first["diagonalization"]["n_iter"]))
        nc = int(first.get("n_matvec_coarse", 0))
        info = first
        for _ in range(40):
            if info["converged"]:
                break
            info = st.step()
        assert info["converged"]
        res[name] = (it1, nc, st.finalize()["energies"].total)
    assert res["two-level"][1] > 0 and res["random"][1] == 0 and res["no companion"][1] == 0
    assert res["two-level"][0] < 0.6 * res["random"][0], res            # fine LOBPCG iterations of the first step
    assert abs(res["two-level"][2] - res["random"][2]) < 1e-8 and abs(res["no companion"][2] - res["random"][2]) < 1e-8


TWO_RANK_WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch, torch.distributed as dist
import dftk_jl_amd as dftk
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["PORT"],
                        rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
comm = dftk.KptComm.from_torch()
lat, atoms, pos = dftk.silicon_cell((2, 2, 2))
model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
basis = dftk.PlaneWaveBasis(model, 16, dftk.MonkhorstPack((1, 1, 1)), device="cuda:0", comm_pw=comm, coarse_start=True)
assert basis.coarse is not None and basis.coarse.comm_pw.size == 2
st = dftk.ScfStepper(basis, tol=1e-9, seed=5)
first = st.step()
it1 = float(np.mean(first["diagonalization"]["n_iter"]))
nc = int(first.get("n_matvec_coarse", 0))
info = first
for _ in range(60):
    if info["converged"]:
        break
    info = st.step()
fin = st.finalize()
if comm.rank == 0:
    print("RESULT " + json.dumps({"E": fin["energies"].total, "converged": bool(info["converged"]), "it1": it1, "nc": nc}))
dist.barrier(); dist.destroy_process_group()
'''


def test_two_level_start_of_a_plane_wave_sharded_block_two_ranks_one_gpu(tmp_path):
    """The companion basis of a plane-wave sharded basis is sharded the same way; the zero-padding sums the slabs of the coarse
    block over the ranks and cuts this rank's slab of the fine sphere.  Two ranks on one GPU (host-staged communicator): the
    first step takes the coarse path with as few fine iterations as the one-rank run, the converged energy is the one-rank one."""
    import json
    import os
    import sys
    from conftest import free_port
    from test_gpu_multirank import _spawn, ROOT
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    script = tmp_path / "two_level_worker.py"
    script.write_text(TWO_RANK_WORKER)
    base = dict(os.environ, WORLD_SIZE="2", PORT=free_port(), REPO=ROOT, MASTER_ADDR="127.0.0.1")
    outs = _spawn([([sys.executable, str(script)], dict(base, RANK=str(r))) for r in range(2)], timeout=900.0)
    got = json.loads([ln for ln in outs[0].splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):])
    assert got["converged"] and got["nc"] > 0
    lat, atoms, pos = dftk.silicon_cell((2, 2, 2))
    model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
    basis = dftk.PlaneWaveBasis(model, 16, dftk.MonkhorstPack((1, 1, 1)), coarse_start=True)
    st = dftk.ScfStepper(basis, tol=1e-9, seed=5)
    first = st.step()
    it1 = float(np.mean(first["diagonalization"]["n_iter"]))
    info = first
    for _ in range(60):
        if info["converged"]:
            break
        info = st.step()
    assert info["converged"]
    assert got["it1"] <= it1 + 2, (got["it1"], it1)
    assert abs(got["E"] - st.finalize()["energies"].total) < 1e-8 * 16
