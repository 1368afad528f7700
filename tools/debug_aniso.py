import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk, oracle
lat = np.array([[4.66, -2.33, 0.0], [0.0, 4.0357, 0.0], [0.0, 0.0, 18.0]])
C_ = dftk.ElementPsp("C", dftk.load_psp("C", "lda"))
pos = [np.array([0.0, 0.0, 0.0]), np.array([1 / 3, 2 / 3, 0.0])]
model = dftk.model_DFT(lat, [C_, C_], pos, functionals=("lda_x", "lda_c_vwn"))
kg = dftk.ExplicitKpoints([[1 / 3, 1 / 3, 0.0], [0.0, 0.0, 0.0]], [0.5, 0.5])
basis = dftk.PlaneWaveBasis(model, 12, kg)
oC = oracle.ElementPsp("C", oracle.load_psp_hgh("C", "lda"))
ob = oracle.PlaneWaveBasis(oracle.model_DFT(lat, [oC, oC], pos, functionals=("lda_x", "lda_c_vwn")), 12, oracle.ExplicitKpoints(kg.kcoords, kg.kweights))
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
print("fft", basis.fft_size, "V_loc", rel(basis.terms.V_loc.cpu().numpy(), ob.terms.V_loc))
print("poisson", rel(basis.terms.poisson.cpu().numpy(), ob.terms.poisson))
rho0 = dftk.guess_density(basis); orho0 = oracle.guess_density(ob)
E, ham = dftk.energy_hamiltonian(basis, None, None, rho=rho0)
oE, oham = oracle.energy_hamiltonian(ob, None, None, rho=orho0)
print({k: E[k] - oE[k] for k in oE})
print("V", rel(ham[0].potential.cpu().numpy(), oham[0].potential))
rng = np.random.default_rng(11)
for ik, (H, oH) in enumerate(zip(ham, oham)):
    print("kin", rel(basis.terms.kinetic[ik].cpu().numpy(), oH.kinetic), "P", rel(basis.terms.P[ik].cpu().numpy().T, oH.P), "map", np.array_equal(H.kpoint.mapping, oH.kpoint.mapping))
    psi = np.linalg.qr(rng.standard_normal((oH.n_G, 9)) + 1j * rng.standard_normal((oH.n_G, 9)))[0]
    pd = torch.from_numpy(psi.T.copy()).cuda()
    for which, ref in ((1, oH.apply_local(psi)), (2, oH.kinetic[:, None] * psi), (4, oH.apply_nonlocal(psi))):
        got = H.mul_(torch.empty_like(pd), pd, which).cpu().numpy().T
        print(ik, which, rel(got, ref))
from dftk_jl_amd.terms import xc_energy_potential
from oracle.terms import xc_energy_potential as oxc
e, v = xc_energy_potential(basis, rho0)
oe, ov = oxc(ob, orho0)
d = np.abs(v.cpu().numpy() - ov); i = np.unravel_index(d.argmax(), d.shape)
print("vxc maxdiff", d.max(), "at", i, "rho", orho0[i], rho0.cpu().numpy()[i], "v", ov[i], v.cpu().numpy()[i])
rho_G = basis.fft(rho0); vh = basis.irfft(basis.terms.poisson * rho_G).cpu().numpy()
ovh = ob.irfft_cube(ob.terms.poisson * ob.fft_cube(orho0))
d = np.abs(vh - ovh); i = np.unravel_index(d.argmax(), d.shape)
print("vh maxdiff", d.max(), "at", i, ovh[i], vh[i], "rhoG rel", rel(rho_G.cpu().numpy(), ob.fft_cube(orho0)))
