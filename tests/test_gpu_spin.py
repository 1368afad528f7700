"""Collinear spin through the boundary (src/Model.jl:29-39, src/PlaneWaveBasis.jl:50-53, src/Kpoint.jl:58-74,
src/densities.jl:39, src/terms/xc.jl:84-175): the device path against the oracle, whose spin path is pinned by the
reference's ABINIT values for bcc iron (tests/test_oracle_golden.py)."""
import ctypes as C

import numpy as np
import pytest

from conftest import free_port  # noqa: E402

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import dftk_jl_amd as dftk  # noqa: E402
import oracle  # noqa: E402
from dftk_jl_amd._lib import check  # noqa: E402
from tests.test_oracle_golden import IRON_LATTICE, IRON_REF_ETOT, _match_reference_spectra, iron_oracle_basis  # noqa: E402


@pytest.fixture(autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    torch.manual_seed(3)


def iron_device_basis():
    Fe = dftk.ElementPsp("Fe", dftk.load_psp("Fe", "lda"))
    model = dftk.model_DFT(IRON_LATTICE, [Fe], [np.zeros(3)], functionals=("lda_xc_teter93",), temperature=0.01,
                           smearing="fermi_dirac", magnetic_moments=(4.0,), symmetries=True)
    return dftk.PlaneWaveBasis(model, 15, dftk.MonkhorstPack((4, 4, 4), (0.5, 0.5, 0.5)), fft_size=(20, 20, 20))


def test_collinear_local_potential_behind_abi_matches_oracle():
    """dftk_mi_local_potential_collinear: (rho_up, rho_down) -> (V_up, V_down) = V_loc + V_H[rho_tot] + v_xc,s and the three
    energies, against the oracle's spin-polarised closed forms (complex-step derivatives there, forward-mode dual numbers
    in the kernel), for lda_xc_teter93 and lda_x + lda_c_pw; the unpolarised lda_xc_teter93 bit of dftk_mi_local_potential."""
    ob = iron_oracle_basis()
    db = iron_device_basis()
    rho = oracle.guess_density(ob, (4.0,))
    drho = dftk.guess_density(db, (4.0,))
    assert drho.shape == (2, 20, 20, 20)
    np.testing.assert_allclose(drho.cpu().numpy(), rho, atol=1e-12)
    T = ob.terms
    vh = ob.irfft_cube(T.poisson * ob.fft_cube(rho.sum(axis=0)))
    for funs, mask in ((("lda_xc_teter93",), 32), (("lda_x", "lda_c_pw"), 5)):
        ob.model.functionals = funs
        exc, vxc = oracle.terms.xc_energy_potential_spin(ob, rho)
        V = torch.empty_like(drho)
        E3 = (C.c_double * 3)()
        torch.cuda.synchronize()
        check(db.lib.dftk_mi_local_potential_collinear(db._cube_handle, drho.data_ptr(), db.terms.V_loc.data_ptr(),
                                                       db.terms.poisson.data_ptr(), mask, V.data_ptr(), E3))
        want = vxc + (T.V_loc + vh)[None]
        assert np.max(np.abs(V.cpu().numpy() - want)) < 1e-10 * max(1.0, np.max(np.abs(want)))
        rho_G = ob.fft_cube(rho.sum(axis=0))
        assert abs(E3[0] - float(np.real(np.vdot(T.poisson * rho_G, rho_G)) / 2)) < 1e-10
        assert abs(E3[1] - exc) < 1e-10 and abs(E3[2] - float(np.sum(rho.sum(axis=0) * T.V_loc) * ob.dvol)) < 1e-10
    ob.model.functionals = ("lda_xc_teter93",)
    # unsupported functional with spin: refused, not silently unpolarised
    assert db.lib.dftk_mi_local_potential_collinear(db._cube_handle, drho.data_ptr(), None, None, 2, None, E3) < 0
    # unpolarised teter93
    rt = torch.from_numpy(rho.sum(axis=0)).cuda().contiguous()
    V1 = torch.empty_like(rt)
    torch.cuda.synchronize()
    check(db.lib.dftk_mi_local_potential(db._cube_handle, rt.data_ptr(), None, None, 32, V1.data_ptr(), E3))
    e1, v1 = oracle.terms._lda_xc_teter93(np.maximum(rho.sum(axis=0), 1e-300))
    assert np.max(np.abs(V1.cpu().numpy() - v1)) < 1e-11 and abs(E3[1] - e1.sum() * ob.dvol) < 1e-11


def test_density_accumulate_spin_indexes_the_cube_of_the_kblock():
    """dftk_mi_density_accumulate_spin: rho[:, :, :, kpt.spin] += ... (densities.jl:39) -- the bands of a spin-down block
    land in the second cube only, and equal what the unindexed entry adds to a single cube."""
    db = iron_device_basis()
    kup, kdn = db.kpoints[0], db.kpoints[6]
    assert (kup.spin, kdn.spin) == (1, 2) and np.allclose(kup.coordinate, kdn.coordinate)
    psi = dftk.random_orbitals(db, kdn, 5)
    w = np.array([1.0, 0.7, 0.3, 0.0, 0.05]) * db.kweights[6] * db.ifft_normalization ** 2
    rho2 = torch.zeros((2, 20, 20, 20), dtype=torch.float64, device="cuda")
    rho1 = torch.zeros((20, 20, 20), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    check(db.lib.dftk_mi_density_accumulate_spin(kdn.handle, 5, psi.data_ptr(), psi.stride(0), w.ctypes.data, rho2.data_ptr(), 1, 2))
    check(db.lib.dftk_mi_density_accumulate(kdn.handle, 5, psi.data_ptr(), psi.stride(0), w.ctypes.data, rho1.data_ptr()))
    db.sync()
    assert float(rho2[0].abs().max()) == 0.0 and torch.equal(rho2[1], rho1) and float(rho1.sum()) > 0
    assert db.lib.dftk_mi_density_accumulate_spin(kdn.handle, 5, psi.data_ptr(), psi.stride(0), w.ctypes.data,
                                                  rho2.data_ptr(), 2, 2) < 0          # spin index out of range


def test_iron_lda_collinear_scf_matches_oracle_and_reference_abinit_values():
    """The reference's spin-polarised SCF test (test/iron_lda.jl) on the device: energies term by term and the spectra of
    both spin channels against the ORACLE (1e-8 Ha per atom, 1e-7), and against the reference's ABINIT pins at its own
    tolerance (5e-6); the magnetisation survives (ferromagnetic solution, ~2.5 mu_B)."""
    db = iron_device_basis()
    assert len(db.kpoints) == 12 and [k.spin for k in db.kpoints] == [1] * 6 + [2] * 6 and abs(sum(db.kweights) - 2) < 1e-14
    res = dftk.self_consistent_field(db, rho=dftk.guess_density(db, (4.0,)), tol=1e-9)
    assert res["converged"]
    ob = iron_oracle_basis()
    ores = oracle.self_consistent_field(ob, rho=oracle.guess_density(ob, (4.0,)), tol=1e-9)
    assert ores["converged"]
    assert abs(res["energies"].total - ores["energies"].total) < 1e-8
    for name, v in ores["energies"].items():
        assert abs(res["energies"][name] - v) < 1e-7, name
    nconv = ores["n_bands_converge"]
    for lam, olam in zip(res["eigenvalues"], ores["eigenvalues"]):
        np.testing.assert_allclose(np.asarray(lam)[:nconv - 2], np.asarray(olam)[:nconv - 2], atol=1e-7)
    assert abs(res["eF"] - ores["eF"]) < 1e-7
    rho, orho = res["rho"].cpu().numpy(), ores["rho"]
    assert rho.shape == (2, 20, 20, 20) and np.linalg.norm(rho - orho) * np.sqrt(ob.dvol) < 1e-7
    mag = float((rho[0] - rho[1]).sum() * ob.dvol)
    assert 2.0 < mag < 3.0
    assert abs(res["energies"].total - IRON_REF_ETOT) < 5e-6
    worst, used = _match_reference_spectra(res["eigenvalues"], n_check=nconv - 3)
    assert worst < 5e-6 and used == set(range(12))
    # the wire format indexes [spin][kpoint][band] (input_output.jl:345-386)
    d = dftk.scfres_to_dict(res)
    assert np.array(d["eigenvalues"]).shape[:2] == (2, 6) and np.array(d["ρ"]).shape == (2, 20, 20, 20)
    assert d["spin_polarization"] == "collinear" and d["n_spin_components"] == 2


def test_collinear_checkpoint_round_trip_and_restart(tmp_path):
    """save_scfres / load_scfres (src/scf/scfres.jl:1-35, :69-86) of a collinear run: the .npz checkpoint holds one orbital
    block per (k-point, spin) -- spin-up blocks, then spin-down ones -- and restarts the SCF at its fixed point."""
    db = iron_device_basis()
    res = dftk.self_consistent_field(db, rho=dftk.guess_density(db, (4.0,)), tol=1e-8)
    assert res["converged"]
    fn = str(tmp_path / "iron.npz")
    dftk.save_scfres(fn, res)
    back = dftk.load_scfres(fn, db)
    assert len(back["psi"]) == len(db.kpoints) == 12 and back["rho"].shape == (2, 20, 20, 20)
    for p, q in zip(back["psi"], res["psi"]):
        assert torch.equal(p, q)
    assert np.array(back["kpt_n_G_vectors"]).shape == (2, 6)
    again = dftk.self_consistent_field(db, rho=back["rho"], psi=back["psi"], tol=1e-8)
    assert again["converged"] and again["n_iter"] <= 3
    assert abs(again["energies"].total - res["energies"].total) < 1e-8
    # an unpolarised basis of the same cell is refused (half the k-blocks)
    Fe = dftk.ElementPsp("Fe", dftk.load_psp("Fe", "lda"))
    m1 = dftk.model_DFT(IRON_LATTICE, [Fe], [np.zeros(3)], functionals=("lda_xc_teter93",), temperature=0.01,
                        smearing="fermi_dirac", symmetries=True)
    b1 = dftk.PlaneWaveBasis(m1, 15, dftk.MonkhorstPack((4, 4, 4), (0.5, 0.5, 0.5)), fft_size=(20, 20, 20))
    with pytest.raises(ValueError):
        dftk.load_scfres(fn, b1)


def test_collinear_model_without_magnetisation_equals_the_unpolarised_model_on_device():
    """n_spin = 2 with zero spin density == the unpolarised model (lda_x + lda_c_pw, the LDA() default of the BASELINE
    configs): same energy, both channels' eigenvalues = the unpolarised ones, rho_up = rho_down = rho / 2."""
    lat, atoms, pos = dftk.silicon_cell()
    kg = dftk.ExplicitKpoints([[0, 0, 0], [0.25, 0.0, -0.5]], [0.5, 0.5])
    m1 = dftk.model_DFT(lat, atoms, pos)
    m2 = dftk.model_DFT(lat, atoms, pos, spin_polarization="collinear")
    b1 = dftk.PlaneWaveBasis(m1, 7, kg, fft_size=(18, 18, 18))
    b2 = dftk.PlaneWaveBasis(m2, 7, kg, fft_size=(18, 18, 18))
    r1 = dftk.self_consistent_field(b1, tol=1e-9)
    r2 = dftk.self_consistent_field(b2, tol=1e-9)
    assert r1["converged"] and r2["converged"] and r2["rho"].shape == (2, 18, 18, 18)
    assert abs(r1["energies"].total - r2["energies"].total) < 1e-9
    for ik in range(2):
        for s_ in range(2):
            np.testing.assert_allclose(r2["eigenvalues"][ik + 2 * s_][:4], r1["eigenvalues"][ik][:4], atol=1e-7)
    assert float(torch.linalg.norm(r2["rho"][0] - r1["rho"] / 2)) * np.sqrt(b1.dvol) < 1e-7


SPIN_WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch, torch.distributed as dist
import dftk_jl_amd as dftk
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["PORT"],
                        rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
comm = dftk.KptComm.from_torch()
lat = 2.71176 * np.array([[-1, 1, 1], [1, -1, 1], [1, 1, -1.0]])
Fe = dftk.ElementPsp("Fe", dftk.load_psp("Fe", "lda"))
model = dftk.model_DFT(lat, [Fe], [np.zeros(3)], functionals=("lda_xc_teter93",), temperature=0.01, smearing="fermi_dirac",
                       magnetic_moments=(4.0,), symmetries=True)
basis = dftk.PlaneWaveBasis(model, 15, dftk.MonkhorstPack((4, 4, 4), (0.5, 0.5, 0.5)), fft_size=(20, 20, 20), device="cuda:0",
                            comm_kpts=comm)
# 6 irreducible k-points over 2 ranks: every rank holds 3 spin-up blocks followed by its 3 spin-down blocks
assert len(basis.kpoints) == 6 and [k.spin for k in basis.kpoints] == [1, 1, 1, 2, 2, 2]
res = dftk.self_consistent_field(basis, rho=dftk.guess_density(basis, (4.0,)), tol=1e-9)
d = dftk.scfres_to_dict(res)
if comm.rank == 0:
    rho = res["rho"]
    print("RESULT " + json.dumps({"E": res["energies"].total, "terms": dict(res["energies"]), "eF": res["eF"],
                                  "eig": d["eigenvalues"], "converged": bool(res["converged"]),
                                  "mag": float((rho[0] - rho[1]).sum()) * basis.dvol}))
dist.barrier(); dist.destroy_process_group()
'''


def test_collinear_spin_with_kpoints_split_over_two_ranks_equals_single_rank(tmp_path):
    """k-point sharding of a collinear model (PlaneWaveBasis.jl:218-232: every rank lists ITS spin-up blocks, then ITS
    spin-down blocks; the density all-reduce carries both spin cubes; the Fermi level is found from the gathered
    eigenvalues of all (k, spin) blocks): two ranks on the one GPU of the box (host-staged transport over gloo) against
    the one-rank run, and the [spin][kpoint][band] wire format assembled from both ranks in global k order."""
    import json
    import os
    import sys
    from tests.test_gpu_multirank import ROOT, _spawn
    script = tmp_path / "spin_worker.py"
    script.write_text(SPIN_WORKER)
    port = free_port()
    base = dict(os.environ, WORLD_SIZE="2", PORT=port, REPO=ROOT, MASTER_ADDR="127.0.0.1")
    outs = _spawn([([sys.executable, str(script)], dict(base, RANK=str(r))) for r in range(2)])
    got = json.loads([ln for ln in outs[0].splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):])
    assert got["converged"]
    db = iron_device_basis()
    ref = dftk.self_consistent_field(db, rho=dftk.guess_density(db, (4.0,)), tol=1e-9)
    assert ref["converged"]
    assert abs(got["E"] - ref["energies"].total) < 1e-8
    for name, v in ref["energies"].items():
        assert abs(got["terms"][name] - v) < 1e-7, name
    assert abs(got["eF"] - ref["eF"]) < 1e-7
    nconv = ref["n_bands_converge"]
    eig = np.array(got["eig"])                                        # [spin][kpoint][band], global k order
    assert eig.shape[:2] == (2, 6)
    for s_ in range(2):
        for ik in range(6):
            np.testing.assert_allclose(eig[s_, ik, :nconv - 2], np.asarray(ref["eigenvalues"][6 * s_ + ik])[:nconv - 2], atol=1e-7)
    rho = ref["rho"]
    assert abs(got["mag"] - float((rho[0] - rho[1]).sum()) * db.dvol) < 1e-6
