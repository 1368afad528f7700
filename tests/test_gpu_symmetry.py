"""Symmetries on the device path (dftk.jl_amd/symmetry.py: detection, irreducible k-points, density symmetrisation
after ``compute_density``, src/densities.jl:47, src/symmetry.jl:282-357): the reference's own recipe
(test/bzmesh_symmetry.jl:95-128: symmetrised == unsymmetrised) plus the oracle."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import dftk_jl_amd as dftk  # noqa: E402
import oracle  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
A_SI = 5.131570667152971
LATTICE = np.array([[0, A_SI, A_SI], [A_SI, 0, A_SI], [A_SI, A_SI, 0.0]])
POSITIONS = [np.ones(3) / 8, -np.ones(3) / 8]


def _models(symmetries, functionals=("lda_x", "lda_c_pw")):
    dSi = dftk.ElementPsp("Si", dftk.load_psp("Si", "lda"))
    oSi = oracle.ElementPsp("Si", oracle.load_psp_hgh("Si", "lda"))
    return (dftk.model_DFT(LATTICE, [dSi, dSi], POSITIONS, functionals=functionals, symmetries=symmetries),
            oracle.model_DFT(LATTICE, [oSi, oSi], POSITIONS, functionals=functionals, symmetries=symmetries))


@pytest.fixture(autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    torch.manual_seed(5)


def test_symmetrize_rho_matches_oracle_and_is_a_projection():
    dm, om = _models(True)
    kg = (2, 2, 2)
    db = dftk.PlaneWaveBasis(dm, 7, dftk.MonkhorstPack(kg))
    ob = oracle.PlaneWaveBasis(om, 7, oracle.MonkhorstPack(kg))
    assert db.fft_size == ob.fft_size and len(db.symmetries) == len(ob.symmetries) == 48
    rng = np.random.default_rng(1)
    rho = rng.standard_normal(ob.fft_size[::-1])
    from oracle.symmetry import symmetrize_rho as osym
    for lowpass in (False, True):
        ref = osym(ob, rho, do_lowpass=lowpass)
        got = dftk.symmetrize_rho(db, torch.from_numpy(rho).cuda(), do_lowpass=lowpass)
        assert np.linalg.norm(got.cpu().numpy() - ref) < 1e-12 * np.linalg.norm(ref)
    once = dftk.symmetrize_rho(db, torch.from_numpy(rho).cuda(), do_lowpass=True)
    twice = dftk.symmetrize_rho(db, once, do_lowpass=True)
    assert float((once - twice).norm()) < 1e-12 * float(once.norm())       # idempotent
    g = dftk.guess_density(db)                                             # the guess is already symmetric
    assert float((dftk.symmetrize_rho(db, g, do_lowpass=False) - g).norm()) < 1e-12 * float(g.norm())


@pytest.mark.parametrize("kg,shift", [((2, 2, 2), (0.5, 0, 0)), ((2, 2, 2), (0, 0, 0)), ((3, 2, 3), (0, 0.5, 0.5))])
def test_symmetrised_scf_equals_unsymmetrised_and_oracle(kg, shift):
    """test/bzmesh_symmetry.jl:95-128 on the device: |dE| < 1e-10, |d rho| sqrt(dvol) < 1e-8; and == the oracle's
    symmetrised run on the same irreducible k-points."""
    d0, _ = _models(False)
    d1, o1 = _models(True)
    b0 = dftk.PlaneWaveBasis(d0, 5, dftk.MonkhorstPack(kg, shift))
    b1 = dftk.PlaneWaveBasis(d1, 5, dftk.MonkhorstPack(kg, shift))
    ob = oracle.PlaneWaveBasis(o1, 5, oracle.MonkhorstPack(kg, shift))
    assert len(b1.kpoints) == len(ob.kpoints) < len(b0.kpoints) and b0.fft_size == b1.fft_size
    r0 = dftk.self_consistent_field(b0, tol=1e-10)
    r1 = dftk.self_consistent_field(b1, tol=1e-10)
    ro = oracle.self_consistent_field(ob, tol=1e-10)
    assert r0["converged"] and r1["converged"] and ro["converged"]
    assert abs(r0["energies"].total - r1["energies"].total) < 1e-10 * 10      # 2-atom cell, SCF tol 1e-10
    assert float((r0["rho"] - r1["rho"]).norm()) * np.sqrt(b0.dvol) < 1e-8
    assert abs(r1["energies"].total - ro["energies"].total) < 2e-8
    for lam, olam in zip(r1["eigenvalues"], ro["eigenvalues"]):
        np.testing.assert_allclose(lam[:4], olam[:4], atol=1e-7)


def test_cfg1_with_symmetries_as_the_reference_runs_it():
    """BASELINE configs[0] the way DFTK's constructor builds it: symmetries on -> 8 irreducible k-points of the 4x4x4
    mesh, FFT 30^3 (PlaneWaveBasis.jl:349-361); same converged energy as the oracle's unreduced 64-point run on
    the 30^3 cube (tests/golden/baseline_cfg1_si_ecut15_k4_fft30.json)."""
    with open(os.path.join(GOLDEN, "baseline_cfg1_si_ecut15_k4_fft30.json")) as fh:
        g = json.load(fh)
    lat, atoms, pos = dftk.silicon_cell()
    model = dftk.model_DFT(lat, atoms, pos, functionals=tuple(g["functionals"]), symmetries=True)
    basis = dftk.PlaneWaveBasis(model, 15, dftk.MonkhorstPack((4, 4, 4)))
    assert basis.fft_size == (30, 30, 30) and len(basis.kpoints) == 8
    res = dftk.self_consistent_field(basis, tol=1e-10)
    assert res["converged"]
    assert abs(res["energies"].total - g["E_total"]) < 1e-8 * 2
    for kpt, lam in zip(basis.kpoints, res["eigenvalues"]):
        ik = [i for i, k in enumerate(g["kcoords"]) if np.allclose(k, kpt.coordinate)][0]
        np.testing.assert_allclose(lam[:4], np.array(g["eigenvalues"][ik])[:4], atol=1e-7)


@pytest.mark.parametrize("fft_size", [(20, 20, 20), (18, 20, 24), (15, 15, 15)])
def test_symmetrize_rho_kernel_matches_torch_twin_and_oracle(fft_size, monkeypatch):
    """``dftk_mi_symmetrize_rho`` (ONE kernel over G for all symmetries, src/symmetry.jl:282-357) against the torch
    formulation (one gather per symmetry) and the oracle, on cubes where the low-pass matters (even sizes: the Nyquist
    planes leave the grid under rotations; anisotropic sizes lose most rotations on the r-grid unless the cube is
    given) -- explicit ``fft_size`` keeps all 48 operations, so S G regularly leaves the cube."""
    dm, om = _models(True)
    db = dftk.PlaneWaveBasis(dm, 7, dftk.MonkhorstPack((2, 2, 2)), fft_size=fft_size)
    ob = oracle.PlaneWaveBasis(om, 7, oracle.MonkhorstPack((2, 2, 2)), fft_size=fft_size)
    assert len(db.symmetries) == len(ob.symmetries) == 48
    rng = np.random.default_rng(3)
    rho = rng.standard_normal(fft_size[::-1])
    from oracle.symmetry import symmetrize_rho as osym
    for lowpass in (True, False):
        got = dftk.symmetrize_rho(db, torch.from_numpy(rho).cuda(), do_lowpass=lowpass)
        monkeypatch.setenv("DFTK_MI_TORCH_LOCAL", "1")
        twin = dftk.symmetrize_rho(db, torch.from_numpy(rho).cuda(), do_lowpass=lowpass)
        monkeypatch.delenv("DFTK_MI_TORCH_LOCAL")
        ref = osym(ob, rho, do_lowpass=lowpass)
        assert float((got - twin).norm()) < 1e-13 * float(twin.norm())
        assert np.linalg.norm(got.cpu().numpy() - ref) < 1e-12 * np.linalg.norm(ref)
    # in place (rho_out aliases rho_in at the ABI) and the identity-only shortcut
    import ctypes as C
    from dftk_jl_amd._lib import check
    S_h = np.ascontiguousarray(np.stack([np.asfortranarray(s.S).ravel(order="F") for s in db.symmetries]), dtype=np.int32)
    tau_h = np.ascontiguousarray(np.stack([s.tau for s in db.symmetries]))
    buf = torch.from_numpy(rho).cuda()
    torch.cuda.synchronize()
    check(db.lib.dftk_mi_symmetrize_rho(db._cube_handle, 48, S_h.ctypes.data, tau_h.ctypes.data, 1, buf.data_ptr(),
                                        buf.data_ptr()))
    db.sync()                                                    # (asynchronous on the basis' stream, as the header says)
    assert float((buf - got.new_tensor(osym(ob, rho, do_lowpass=True))).norm()) < 1e-12 * float(buf.norm())
    one = torch.from_numpy(rho).cuda()
    out = torch.empty_like(one)
    check(db.lib.dftk_mi_symmetrize_rho(db._cube_handle, 1, S_h[:1].ctypes.data, tau_h[:1].ctypes.data, 1,
                                        one.data_ptr(), out.data_ptr()))
    db.sync()
    assert db.symmetries[0].isone() and torch.equal(out, one)
    bad = S_h[:1].copy()
    bad[0, 0] = 2                                                # det = 2: not a lattice symmetry
    assert db.lib.dftk_mi_symmetrize_rho(db._cube_handle, 1, bad.ctypes.data, tau_h[:1].ctypes.data, 1,
                                         one.data_ptr(), out.data_ptr()) != 0
