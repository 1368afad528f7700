"""Lock-step batching of many small k-blocks (``dftk_mi_lobpcg_multi``, dftk.jl_amd/csrc/batch.{h,cpp}): the k loop of
``diagonalize_all_kblocks`` (src/eigen/diag.jl:24-48) as ONE library call.  Every k-block runs the same LOBPCG driver
as ``dftk_mi_lobpcg`` (as a fiber whose device operations are merged with its siblings'), so the results must agree
with the one-by-one calls to round-off, with the oracle, and the SCF on top of it with the lane-pool path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import dftk_jl_amd as dftk  # noqa: E402
import oracle  # noqa: E402
from dftk_jl_amd.eigen import batch_stats, lobpcg_hyper_multi  # noqa: E402

A_AL = 7.6324708938577865


@pytest.fixture(autouse=True)
def _gpu(monkeypatch):
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    torch.manual_seed(3)
    # (the batched k loop is the default for these small blocks; forced here so that the tests do not depend on the rule)
    monkeypatch.setenv("DFTK_MI_KBATCH", "1")


def _si_basis(kgrid=(3, 3, 3), Ecut=12, **kw):
    lat, atoms, pos = dftk.silicon_cell()
    model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
    return dftk.PlaneWaveBasis(model, Ecut, dftk.MonkhorstPack(kgrid), gamma_real=False, **kw)


@pytest.mark.parametrize("sequential", [True, False])
def test_lobpcg_multi_equals_one_by_one_calls(sequential, monkeypatch):
    """27 k-blocks (n_G ~ 500, 7 bands) from the same random start vectors: ``dftk_mi_lobpcg_multi`` against 27 calls of
    ``dftk_mi_lobpcg``.  ``DFTK_MI_KBATCH_SEQUENTIAL=1`` executes every recorded operation through its original entry
    point (only the fibers + recording are exercised: results identical to the last bit); with the merged launches the
    summation orders differ, so eigenvalues agree to 1e-10 and the iteration counts match."""
    if sequential:
        monkeypatch.setenv("DFTK_MI_KBATCH_SEQUENTIAL", "1")
    basis = _si_basis()
    assert basis.kbatch and basis.n_lanes == 1 and len(basis.kpoints) == 27
    _, ham = dftk.energy_hamiltonian(basis, None, None, rho=dftk.guess_density(basis))
    gen = torch.Generator(device="cuda").manual_seed(5)
    X0 = [dftk.random_orbitals(basis, H.kpoint, 7, gen) for H in ham]
    kw = dict(tol=1e-7, n_conv_check=4, maxiter=60)
    one = [dftk.lobpcg_hyper(H, x, prec=dftk.PreconditionerTPA(H), seed=11 + i, **kw) for i, (H, x) in enumerate(zip(ham, X0))]
    many = lobpcg_hyper_multi(ham, X0, prec=True, seeds=[11 + i for i in range(len(ham))], **kw)
    st = batch_stats(basis)
    assert st["rounds"] > 3 and st["ops"] > 100 * len(ham)
    if sequential:
        assert st["merged_launches"] == 0 and st["sequential_ops"] == st["ops"]
    for a, b in zip(one, many):
        assert a.converged and b.converged
        if sequential:
            assert np.array_equal(a.λ, b.λ) and a.n_iter == b.n_iter and a.n_matvec == b.n_matvec
            assert torch.equal(a.X, b.X)
        else:
            np.testing.assert_allclose(b.λ[:4], a.λ[:4], atol=1e-10)
            assert abs(a.n_iter - b.n_iter) <= 1
            # same invariant subspace of the converged bands: | X_a' X_b | is unitary on it
            S = (a.X[:4].conj() @ b.X[:4].T).cpu().numpy()
            assert np.allclose(np.linalg.svd(S, compute_uv=False), 1.0, atol=1e-5)
    if not sequential:
        assert st["merged_launches"] > 0 and st["sequential_ops"] < 0.2 * st["ops"]
    # the first eigenpairs against the oracle's dense diagonalisation of one k-block
    H = ham[5]
    olat, oatoms, opos = oracle.basis.silicon_primitive(a=10.26, functional="lda")
    ob = oracle.PlaneWaveBasis(oracle.model_DFT(olat, oatoms, opos), 12, oracle.MonkhorstPack((3, 3, 3)))
    _, oham = oracle.energy_hamiltonian(ob, None, None, rho=oracle.guess_density(ob))
    ik = [i for i, k in enumerate(ob.kpoints) if np.allclose(k.coordinate, H.kpoint.coordinate)][0]
    dense = np.linalg.eigvalsh(oham[ik].to_dense())[:4]
    np.testing.assert_allclose(many[5].λ[:4], dense, atol=1e-8)


def test_scf_with_kbatch_equals_lane_pool(monkeypatch):
    """``self_consistent_field`` on a k-point mesh with the batched k loop and with the lane pool
    (``DFTK_MI_KBATCH=0``): same energy, eigenvalues and density; Al (metal, PBE, smearing) so that the Gamma point
    (real-symmetric iteration, its own call) and LDOS mixing take part."""
    lat = A_AL / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
    Al = dftk.ElementPsp("Al", dftk.load_psp("Al", "pbe"))
    model = dftk.model_DFT(lat, [Al], [np.zeros(3)], functionals=("gga_x_pbe", "gga_c_pbe"), temperature=1e-3,
                           smearing="gaussian", symmetries=True)
    b1 = dftk.PlaneWaveBasis(model, 20, dftk.MonkhorstPack((6, 6, 6)))
    assert b1.kbatch and b1.n_lanes == 1 and len(b1.kpoints) > 10
    r1 = dftk.self_consistent_field(b1, tol=1e-9)
    monkeypatch.setenv("DFTK_MI_KBATCH", "0")
    b0 = dftk.PlaneWaveBasis(model, 20, dftk.MonkhorstPack((6, 6, 6)))
    assert not b0.kbatch and b0.n_lanes > 1
    r0 = dftk.self_consistent_field(b0, tol=1e-9)
    assert r0["converged"] and r1["converged"]
    assert abs(r0["energies"].total - r1["energies"].total) < 1e-8
    assert abs(r0["eF"] - r1["eF"]) < 1e-7
    for l0, l1 in zip(r0["eigenvalues"], r1["eigenvalues"]):
        np.testing.assert_allclose(l1[:3], l0[:3], atol=1e-7)
    assert float((r0["rho"] - r1["rho"]).norm()) * np.sqrt(b0.dvol) < 1e-7


def test_density_and_apply_H_multi_equal_per_kblock_calls(monkeypatch):
    """The multi-k pipelines (one launch of each FFT stage over the bands of ALL k-blocks, job table per band):
    ``compute_density`` through ``dftk_mi_density_accumulate_multi`` against the per-k-block accumulation, with an
    occupation pattern that drops bands, and H psi inside the batched LOBPCG against ``mul_`` on the returned vectors
    (residual norms recomputed with the one-by-one operator agree with those the batched run reported)."""
    basis = _si_basis(kgrid=(3, 2, 2), Ecut=14)
    assert basis.kbatch
    _, ham = dftk.energy_hamiltonian(basis, None, None, rho=dftk.guess_density(basis))
    gen = torch.Generator(device="cuda").manual_seed(9)
    X0 = [dftk.random_orbitals(basis, H.kpoint, 6, gen) for H in ham]
    res = lobpcg_hyper_multi(ham, X0, prec=True, tol=1e-6, n_conv_check=4, maxiter=60)
    st = batch_stats(basis)
    assert st["sequential_ops"] == 0, st          # every recorded operation had a batched form
    for H, r in zip(ham, res):
        HX = H @ r.X
        lam = (r.X.conj() * HX).sum(dim=1).real
        rn = torch.linalg.norm(HX - lam[:, None] * r.X, dim=1).cpu().numpy()
        # (columns locked before the last iteration report 0: resid_history is only written for active columns,
        #  lobpcg_hyper_impl.jl:443-446 -- their true residuals are below the tolerance)
        rep = r.residual_norms[:4]
        np.testing.assert_allclose(rn[:4][rep > 0], rep[rep > 0], atol=1e-9)
        assert np.all(rn[:4] < 1e-6)
        np.testing.assert_allclose(lam.cpu().numpy()[:4], r.λ[:4], atol=1e-10)
    occ = [np.array([2.0, 2.0, 0.0, 1.5, 1e-9, 0.0]) for _ in ham]
    psi = [r.X for r in res]
    # band-wise kinetic energies of all k-blocks in one call (dftk_mi_band_kinetic_multi) == per-block reductions
    import ctypes as C
    from dftk_jl_amd._lib import check
    n = len(psi)
    out = np.zeros(6 * n)
    check(basis.lib.dftk_mi_band_kinetic_multi(n, (C.c_void_p * n)(*[k.handle.value for k in basis.kpoints]),
                                               (C.c_int * n)(*[6] * n), (C.c_void_p * n)(*[p.data_ptr() for p in psi]),
                                               (C.c_int64 * n)(*[p.stride(0) for p in psi]), out.ctypes.data))
    for ik, (kpt, p) in enumerate(zip(basis.kpoints, psi)):
        ref = ((p.real ** 2 + p.imag ** 2) * kpt.kinetic[None, :]).sum(dim=1).cpu().numpy()
        np.testing.assert_allclose(out[6 * ik:6 * ik + 6], ref, rtol=1e-13)
    rho1 = dftk.compute_density(basis, psi, occ, occupation_threshold=1e-6)
    monkeypatch.setenv("DFTK_MI_KBATCH", "0")
    b0 = _si_basis(kgrid=(3, 2, 2), Ecut=14)
    assert not b0.kbatch
    rho0 = dftk.compute_density(b0, psi, occ, occupation_threshold=1e-6)
    assert float((rho1 - rho0).norm()) < 1e-13 * float(rho0.norm())
    assert abs(float(rho1.sum()) * basis.dvol - 5.5) < 1e-10


def test_kbatch_default_threshold(monkeypatch):
    """Default choice between the two k loops: batched from 32 local k-points on whatever the block size (BASELINE
    configs[2]: 72), and from TWO k-points on when the blocks are small enough for the library's small-block LOBPCG driver
    (M <= 8, n_G * M <= 65536: configs[0] 8 k-points, configs[3] 12); stream lanes for a few LARGE k-blocks."""
    monkeypatch.delenv("DFTK_MI_KBATCH")
    few = _si_basis(kgrid=(3, 3, 3))
    assert len(few.kpoints) == 27 and few.kbatch and few.n_lanes == 1          # 7 bands of ~200 plane waves: small blocks
    many = _si_basis(kgrid=(4, 4, 4), Ecut=8)
    assert len(many.kpoints) == 64 and many.kbatch and many.n_lanes == 1
    lat, atoms, pos = dftk.silicon_cell((2, 2, 2))                              # 16 atoms, 32 + 3 bands: not small blocks
    big = dftk.PlaneWaveBasis(dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw")), 8,
                              dftk.MonkhorstPack((2, 2, 2)), fft_size=(32, 32, 32))
    assert len(big.kpoints) > 1 and not big.kbatch and big.n_lanes > 1


def test_gamma_point_of_a_batched_mesh_stays_in_the_batch():
    """Automatic choice of the real-symmetric Gamma iteration: taken for a Gamma point of the stream-lane path, NOT for
    the Gamma point of a lock-step batched mesh (it would leave the batch for a launch-latency-bound call of its own --
    8 ms of a 26 ms SCF step for 1 of 72 k-points on the Al workload); an explicit request is honoured either way."""
    lat, atoms, pos = dftk.silicon_cell()
    model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
    auto = dftk.PlaneWaveBasis(model, 10, dftk.MonkhorstPack((3, 3, 3)))
    assert auto.kbatch and not any(kpt.gamma_real for kpt in auto.kpoints)
    ig = [i for i, kpt in enumerate(auto.kpoints) if not np.asarray(kpt.coordinate).any()]
    assert len(ig) == 1                                                # the mesh does contain Gamma
    forced = dftk.PlaneWaveBasis(model, 10, dftk.MonkhorstPack((3, 3, 3)), gamma_real=True)
    assert forced.kbatch and forced.kpoints[ig[0]].gamma_real
    lanes = dftk.PlaneWaveBasis(model, 10, dftk.MonkhorstPack((3, 3, 3)), n_lanes=2)
    assert not lanes.kbatch and lanes.kpoints[ig[0]].gamma_real
    # both choices give the same SCF energy (the real-symmetric iteration is an exact restatement at Gamma)
    e = []
    for basis in (auto, forced):
        res = dftk.self_consistent_field(basis, tol=1e-8, maxiter=40, seed=1)
        assert res["converged"]
        e.append(res["energies"].total)
    assert abs(e[0] - e[1]) < 1e-8
