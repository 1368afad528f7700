"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol of the header, the
host planning logic (1-D plans, sphere pruning tables) drives a NumPy emulation of the kernel
pipeline to the exact FFT answer, the host mirror's set-up arrays equal the oracle's, the hot
path refuses to run without a GPU, and the k-point sharding works under gloo with 2 ranks."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import free_port  # noqa: E402
import torch

import dftk_jl_amd as dftk
from dftk_jl_amd._lib import check

import oracle
from oracle.basis import G_axis

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    return dftk.load_library()


def test_library_exports_every_header_symbol(lib):
    header = open(os.path.join(ROOT, "include", "dftk_mi355x.h")).read()
    names = set(re.findall(r"\b(dftk_mi_[A-Za-z0-9_]+)\s*\(", header))
    assert names, "no prototypes found in the header"
    for name in sorted(names):
        assert hasattr(lib, name), f"{name} declared in include/dftk_mi355x.h but not exported"
    assert names == set(dftk.EXPORTED_SYMBOLS), names ^ set(dftk.EXPORTED_SYMBOLS)
    assert b"gfx950" in lib.dftk_mi_version()


def test_no_gpu_fails_loudly(lib):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    st = lib.dftk_mi_basis_create(8, 8, 8, 1.0, 0, C.byref(h))
    assert st == -100 and b"no CPU fallback" in lib.dftk_mi_last_error()
    lat, atoms, pos = dftk.silicon_cell()
    basis = dftk.PlaneWaveBasis(dftk.model_DFT(lat, atoms, pos), 5, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dftk.energy_hamiltonian(basis, None, None, rho=None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dftk.compute_density(basis, [], [])


def get_plan(lib, n):
    nr = C.c_int()
    rad = (C.c_int * 32)()
    pos = (C.c_int * n)()
    check(lib.dftk_mi_fft_plan_host(n, C.byref(nr), rad, pos))
    return list(rad[:nr.value]), np.array(pos[:n])


def dit(buf, n, rad, sgn):
    """NumPy twin of fft_tile<DIF=false> (dftk.jl_amd/csrc/fft_kernels.hip)."""
    tw = np.exp(sgn * 2j * np.pi * np.arange(n) / n)
    m = 1
    for r in rad:
        out = buf.copy()
        for b in range(n // r):
            jj, g = b % m, b // m
            base = g * r * m + jj
            a = [buf[base + q * m] * tw[q * jj * (n // (r * m))] for q in range(r)]
            for p in range(r):
                out[base + p * m] = sum(a[q] * tw[((p * q) % r) * (n // r)] for q in range(r))
        buf, m = out, m * r
    return buf


def dif(buf, n, rad, sgn):
    """NumPy twin of fft_tile<DIF=true>."""
    tw = np.exp(sgn * 2j * np.pi * np.arange(n) / n)
    m = n
    for r in reversed(rad):
        m //= r
        out = buf.copy()
        for b in range(n // r):
            jj, g = b % m, b // m
            base = g * r * m + jj
            a = [buf[base + q * m] for q in range(r)]
            for p in range(r):
                out[base + p * m] = (sum(a[q] * tw[((p * q) % r) * (n // r)] for q in range(r))
                                     * tw[p * jj * (n // (r * m))])
        buf = out
    return buf


@pytest.mark.parametrize("n", [1, 2, 8, 12, 15, 17, 21, 27, 30, 33, 36, 40, 120, 150, 192])
def test_fft_plan_tables(lib, n):
    rad, pos = get_plan(lib, n)
    assert int(np.prod(rad)) == n or (n == 1 and rad == [])
    assert sorted(pos) == list(range(n))
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    buf = np.zeros(n, complex)
    buf[pos] = x
    np.testing.assert_allclose(dit(buf, n, rad, +1), np.fft.ifft(x) * n, atol=1e-12 * n)
    np.testing.assert_allclose(dif(x.copy(), n, rad, -1)[pos], np.fft.fft(x), atol=1e-12 * n)


def test_fft_plan_rejects_large_primes(lib):
    nr, rad, pos = C.c_int(), (C.c_int * 32)(), (C.c_int * 67)()
    assert lib.dftk_mi_fft_plan_host(67, C.byref(nr), rad, pos) == -1


def test_pruned_pipeline_emulation(lib):
    """Stages A-E of the kernel pipeline, emulated with the library's own tables, reproduce
    FFT[V * iFFT[pad(c)]] restricted to the sphere (Hamiltonian.jl:155-163)."""
    a = 5.13
    lat = a * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
    model = oracle.Model(lat, [], [], terms=("Kinetic",), n_electrons=2)
    fft_size = (12, 15, 10)
    ob = oracle.PlaneWaveBasis(model, 4.0, oracle.ExplicitKpoints([[0.1, -0.2, 0.3]], [1.0]), fft_size=fft_size)
    kpt = ob.kpoints[0]
    nx, ny, nz = fft_size
    n_G = len(kpt.mapping)
    nl, nzp = C.c_int64(), C.c_int()
    m = np.ascontiguousarray(kpt.mapping)
    check(lib.dftk_mi_sphere_tables_host(nx, ny, nz, n_G, m.ctypes.data, C.byref(nl), C.byref(nzp), None, None))
    line_id = np.zeros(nl.value, dtype=np.int64)
    line_start = np.zeros(nl.value + 1, dtype=np.int64)
    check(lib.dftk_mi_sphere_tables_host(nx, ny, nz, n_G, m.ctypes.data, C.byref(nl), C.byref(nzp),
                                         line_id.ctypes.data, line_start.ctypes.data))
    assert line_start[-1] == n_G and np.all(np.diff(line_id) > 0)
    plans = [get_plan(lib, n) for n in fft_size]
    rng = np.random.default_rng(0)
    c = rng.standard_normal(n_G) + 1j * rng.standard_normal(n_G)
    V = rng.standard_normal((nz, ny, nx))
    zvals = sorted(set(line_id // ny))
    assert len(zvals) == nzp.value
    # A: x-lines
    T1 = np.zeros((nl.value, nx), complex)
    for l in range(nl.value):
        buf = np.zeros(nx, complex)
        for j in range(line_start[l], line_start[l + 1]):
            buf[plans[0][1][kpt.mapping[j] - line_id[l] * nx]] = c[j]
        T1[l] = dit(buf, nx, plans[0][0], +1)
    # B: y-lines per z plane
    T2 = np.zeros((len(zvals), ny, nx), complex)
    for zi, iz in enumerate(zvals):
        buf = np.zeros((ny, nx), complex)
        for l in np.nonzero(line_id // ny == iz)[0]:
            buf[plans[1][1][line_id[l] % ny]] = T1[l]
        for x in range(nx):
            T2[zi, :, x] = dit(buf[:, x].copy(), ny, plans[1][0], +1)
    # C: z backward, multiply, z forward
    for y in range(ny):
        for x in range(nx):
            buf = np.zeros(nz, complex)
            for zi, iz in enumerate(zvals):
                buf[plans[2][1][iz]] = T2[zi, y, x]
            col = dit(buf, nz, plans[2][0], +1) * V[:, y, x] / (nx * ny * nz)
            col = dif(col, nz, plans[2][0], -1)
            for zi, iz in enumerate(zvals):
                T2[zi, y, x] = col[plans[2][1][iz]]
    # D: y forward, keep sphere lines ; E: x forward + gather
    out = np.zeros(n_G, complex)
    for zi, iz in enumerate(zvals):
        planes = np.stack([dif(T2[zi, :, x].copy(), ny, plans[1][0], -1) for x in range(nx)], axis=1)
        for l in np.nonzero(line_id // ny == iz)[0]:
            row = dif(planes[plans[1][1][line_id[l] % ny]].copy(), nx, plans[0][0], -1)
            for j in range(line_start[l], line_start[l + 1]):
                out[j] = row[plans[0][1][kpt.mapping[j] - line_id[l] * nx]]
    ref = ob.fft(kpt, ob.ifft(kpt, c, normalize=False) * V / (nx * ny * nz), normalize=False)
    np.testing.assert_allclose(out, ref, atol=1e-12 * np.abs(ref).max())


def test_host_mirror_setup_matches_oracle():
    """Package set-up (torch) vs oracle (NumPy): k-sphere, kinetic, projectors P, coupling D,
    Ewald, psp correction -- parity level P0 of SURVEY.md appendix B."""
    lat, atoms, pos = dftk.silicon_cell((2, 1, 1))
    model = dftk.model_DFT(lat, atoms, pos)
    kg = dftk.ExplicitKpoints([[0, 0, 0], [0.25, -0.125, 0.5]], [0.5, 0.5])
    basis = dftk.PlaneWaveBasis(model, 8.0, kg, device="cpu")
    oSi = oracle.ElementPsp("Si", oracle.load_psp_hgh("Si", "lda"))
    omodel = oracle.model_DFT(lat, [oSi] * len(atoms), pos)
    ob = oracle.PlaneWaveBasis(omodel, 8.0, oracle.ExplicitKpoints(kg.kcoords, kg.kweights))
    assert basis.fft_size == ob.fft_size
    for k, ok in zip(basis.kpoints, ob.kpoints):
        np.testing.assert_array_equal(k.mapping, ok.mapping)
        np.testing.assert_array_equal(k.G_vectors.numpy(), ok.G_vectors)
    for ik in range(2):
        np.testing.assert_allclose(basis.terms.kinetic[ik].numpy(), ob.terms.kinetic[ik], rtol=1e-14)
        np.testing.assert_allclose(basis.terms.P[ik].numpy().T, ob.terms.P[ik], rtol=0, atol=1e-13)
    np.testing.assert_array_equal(basis.terms.D, ob.terms.D)
    assert basis.terms.E_ewald == pytest.approx(ob.terms.E_ewald, abs=1e-11)
    assert basis.terms.E_pspcorr == pytest.approx(ob.terms.E_pspcorr, rel=1e-14)
    # golden values (test/ewald.jl:14-26; test/PspHgh.jl:41-52) through the package's own code
    from dftk_jl_amd.terms import energy_ewald
    from dftk_jl_amd.psp import eval_psp_local_fourier, load_psp
    a = 5.131570667152971
    lat0 = np.array([[0, a, a], [a, 0, a], [a, a, 0.0]])
    assert energy_ewald(lat0, [14, 14], [np.ones(3) / 8, -np.ones(3) / 8]) == pytest.approx(-102.8741963352893, abs=1e-8)
    v = eval_psp_local_fourier(load_psp("Si"), torch.tensor([0.1], dtype=torch.float64)).item()
    assert v == pytest.approx(-400.395448865164 * 4 * np.pi, rel=1e-12)
    assert dftk.compute_fft_size(lat0, 15) == (27, 27, 27) and dftk.compute_fft_size(lat0, 30) == (40, 40, 40)


def test_supercell_sizes_match_survey_table():
    """SURVEY.md section 8 size table: cfg 2 = 150^3 / n_G 135 491, cfg 5 = 192^3."""
    lat, atoms, pos = dftk.silicon_cell((4, 4, 4))
    assert len(atoms) == 128 and dftk.compute_fft_size(lat, 30) == (150, 150, 150)
    lat5, atoms5, _ = dftk.silicon_cell((5, 5, 5))
    assert len(atoms5) == 250 and dftk.compute_fft_size(lat5, 30) == (192, 192, 192)


def test_split_and_duplicate_kpoints():
    """PlaneWaveBasis.jl:183-235: contiguous split; more ranks than k-points duplicates the heaviest."""
    assert [list(r) for r in dftk.split_evenly(7, 3)] == [[0, 1, 2], [3, 4], [5, 6]]
    comm = dftk.KptComm(rank=2, size=3)
    kc, kw, allk, allw, ranges = dftk.distribute_kpoints([[0, 0, 0], [0.5, 0, 0]], [0.25, 0.75], comm)
    assert len(allk) == 3 and allw == [0.25, 0.375, 0.375] and abs(sum(allw) - 1) < 1e-15
    assert list(ranges[2]) == [2] and np.allclose(kc[0], [0.5, 0, 0]) and kw == [0.375]


def test_anderson_and_occupation_match_oracle():
    from oracle.scf import AndersonAcceleration as OA
    rng = np.random.default_rng(0)
    A = rng.standard_normal((40, 40))
    A = A @ A.T / 40 + np.eye(40)
    bvec = rng.standard_normal(40)
    x_t = torch.zeros(40, dtype=torch.float64)
    x_n = np.zeros(40)
    acc_t, acc_n = dftk.AndersonAcceleration(m=5), OA(m=5)
    for _ in range(25):
        x_t = acc_t(x_t, 0.5, torch.from_numpy(bvec - A @ x_t.numpy()))
        x_n = acc_n(x_n, 0.5, bvec - A @ x_n)
        np.testing.assert_allclose(x_t.numpy(), x_n, atol=1e-7)
    assert np.linalg.norm(A @ x_n - bvec) < 1e-4
    # Fermi level / occupations: insulator at T = 0 and Gaussian smearing
    lat, atoms, pos = dftk.silicon_cell()
    for T, sm in ((0.0, None), (0.01, "gaussian"), (0.02, "fermi_dirac")):
        model = dftk.model_DFT(lat, atoms, pos, temperature=T, smearing=sm)
        basis = dftk.PlaneWaveBasis(model, 5, dftk.MonkhorstPack((2, 1, 1)), device="cpu", build_terms=False)
        ev = [np.sort(rng.standard_normal(8)) * 0.3 for _ in basis.kpoints]
        occ, eF = dftk.compute_occupation(basis, ev)
        om = oracle.Model(lat, [oracle.ElementPsp("Si", oracle.load_psp_hgh("Si"))] * 2, pos, temperature=T,
                          smearing=sm or "none")
        ob = oracle.PlaneWaveBasis(om, 5, oracle.MonkhorstPack((2, 1, 1)), build_terms=False)
        oocc, oeF = oracle.compute_occupation(ob, ev)
        assert eF == pytest.approx(oeF, abs=1e-12)
        for o1, o2 in zip(occ, oocc):
            np.testing.assert_allclose(o1, o2, atol=1e-12)
        assert sum(w * o.sum() for w, o in zip(basis.kweights, occ)) == pytest.approx(8.0, abs=1e-6)


GLOO_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch, torch.distributed as dist
import dftk_jl_amd as dftk
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["PORT"],
                        rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
comm = dftk.KptComm.from_torch()
lat, atoms, pos = dftk.silicon_cell()
model = dftk.model_DFT(lat, atoms, pos)
basis = dftk.PlaneWaveBasis(model, 5, dftk.MonkhorstPack((1, 1, 3)), device="cpu", comm_kpts=comm, build_terms=False)
# every k-point is owned by exactly one rank and the weights sum to one over the communicator
assert abs(comm.sum_scalar(sum(basis.kweights)) - 1.0) < 1e-14
n_local = len(basis.kpoints)
assert comm.sum_scalar(n_local) == 3 and n_local == len(basis.krange_thisproc)
# density-style reduction: each rank contributes its k-points' partial sums (the path compute_density takes)
nx, ny, nz = basis.fft_size
part = torch.zeros((nz, ny, nx), dtype=torch.float64)
for w, k in zip(basis.kweights, basis.kpoints):
    part += w * float(k.n_G)
comm.sum_(part)
tot = float(part[0, 0, 0])
all_nG = comm.gather_lists([k.n_G for k in basis.kpoints])
flat = [n for sub in all_nG for n in sub]
assert len(flat) == 3 and abs(tot - sum(flat) / 3) < 1e-12
# Fermi level over sharded eigenvalues equals the serial answer
rng = np.random.default_rng(0)
ev_all = [np.concatenate([np.sort(rng.uniform(-1, -0.5, 4)), np.sort(rng.uniform(0.5, 1, 2))]) for _ in range(3)]
mine = [ev_all[i] for i in basis.krange_thisproc]
occ, eF = dftk.compute_occupation(basis, mine)
serial = dftk.PlaneWaveBasis(model, 5, dftk.MonkhorstPack((1, 1, 3)), device="cpu", build_terms=False)
occ_s, eF_s = dftk.compute_occupation(serial, ev_all)
assert abs(eF - eF_s) < 1e-14
for i, o in zip(basis.krange_thisproc, occ):
    assert np.allclose(o, occ_s[i])
assert comm.max_scalar(float(comm.rank)) == comm.size - 1
dist.barrier(); dist.destroy_process_group()
print("rank", comm.rank, "ok")
'''


def test_kpoint_sharding_gloo_world2(tmp_path):
    """N > 1 path on CPU: 2 ranks over gloo (k-point split, weights, reductions, Fermi level)."""
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", PORT=port, REPO=ROOT, MASTER_ADDR="127.0.0.1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


def test_pbe_pointwise_terms_match_oracle():
    """The device mirror evaluates the PBE energy density with torch and differentiates it with autograd; the
    oracle uses NumPy and complex-step derivatives: e, de/drho, de/dsigma agree over the physical range."""
    from oracle.terms import _gga_terms, _GGA_FUNCTIONALS as OG
    from dftk_jl_amd.terms import _GGA_FUNCTIONALS as DG
    rng = np.random.default_rng(0)
    rho = 10 ** rng.uniform(-12, 0.5, 5000)
    s2 = 10 ** rng.uniform(-6, 4, 5000)                       # reduced gradient squared
    kf = (3 * np.pi ** 2 * rho) ** (1 / 3)
    sig = s2 * 4 * kf * kf * rho * rho
    for name in ("gga_x_pbe", "gga_c_pbe"):
        e, vr, vs = _gga_terms(OG[name], rho, sig)
        r = torch.tensor(rho, requires_grad=True)
        sg = torch.tensor(sig, requires_grad=True)
        et = DG[name](r, sg)
        gr, gs = torch.autograd.grad(et.sum(), (r, sg))
        assert np.abs(et.detach().numpy() - e).max() < 1e-14
        assert np.abs(gr.numpy() - vr).max() < 1e-13
        assert np.abs((gs.numpy() - vs) * sig).max() < 1e-13   # V_sigma enters multiplied by grad rho
    # sigma = 0 limits: LDA exchange exactly, PW92 ("mod" parameters) correlation
    from oracle.terms import _lda_x
    e0, v0, _ = _gga_terms(OG["gga_x_pbe"], rho, np.zeros_like(rho))
    el, vl = _lda_x(rho)
    assert np.abs(e0 - el).max() < 1e-15 and np.abs(v0 - vl).max() < 1e-14


def _gemm_plan(lib, trans, m, n, k, flags=0):
    out = (C.c_int * 12)()
    check(lib.dftk_mi_zgemm_plan_host(trans.encode(), m, n, k, flags, out))
    keys = ("bn", "gmf", "gnf", "nright", "nbottom", "nsI", "kcI", "zmI", "nsB", "kcB", "zmB", "shift")
    return dict(zip(keys, out[:12]))


@pytest.mark.parametrize("trans,m,n,k", [("C", 259, 259, 135491), ("C", 640, 259, 135491), ("C", 777, 777, 135491),
                                          ("C", 5, 5, 135491), ("N", 135491, 259, 259), ("N", 135491, 54, 777),
                                          ("C", 259, 259, 518), ("C", 259, 1, 777), ("N", 777, 1, 259), ("C", 1, 1, 1)])
def test_zgemm_launch_plan_invariants(lib, trans, m, n, k):
    """Host-side planning of the MFMA zgemm (tiling into full 128 x 32 tiles + ragged border, K chunks that
    fill the resident workgroup slots, chunk -> XCD placement): structural invariants for the LOBPCG shapes."""
    for flags in (0, 1):
        if flags and (trans != "C" or m != n):
            continue
        p = _gemm_plan(lib, trans, m, n, k, flags)
        assert p["bn"] == 32 and p["gmf"] == m // 128 and p["gnf"] == n // 32
        # a ragged n >= 32 is covered by a full tile shifted left to end at column n (no right strip, which
        # would stream A a second time); only n < 32 keeps the predicated right strip
        shift = 1 if (n % 32 and n >= 32) else 0
        assert p["shift"] == shift
        assert p["nright"] == (-(-m // 128) if (n % 32 and not shift) else 0)
        # likewise a ragged m >= 128 is covered by a full tile row shifted up to end at row m: no bottom strip
        shift_r = 1 if (m % 128 and m >= 128) else 0
        assert p["nbottom"] == (n // 32 + shift if (m % 128 and not shift_r) else 0)
        for ns, kc in ((p["nsI"], p["kcI"]), (p["nsB"], p["kcB"])):
            assert ns >= 1 and kc % 8 == 0 and ns * kc >= k and (ns - 1) * kc < k     # chunks tile [0, k) exactly
            assert ns == 1 or kc >= 64                                                   # no degenerate chunks
        full = (p["gmf"] + shift_r) * (p["gnf"] + p["shift"])
        if 0 < full < 512 and k >= 2048 and not flags:
            # at least half a round of the 512 resident workgroups, at most a few balanced rounds
            assert 256 <= full * p["nsI"] <= 4 * 512
        if p["zmI"] == 1:
            assert p["nsI"] >= 8 and k >= 2048
        # UPPER interior launches run over their live tiles only (mapping 2): row panel r keeps the column tiles >= 4 r
        assert (p["zmI"] == 2) == bool(flags)
        if flags:
            gmi, gni = p["gmf"] + shift_r, p["gnf"] + p["shift"]
            live = sum(max(0, gni - 4 * r) for r in range(gmi))
            assert live == sum(1 for r in range(gmi) for c in range(gni) if r * 128 < (n if (shift and c == gni - 1) else (c + 1) * 32))
            if 0 < live < 512 and k >= 2048:
                assert 256 <= live * p["nsI"] <= 4 * 512
    if trans == "N" and m > 100000:
        assert _gemm_plan(lib, trans, m, n, k)["nsI"] == 1      # thousands of tiles: no K split
    # the REAL (half-sphere) products run 128 x 64 tiles
    pr = _gemm_plan(lib, trans, m, n, k, 8)
    assert pr["bn"] == 64 and pr["gnf"] == n // 64 and pr["shift"] == (1 if (n % 64 and n >= 64) else 0)


@pytest.mark.parametrize("n", [1, 16, 33, 100, 259, 518, 777, 1509])
def test_jacobi_round_schedule(lib, n):
    """Host view of the blocked-Jacobi schedule (dense_kernels.hip): every round pairs each 16-wide block exactly
    once, a sweep (round -1 = neighbours, rounds 0 .. nb-2 = round-robin tournament) meets every block pair
    exactly once across its cross rounds, and the inverse map used by the look-ahead pair solve is consistent."""
    nb = C.c_int()
    check(lib.dftk_mi_jacobi_schedule_host(n, 0, C.byref(nb), None, None))
    nb = nb.value
    assert nb % 2 == 0 and nb >= 2 and nb * 16 >= n and (nb - 2) * 16 < max(n, 17)
    met = set()
    for rnd in range(-1, nb - 1):
        pairs = (C.c_int * nb)()
        where = (C.c_int * (2 * nb))()
        check(lib.dftk_mi_jacobi_schedule_host(n, rnd, C.byref(C.c_int()), pairs, where))
        pr = [(pairs[2 * k], pairs[2 * k + 1]) for k in range(nb // 2)]
        assert all(0 <= p < q < nb for p, q in pr)
        assert sorted(b for pq in pr for b in pq) == list(range(nb))          # a perfect matching of the blocks
        for blk in range(nb):                                                 # inverse map
            k, h = where[2 * blk], where[2 * blk + 1]
            assert pr[k][h] == blk
        if rnd < 0:
            assert pr == [(2 * k, 2 * k + 1) for k in range(nb // 2)]
        else:
            for pq in pr:
                assert pq not in met
                met.add(pq)
    assert len(met) == nb * (nb - 1) // 2                                      # all block pairs, once per sweep
    assert lib.dftk_mi_jacobi_schedule_host(n, nb - 1, C.byref(C.c_int()), (C.c_int * nb)(), None) < 0


def test_jacobi_lookahead_identity(lib):
    """NumPy twin of the look-ahead of k_jacobi_round (dense_kernels.hip), driven by the library's own schedule:
    the 2 x 2 block problem of round r, assembled from the matrix BEFORE round r-1 with last round's rotations of
    the two pairs the blocks belonged to (three 32 x 32 tiles, one 16 x 16 quadrant of each), equals the block
    problem read from the fully updated matrix."""
    n, JB = 100, 16
    nbc = C.c_int()
    check(lib.dftk_mi_jacobi_schedule_host(n, 0, C.byref(nbc), None, None))
    nb = nbc.value
    N = nb * JB
    rng = np.random.default_rng(7)
    A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    A = (A + A.conj().T) / 2

    def schedule(rnd):
        pairs, where = (C.c_int * nb)(), (C.c_int * (2 * nb))()
        check(lib.dftk_mi_jacobi_schedule_host(n, rnd, C.byref(C.c_int()), pairs, where))
        return ([(pairs[2 * k], pairs[2 * k + 1]) for k in range(nb // 2)],
                [(where[2 * b], where[2 * b + 1]) for b in range(nb)])

    def idx(p, q):      # global indices of the local 0..31 of a pair: block p first, then block q
        return np.r_[p * JB:(p + 1) * JB, q * JB:(q + 1) * JB]

    for prev, rnd in ((-1, 0), (0, 1), (3, 4), (nb - 2, -1)):
        pr_prev, where_prev = schedule(prev)
        U = [np.linalg.qr(rng.standard_normal((2 * JB, 2 * JB)) + 1j * rng.standard_normal((2 * JB, 2 * JB)))[0]
             for _ in pr_prev]
        J = np.zeros((N, N), complex)
        for (p, q), u in zip(pr_prev, U):
            J[np.ix_(idx(p, q), idx(p, q))] = u
        A_upd = J.conj().T @ A @ J                                  # what the update workgroups write
        for bp, bq in schedule(rnd)[0]:
            (ka, ha), (kb, hb) = where_prev[bp], where_prev[bq]
            Ia, Ib = idx(*pr_prev[ka]), idx(*pr_prev[kb])
            ua, ub = U[ka][:, JB * ha:JB * (ha + 1)], U[kb][:, JB * hb:JB * (hb + 1)]
            S = np.zeros((2 * JB, 2 * JB), complex)
            S[:JB, :JB] = ua.conj().T @ A[np.ix_(Ia, Ia)] @ ua
            S[:JB, JB:] = ua.conj().T @ A[np.ix_(Ia, Ib)] @ ub
            S[JB:, JB:] = ub.conj().T @ A[np.ix_(Ib, Ib)] @ ub
            S[JB:, :JB] = S[:JB, JB:].conj().T
            np.testing.assert_allclose(S, A_upd[np.ix_(idx(bp, bq), idx(bp, bq))], atol=1e-12)


@pytest.mark.parametrize("trans,m,n,k", [("C", 259, 259, 135491), ("C", 640, 259, 135491), ("C", 777, 777, 4000),
                                          ("N", 1500, 259, 259), ("N", 1000, 54, 777), ("C", 130, 33, 512),
                                          ("N", 127, 31, 64), ("C", 128, 64, 64), ("C", 5, 5, 100), ("N", 300, 1, 259)])
def test_zgemm_tiling_covers_every_entry_once(lib, trans, m, n, k):
    """The launches the planner describes -- full 128 x 32 tiles (a ragged last column as a tile shifted left
    that stores only its new columns, a ragged last row as a tile shifted up that stores only its new rows), right
    strip, bottom strip (m < 128 only) -- write every entry of C exactly once; with
    the upper-only flag every entry on or above the diagonal exactly once (tiles strictly below are skipped)."""
    BM = 128
    for flags in (0, 1):
        if flags and (trans != "C" or m != n):
            continue
        p = _gemm_plan(lib, trans, m, n, k, flags)
        bn, gmf, gnf, shift = p["bn"], p["gmf"], p["gnf"], p["shift"]
        gm = -(-m // BM)
        cover = np.zeros((m, n), dtype=int)

        def tile(tr, j_lo, j_hi, j_end_for_liveness):
            if flags and tr * BM >= j_end_for_liveness:
                return                                              # strictly below the diagonal: not computed
            cover[tr * BM:min(m, tr * BM + BM), j_lo:j_hi] += 1

        def column_tile(tr, tc):
            if shift and tc == gnf:                                 # shifted tile: spans [n - bn, n), stores [gnf bn, n)
                tile(tr, gnf * bn, n, n)
            else:
                tile(tr, tc * bn, min(n, tc * bn + bn), tc * bn + bn)

        shift_r = 1 if (m % BM and m >= BM) else 0                 # ragged last tile row as a tile shifted up
        for tr in range(gmf + shift_r):                             # interior launch
            for tc in range(gnf + shift):
                column_tile(tr, tc)
        assert p["nright"] in (0, gm) and p["nbottom"] == (0 if shift_r or not m % BM else gnf + shift)
        for e in range(p["nright"]):                                # border list: right strip, then bottom strip
            column_tile(e, gnf)
        for e in range(p["nbottom"]):
            column_tile(gmf, e)
        if flags:
            iu = np.triu_indices(m)
            assert np.all(cover[iu] == 1)
            assert cover.max() == 1
        else:
            assert np.all(cover == 1)


def test_bench_and_smoke_refuse_to_run_without_gpu():
    """No silent CPU path in the measured / smoke entry points: without a GPU they stop with a clear message."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]          # no measurement line
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], capture_output=True,
                       text=True, cwd=ROOT, timeout=300)
    assert r.returncode != 0 and "needs a GPU" in r.stderr


@pytest.mark.parametrize("Ecut,fft_size", [(4.0, (15, 15, 15)), (3.0, (15, 13, 13)), (4.0, (11, 13, 11))])
def test_host_mirror_planewave_basis_invariants(Ecut, fft_size):
    """The reference's PlaneWaveBasis tests (test/PlaneWaveBasis.jl:1-58) on the host mirror's descriptors: reciprocal
    lattice, G-vector bounds, `g_all[kpt.mapping] == G_vectors(kpt)`, cutoff respected -- and the sphere complete."""
    from test_oracle_golden import _check_basis_invariants, LATTICE as LAT
    model = dftk.Model(LAT, [], [], ("Kinetic",), n_electrons=2)
    basis = dftk.PlaneWaveBasis(model, Ecut, dftk.MonkhorstPack((2, 5, 5), (0.5, 0, 0)), fft_size=fft_size,
                                device="cpu", build_terms=False)
    assert basis.fft_size == tuple(fft_size) and len(basis.kpoints) == 50
    assert abs(sum(basis.kweights) - 1.0) < 1e-14
    _check_basis_invariants(basis, basis.kpoints, LAT, Ecut, to_numpy=lambda t: t.cpu().numpy())
    # identical descriptors to the oracle's (same k-point order, same spheres)
    ob = oracle.PlaneWaveBasis(oracle.Model(LAT, [], [], terms=("Kinetic",), n_electrons=2), Ecut,
                               oracle.MonkhorstPack((2, 5, 5), (0.5, 0, 0)), fft_size=fft_size)
    for k, ok in zip(basis.kpoints, ob.kpoints):
        assert np.allclose(k.coordinate, ok.coordinate) and np.array_equal(k.mapping, ok.mapping)


def test_host_mirror_fermi_level_reference_pins():
    """test/occupation.jl:100-139 on the host mirror's compute_occupation (Fermi-Dirac, emulated metal)."""
    from test_oracle_golden import MG_EIGENVALUES, MG_FERMI_DIRAC_PINS, LATTICE as LAT
    kc = [[i / 13.0, 0, 0] for i in range(12)]                      # 12 k-points of equal weight (values irrelevant)
    for T, ref in MG_FERMI_DIRAC_PINS:
        model = dftk.Model(LAT, [], [], ("Kinetic",), n_electrons=4, temperature=T, smearing="fermi_dirac")
        basis = dftk.PlaneWaveBasis(model, 3, dftk.ExplicitKpoints(kc, [1 / 12] * 12), fft_size=(9, 9, 9),
                                    device="cpu", build_terms=False)
        occ, eF = dftk.compute_occupation(basis, [np.array(e) for e in MG_EIGENVALUES], tol_n_elec=1e-10)
        assert abs(eF - ref) < 1e-12
        assert abs(sum(w * o.sum() for w, o in zip(basis.kweights, occ)) - 4.0) < 1e-9


def test_native_fermi_bisection_equals_the_interpreted_twin(monkeypatch):
    """``dftk_mi_fermi_bisection`` (the bisection loop of FermiBisection, occupation.jl:99-132, as one host-only library
    call) against the interpreted loop of ``compute_occupation`` (``DFTK_MI_TORCH_LOCAL=1``): Fermi level to the last few
    ulps (the two sum the occupations in different orders), electron count to 1e-12, for both smearings and ragged band
    counts; the reference pins of test/occupation.jl run through the native path in the test above."""
    from test_oracle_golden import MG_EIGENVALUES, LATTICE as LAT
    kc = [[i / 13.0, 0, 0] for i in range(12)]
    rng = np.random.default_rng(3)
    for smearing in ("fermi_dirac", "gaussian"):
        for T in (1e-3, 0.01, 0.05):
            model = dftk.Model(LAT, [], [], ("Kinetic",), n_electrons=4, temperature=T, smearing=smearing)
            basis = dftk.PlaneWaveBasis(model, 3, dftk.ExplicitKpoints(kc, [1 / 12] * 12), fft_size=(9, 9, 9),
                                        device="cpu", build_terms=False)
            ev = [np.array(e) + 0.01 * rng.standard_normal(len(e)) for e in MG_EIGENVALUES]
            ev = [np.sort(e)[: len(e) - (i % 2)] for i, e in enumerate(ev)]            # ragged band counts
            monkeypatch.delenv("DFTK_MI_TORCH_LOCAL", raising=False)
            occ, eF = dftk.compute_occupation(basis, ev, tol_n_elec=1e-10)
            monkeypatch.setenv("DFTK_MI_TORCH_LOCAL", "1")
            occ_t, eF_t = dftk.compute_occupation(basis, ev, tol_n_elec=1e-10)
            monkeypatch.delenv("DFTK_MI_TORCH_LOCAL", raising=False)
            assert abs(eF - eF_t) < 1e-13 * max(1.0, abs(eF_t)), (smearing, T, eF, eF_t)
            assert abs(sum(w * o.sum() for w, o in zip(basis.kweights, occ)) - 4.0) < 1e-12
            for a, b in zip(occ, occ_t):
                np.testing.assert_allclose(a, b, rtol=0, atol=1e-11)
    # argument checks: no temperature, unknown smearing, inverted bracket
    import ctypes as C
    lib = dftk.load_library()
    nb = np.array([2], dtype=np.int32)
    e = np.array([0.0, 1.0])
    w = np.array([1.0])
    out = C.c_double()
    for args in ((1, 0.0, 0.0, 1.0), (7, 0.01, 0.0, 1.0), (2, 0.01, 1.0, 0.0)):
        assert lib.dftk_mi_fermi_bisection(1, nb.ctypes.data, e.ctypes.data, w.ctypes.data, args[0], args[1], 2.0, 2.0, args[2],
                                           args[3], C.byref(out)) != 0


def test_split_evenly_and_nuclear_energies():
    """test/split_evenly.jl (concatenation of the parts is the range) and the ABINIT pins of test/energy_nuclear.jl
    on the host mirror's set-up code."""
    for n, parts in [(12, 4), (31, 7), (168, 16), (14, 16)]:
        chunks = dftk.split_evenly(n, parts)
        assert len(chunks) == parts and [i for c in chunks for i in c] == list(range(n))
        sizes = [len(c) for c in chunks]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    from dftk_jl_amd.terms import energy_ewald, energy_psp_correction
    lat, atoms, pos = dftk.silicon_cell()
    a = 5.131570667152971
    lat = np.array([[0, a, a], [a, 0, a], [a, a, 0.0]])
    assert abs(energy_ewald(lat, [4, 4], pos) - (-8.39789357839024)) < 1e-10
    model = dftk.Model(lat, atoms, pos, ("PspCorrection",))
    assert abs(energy_psp_correction(model) - (-0.294622067023269)) < 1e-10


def test_host_mirror_entropy_term():
    """Entropy term -TS (terms/entropy.jl, standard_models.jl:56-58): present exactly when temperature != 0, and the
    mirror's smearing entropy equals the oracle's (s' = x f' is checked on the oracle)."""
    from dftk_jl_amd.terms import smearing_entropy
    from oracle.terms import smearing_entropy as oracle_entropy
    lat, atoms, pos = dftk.silicon_cell()
    assert "Entropy" not in dftk.model_DFT(lat, atoms, pos).term_types
    m = dftk.model_DFT(lat, atoms, pos, temperature=1e-3, smearing="gaussian")
    assert m.term_types[-1] == "Entropy"
    x = np.linspace(-40, 40, 2001)
    for kind in ("none", "fermi_dirac", "gaussian"):
        np.testing.assert_allclose(smearing_entropy(kind, x), oracle_entropy(kind, x), rtol=0, atol=1e-16)
    assert np.all(smearing_entropy("fermi_dirac", x) >= 0) and smearing_entropy("fermi_dirac", np.array([0.0]))[0] == pytest.approx(np.log(2))


def test_symmetry_host_logic_matches_oracle():
    """The mirror's symmetry detection / k-mesh reduction (dftk.jl_amd/symmetry.py) against the oracle's (which is
    pinned to the reference's Spglib counts): same operations, same irreducible points and weights, same FFT size."""
    import oracle
    from oracle import symmetry as osy
    lat, atoms, pos = dftk.silicon_cell((2, 1, 1))
    m = dftk.model_DFT(lat, atoms, pos, symmetries=True)
    oops = osy.symmetry_operations(lat, [list(range(len(pos)))], pos)
    assert len(m.symmetries) == len(oops) and m.symmetries[0].isone()
    for a in m.symmetries:      # two different searches (reduced-cell scan vs. vector shells): same SET of operations
        twins = [b for b in oops if np.array_equal(a.W, b.W) and np.allclose(np.mod(a.w - b.w + 0.5, 1.0), 0.5, atol=1e-9)]
        assert len(twins) == 1 and np.allclose(np.mod(a.tau - twins[0].tau + 0.5, 1.0), 0.5, atol=1e-9)
    dftk.check_group(m.symmetries)
    lat, atoms, pos = dftk.silicon_cell()
    m = dftk.model_DFT(lat, atoms, pos, symmetries=True)
    b = dftk.PlaneWaveBasis(m, 15, dftk.MonkhorstPack((4, 4, 4)), device="cpu", build_terms=False)
    okc, okw = osy.irreducible_kcoords((4, 4, 4), osy.symmetries_preserving_kgrid(
        osy.symmetry_operations(lat, [[0, 1]], pos), (4, 4, 4)))
    assert b.fft_size == (30, 30, 30) and len(b.kpoints) == 8 == len(okc) and len(b.symmetries) == 48
    assert np.allclose(b.kweights, okw) and all(np.allclose(k.coordinate, q) for k, q in zip(b.kpoints, okc))
    b2 = dftk.PlaneWaveBasis(m, 15, dftk.MonkhorstPack((4, 4, 4)), device="cpu", build_terms=False,
                             use_symmetries_for_kpoint_reduction=False)
    assert len(b2.kpoints) == 64 and len(b2.symmetries) == 48
    m0 = dftk.model_DFT(lat, atoms, pos)                                   # default: no symmetries
    b0 = dftk.PlaneWaveBasis(m0, 15, dftk.MonkhorstPack((4, 4, 4)), device="cpu", build_terms=False)
    assert b0.fft_size == (27, 27, 27) and len(b0.kpoints) == 64 and len(b0.symmetries) == 1


REF_SI_A = 5.131570667152971
REF_CELLS = {   # test/testcases.jl:10-28 (silicon), :49-58 (magnesium), :107-116 (platinum_hcp)
    "silicon": (np.array([[0, REF_SI_A, REF_SI_A], [REF_SI_A, 0, REF_SI_A], [REF_SI_A, REF_SI_A, 0.0]]),
                [np.ones(3) / 8, -np.ones(3) / 8]),
    "magnesium": (np.array([[-3.0179389205999998, -3.0179389205999998, 0.0], [-5.2272235447000002, 5.2272235447000002, 0.0],
                            [0.0, 0.0, -9.7736219469000005]]), [np.array([2 / 3, 1 / 3, 1 / 4]), np.array([1 / 3, 2 / 3, 3 / 4])]),
    "platinum_hcp": (np.array([[10.0, 0, 0], [5.0, 8.66025403784439, 0], [0, 0, 16.33]]), [np.zeros(3), np.ones(3) / 3]),
}
# test/bzmesh.jl:62-80: (testcase, kgrid, irreducible k-points Spglib finds, supercell, kshift)
REF_IRREDUCIBLE_COUNTS = [
    ("silicon", (1, 1, 1), 1, (1, 1, 1), (0, 0, 0)), ("silicon", (1, 1, 5), 3, (1, 1, 1), (0, 0, 0)),
    ("silicon", (2, 3, 2), 6, (1, 1, 1), (0, 0, 0)), ("silicon", (3, 3, 3), 4, (1, 1, 1), (0, 0, 0)),
    ("silicon", (2, 3, 4), 14, (1, 1, 1), (0, 0, 0)), ("silicon", (9, 11, 13), 644, (1, 1, 1), (0, 0, 0)),
    ("silicon", (3, 3, 3), 6, (1, 1, 1), (.5, .5, .5)), ("silicon", (3, 3, 3), 6, (1, 1, 1), (.5, 0, .5)),
    ("silicon", (3, 3, 3), 6, (1, 1, 1), (0, .5, 0)),
    ("silicon", (1, 4, 4), 7, (2, 1, 1), (0, 0, 0)), ("silicon", (1, 16, 16), 73, (4, 1, 1), (0, 0, 0)),
    ("magnesium", (2, 3, 2), 8, (1, 1, 1), (0, 0, 0)), ("magnesium", (3, 3, 3), 6, (1, 1, 1), (0, 0, 0)),
    ("magnesium", (2, 3, 4), 12, (1, 1, 1), (0, 0, 0)), ("magnesium", (9, 11, 13), 350, (1, 1, 1), (0, 0, 0)),
    ("platinum_hcp", (5, 5, 5), 63, (1, 1, 1), (0, 0, 0)),
]


def _supercell(lat, pos, sc):
    """Plain repetition of the cell (src/supercell.jl:5-20); only the SET of atoms matters to the symmetry search."""
    sc = np.asarray(sc)
    out = [(np.asarray(p) + np.array([i, j, k])) / sc for p in pos for k in range(sc[2]) for j in range(sc[1])
           for i in range(sc[0])]
    return lat * sc[None, :], out


def test_mirror_symmetry_search_reproduces_the_reference_spglib_counts():
    """The PRODUCT's symmetry search (dftk.jl_amd/symmetry.py, not the oracle's) against the numbers the reference pins
    from Spglib: every ``test_reduction`` case of test/bzmesh.jl:62-80 (silicon incl. shifted meshes and 2x1x1 / 4x1x1
    supercells, hcp magnesium, hcp platinum) with the reference's own reconstruction check (:45-57: the images of the
    irreducible points are the whole mesh), the k-weights of test/testcases.jl:24-28 and :59-65, and the 48 operations
    of the CuO2 cell of test/symmetry_issues.jl:9-24."""
    from dftk_jl_amd import symmetry as sy
    for name, size, n_irr, sc, shift in REF_IRREDUCIBLE_COUNTS:
        lat, pos = REF_CELLS[name]
        if sc != (1, 1, 1):
            lat, pos = _supercell(lat, pos, sc)
        ops = sy.symmetry_operations(lat, [list(range(len(pos)))], pos)
        assert ops[0].isone()
        keep = sy.symmetries_preserving_kgrid(ops, size, shift)
        sy.check_group(keep)
        kc, kw = sy.irreducible_kcoords(size, keep, shift)
        assert len(kc) == n_irr, (name, size, shift, sc, len(kc))
        assert abs(sum(kw) - 1) < 1e-13
        red = {sy._grid_key(k, np.array(size), shift) for k in sy.reducible_kcoords(size, shift)}
        img = {sy._grid_key(s.S @ k, np.array(size), shift) for k in kc for s in keep}
        assert img == red, (name, size)
    lat, pos = REF_CELLS["silicon"]
    ops = sy.symmetry_operations(lat, [[0, 1]], pos)
    assert len(ops) == 48
    assert sorted(np.rint(np.array(sy.irreducible_kcoords((3, 3, 3), ops)[1]) * 27).astype(int)) == [1, 6, 8, 12]
    lat, pos = REF_CELLS["magnesium"]
    ops = sy.symmetry_operations(lat, [[0, 1]], pos)
    assert len(ops) == 24
    assert sorted(np.rint(np.array(sy.irreducible_kcoords((3, 3, 3), ops)[1]) * 27).astype(int)) == [1, 2, 2, 4, 6, 12]
    a = 4.474
    latc = np.array([[0, a, a], [a, 0, a], [a, a, 0.0]]).T
    frac = [np.linalg.solve(latc, c) for c in (np.zeros(3), np.array([6.711, 2.237, 6.711]), np.array([6.711, 2.237, 2.237]))]
    assert len(sy.symmetry_operations(latc, [[0], [1, 2]], frac)) == 48
    # a skewed (non-reduced) description of the same silicon lattice: the cell reduction must find all 48 again
    U = np.array([[1, 2, 0], [0, 1, 3], [0, 0, 1]])
    lat, pos = REF_CELLS["silicon"]
    assert len(sy.symmetry_operations(lat @ U, [[0, 1]], [np.linalg.solve(U.astype(float), p) for p in pos])) == 48
    # check_group is an explicit exception (survives python -O)
    with pytest.raises(ValueError):
        sy.check_group(ops[:5])


def test_scfres_dict_layout_on_host():
    """scfres_to_dict (src/input_output.jl:345-386) on a hand-made result: keys and Julia nesting."""
    import torch
    lat, atoms, pos = dftk.silicon_cell()
    b = dftk.PlaneWaveBasis(dftk.model_DFT(lat, atoms, pos), 5, dftk.MonkhorstPack((2, 1, 1)), device="cpu",
                            build_terms=False)
    E = dftk.terms.Energies(Kinetic=1.0, Hartree=2.0)
    res = dict(basis=b, eigenvalues=[np.arange(5.0)] * 2, occupation=[np.array([2, 2, 2, 2, 0.0])] * 2, eF=0.3,
               diagonalization=dict(n_matvec=10, converged=True, residual_norms=[np.zeros(5)] * 2, n_iter=[3, 4]),
               rho=torch.zeros(b.fft_size[::-1]), energies=E, converged=True, history_drho=[1e-3, 1e-7], n_iter=2,
               n_matvec=20, history_Etot=[-1.0, -1.1], n_bands_converge=4,
               psi=[torch.zeros(5, k.n_G) for k in b.kpoints])
    d = dftk.scfres_to_dict(res, save_psi=True)
    assert np.array(d["eigenvalues"]).shape == (1, 2, 5) and np.array(d["ρ"]).shape == (1,) + b.fft_size[::-1]
    assert d["energies"] == {"Kinetic": 1.0, "Hartree": 2.0, "total": 3.0} and d["n_kpoints"] == 2
    assert d["norm_Δρ"] == 1e-7 and d["εF"] == 0.3 and d["diagonalization"]["n_iter"] == [[3, 4]]
    import json
    json.dumps(d)
    assert d["kgrid"] == "MonkhorstPack([2, 1, 1])" and d["symmetries_rotations"] == [np.eye(3, dtype=int).tolist()]
    assert d["use_symmetries_for_kpoint_reduction"] is True and d["symmetries_respect_rgrid"] is True


def test_basis_dict_reports_the_symmetries_it_used():
    """``todict(basis)`` (input_output.jl:181-204): the k-points written are the IRREDUCIBLE ones, so the symmetry
    operations (W as its list of columns, w), the two constructor flags and the grid as given must travel with them --
    DFTK-side post-processing unfolds k-points / densities from exactly these keys."""
    from dftk_jl_amd.io import basis_to_dict
    lat, atoms, pos = dftk.silicon_cell()
    m = dftk.model_DFT(lat, atoms, pos, symmetries=True)
    b = dftk.PlaneWaveBasis(m, 5, dftk.MonkhorstPack((4, 4, 4)), device="cpu", build_terms=False)
    d = basis_to_dict(b)
    assert d["kgrid"] == "MonkhorstPack([4, 4, 4])" and d["n_kpoints"] == len(d["kcoords"]) < 64
    assert d["use_symmetries_for_kpoint_reduction"] is True and d["symmetries_respect_rgrid"] is True
    assert len(d["symmetries_rotations"]) == len(b.symmetries) == 48 == len(d["symmetries_translations"])
    for s, Wl, wl in zip(b.symmetries, d["symmetries_rotations"], d["symmetries_translations"]):
        assert np.array_equal(np.array(Wl).T, s.W) and np.allclose(wl, s.w)      # Julia nesting: list of columns
    # unfolding the irreducible points with S = W' reproduces the full mesh with the stored weights
    full = {tuple(np.round(np.mod(np.array(Wl) @ np.array(k) + 0.5, 1.0) - 0.5, 9) % 1.0)
            for k in d["kcoords"] for Wl in d["symmetries_rotations"]}
    assert len(full) == 64 and abs(sum(d["kweights"]) - 1.0) < 1e-14
    sh = dftk.PlaneWaveBasis(m, 5, dftk.MonkhorstPack((2, 2, 2), kshift=(0.5, 0.5, 0.5)), device="cpu", build_terms=False,
                             use_symmetries_for_kpoint_reduction=False, fft_size=(20, 20, 20))
    ds = basis_to_dict(sh)
    assert ds["kgrid"] == "MonkhorstPack([2, 2, 2], [0.5, 0.5, 0.5])" and ds["n_kpoints"] == 8
    assert ds["use_symmetries_for_kpoint_reduction"] is False and ds["symmetries_respect_rgrid"] is False


def test_kpoint_sphere_host_matches_oracle_and_torch():
    """dftk_mi_kpoint_sphere_host (src/Kpoint.jl:20-41 behind the ABI; a host routine, no GPU needed): mapping,
    kinetic multipliers and integer G vectors identical to the oracle's and to the mirror's torch enumeration, for
    Gamma, a general k-point, an anisotropic cell and a cube with an even axis."""
    import ctypes as C
    import oracle
    from dftk_jl_amd._lib import check
    lib = dftk.load_library()
    cases = [((1, 1, 1), 9.0, [0.0, 0.0, 0.0], None), ((2, 1, 1), 7.0, [0.25, -0.5, 0.125], None),
             ((1, 1, 2), 6.0, [1 / 3, 0.0, -1 / 3], (16, 18, 40))]
    for sc, ecut, k, fft in cases:
        lat, atoms, pos = dftk.silicon_cell(sc)
        m = dftk.model_DFT(lat, atoms, pos)
        b = dftk.PlaneWaveBasis(m, ecut, dftk.ExplicitKpoints([k], [1.0]), device="cpu", build_terms=False, fft_size=fft)
        kp = b.kpoints[0]
        nx, ny, nz = b.fft_size
        B = np.asfortranarray(m.recip_lattice)
        kk = np.asarray(k, dtype=float)
        n = C.c_int64()
        check(lib.dftk_mi_kpoint_sphere_host(nx, ny, nz, B.ctypes.data, kk.ctypes.data, ecut, 0, C.byref(n), None, None, None))
        assert n.value == kp.n_G
        mp, kin, G = np.zeros(n.value, dtype=np.int64), np.zeros(n.value), np.zeros((n.value, 3), dtype=np.int32)
        check(lib.dftk_mi_kpoint_sphere_host(nx, ny, nz, B.ctypes.data, kk.ctypes.data, ecut, n.value, C.byref(n),
                                             mp.ctypes.data, kin.ctypes.data, G.ctypes.data))
        assert np.array_equal(mp, kp.mapping) and np.array_equal(G, kp.G_vectors.numpy())
        assert np.array_equal(kin, kp.kinetic.numpy())
        Si = oracle.ElementPsp("Si", oracle.load_psp_hgh("Si", "lda"))
        ob = oracle.PlaneWaveBasis(oracle.model_DFT(lat, [Si] * len(pos), pos), ecut, oracle.ExplicitKpoints([k], [1.0]),
                                   fft_size=fft, build_terms=False)
        assert np.array_equal(mp, ob.kpoints[0].mapping) and np.array_equal(G, ob.kpoints[0].G_vectors)
        assert lib.dftk_mi_kpoint_sphere_host(nx, ny, nz, B.ctypes.data, kk.ctypes.data, ecut, 3, C.byref(n),
                                              mp.ctypes.data, None, None) == -1          # buffers too small: reported


def test_bench_cli_contract_without_gpu():
    """bench.py parses the driver's flags and refuses to run without a GPU (no CPU fallback, no silent skip)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True)
    assert h.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--mode", "--supercell", "--no-cpu-baseline"):
        assert flag in h.stdout
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"],
                           capture_output=True, text=True)
        assert r.returncode != 0 and "needs a GPU" in (r.stdout + r.stderr)


def test_shard_plan_transposes_slabs_to_bands_and_back():
    """Plane-wave sharding, host logic (dftk_mi_shard_plan_host = the plan the device Transposer uses): emulate the
    slab -> band all-to-all of apply_H for p ranks in NumPy and check that every rank ends up with all rows of exactly
    its bands, and that the way back restores the slabs -- for even / ragged splits, more ranks than bands, 1 rank."""
    import ctypes as C
    from dftk_jl_amd._lib import check
    from dftk_jl_amd.comm import split_evenly
    lib = dftk.load_library()
    rng = np.random.default_rng(0)
    for n_G, p, nb in [(101, 2, 7), (64, 4, 8), (37, 3, 2), (50, 1, 5), (23, 8, 3)]:
        rows = np.array([r.start for r in split_evenly(n_G, p)] + [n_G], dtype=np.int64)
        X = rng.standard_normal((n_G, nb)) + 1j * rng.standard_normal((n_G, nb))
        plans = []
        for me in range(p):
            c0 = np.zeros(p + 1, dtype=np.int32)
            arrs = [np.zeros(p, dtype=np.int64) for _ in range(4)]
            check(lib.dftk_mi_shard_plan_host(p, me, nb, rows.ctypes.data, c0.ctypes.data, *[a.ctypes.data for a in arrs]))
            plans.append((c0, *arrs))
        assert all(np.array_equal(pl[0], plans[0][0]) for pl in plans) and plans[0][0][-1] == nb
        c0 = plans[0][0]
        # packed slabs (column-major n_loc x nb) and the emulated all-to-all
        slabs = [X[rows[r]:rows[r + 1], :].flatten(order="F") for r in range(p)]
        bands = []
        for me in range(p):
            _, so, sc, bo, bc = plans[me]
            mine = c0[me + 1] - c0[me]
            recv = np.zeros(n_G * mine, dtype=complex)
            for r in range(p):
                _, so_r, sc_r, _, _ = plans[r]
                piece = slabs[r][so_r[me]:so_r[me] + sc_r[me]]           # what rank r sends to me
                assert len(piece) == bc[r]
                recv[bo[r]:bo[r] + bc[r]] = piece
            full = np.zeros((n_G, mine), dtype=complex)
            for r in range(p):                                             # the device's strided copies
                nr = rows[r + 1] - rows[r]
                full[rows[r]:rows[r + 1], :] = recv[bo[r]:bo[r] + bc[r]].reshape((nr, mine), order="F")
            assert np.array_equal(full, X[:, c0[me]:c0[me + 1]])
            bands.append(recv)
        for me in range(p):                                                # and back: band pieces -> slab columns
            _, so, sc, bo, bc = plans[me]
            back = np.zeros((rows[me + 1] - rows[me]) * nb, dtype=complex)
            for s_ in range(p):
                _, _, _, bo_s, bc_s = plans[s_]
                back[so[s_]:so[s_] + sc[s_]] = bands[s_][bo_s[me]:bo_s[me] + bc_s[me]]
            assert np.array_equal(back, slabs[me])


PW_GLOO_WORKER = r"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch, torch.distributed as dist
import dftk_jl_amd as dftk
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["PORT"],
                        rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
comm = dftk.KptComm.from_torch()
handle = comm.abi_handle(0)                     # host-staged dftk_mi_comm over gloo: no GPU needed to create it
lib = dftk.load_library()
assert comm._abi_kind == "host" and lib.dftk_mi_comm_size(handle) == comm.size and lib.dftk_mi_comm_rank(handle) == comm.rank
allreduce, alltoallv = comm._callbacks
buf = np.arange(6, dtype=np.float64) * (comm.rank + 1)
assert allreduce(None, buf.ctypes.data_as(C.POINTER(C.c_double)), 6) == 0
assert np.array_equal(buf, np.arange(6) * sum(range(1, comm.size + 1)))
# ragged all-to-all: rank r sends (r + 1) * (s + 1) doubles with value 100 r + s to rank s
p, me = comm.size, comm.rank
scnt = np.array([(me + 1) * (s + 1) for s in range(p)], dtype=np.uint64); soff = np.concatenate([[0], np.cumsum(scnt)[:-1]]).astype(np.uint64)
rcnt = np.array([(r + 1) * (me + 1) for r in range(p)], dtype=np.uint64); roff = np.concatenate([[0], np.cumsum(rcnt)[:-1]]).astype(np.uint64)
send = np.concatenate([np.full(int(scnt[s]), 100.0 * me + s) for s in range(p)])
recv = np.zeros(int(rcnt.sum()))
sz = C.POINTER(C.c_size_t); dp = C.POINTER(C.c_double)
assert alltoallv(None, send.ctypes.data_as(dp), scnt.ctypes.data_as(sz), soff.ctypes.data_as(sz),
                 recv.ctypes.data_as(dp), rcnt.ctypes.data_as(sz), roff.ctypes.data_as(sz)) == 0
want = np.concatenate([np.full(int(rcnt[r]), 100.0 * r + me) for r in range(p)])
assert np.array_equal(recv, want)
assert comm.sum_scalars([1.0, float(me)]) == [float(p), float(sum(range(p)))]
dist.barrier(); dist.destroy_process_group()
print("rank", me, "ok")
"""


def test_host_staged_communicator_callbacks_gloo_world2(tmp_path):
    """The host-staged communicator (dftk_mi_comm_create_host + the gloo callbacks of KptComm: what a plane-wave
    sharded block reduces and transposes through when the process group is not nccl) with 2 ranks on the CPU."""
    script = tmp_path / "pw_worker.py"
    script.write_text(PW_GLOO_WORKER)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", PORT=port, REPO=ROOT, MASTER_ADDR="127.0.0.1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


def test_gamma_pair_tables_host(lib):
    """Host side of the Gamma-real extension (dftk_mi_gamma_tables_host): every row of a k = 0 sphere is paired with
    its -G row exactly once, G = 0 first and alone; a sphere around k != 0 and a cube with Nyquist points are refused."""
    import oracle
    lat = 10.26 / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
    model = oracle.Model(lat, [], [], terms=("Kinetic",), n_electrons=8)
    for fft in ((21, 21, 21), (24, 25, 27), (30, 30, 30)):
        basis = oracle.PlaneWaveBasis(model, 10, oracle.ExplicitKpoints([[0, 0, 0], [0.25, 0, 0]], [0.5, 0.5]), fft_size=fft)
        nx, ny, nz = fft
        m = np.ascontiguousarray(basis.kpoints[0].mapping, dtype=np.int64)
        n = len(m)
        cnt = C.c_int64()
        check(lib.dftk_mi_gamma_tables_host(nx, ny, nz, n, m.ctypes.data, C.byref(cnt), None, None))
        nh = cnt.value
        assert 2 * nh - 1 == n
        g, mg = np.zeros(nh, dtype=np.int32), np.zeros(nh, dtype=np.int32)
        check(lib.dftk_mi_gamma_tables_host(nx, ny, nz, n, m.ctypes.data, C.byref(cnt), g.ctypes.data, mg.ctypes.data))
        assert g[0] == mg[0] == 0 and np.all(np.diff(g) > 0) and np.all(mg[1:] > g[1:])
        assert sorted(np.concatenate([g, mg[1:]]).tolist()) == list(range(n))          # a partition of the sphere
        G = basis.kpoints[0].G_vectors
        assert np.array_equal(G[g], -G[mg])
        m2 = np.ascontiguousarray(basis.kpoints[1].mapping, dtype=np.int64)               # k != 0: no inversion symmetry
        st = lib.dftk_mi_gamma_tables_host(nx, ny, nz, len(m2), m2.ctypes.data, C.byref(cnt), None, None)
        assert st == -1 and b"gamma_real" in lib.dftk_mi_last_error()
    full = np.arange(8 ** 3, dtype=np.int64)
    assert lib.dftk_mi_gamma_tables_host(8, 8, 8, len(full), full.ctypes.data, C.byref(cnt), None, None) == -1
    assert b"Nyquist" in lib.dftk_mi_last_error()


def test_gamma_half_format_restriction_has_the_oracle_spectrum(lib):
    """The claim behind the Gamma-real extension, checked against the ORACLE on the CPU: in the half-sphere format
    (pair tables of the library, row 0 = x(0), rows j > 0 = sqrt(2) x(G_j), 2 n_half - 1 real unknowns) the oracle's
    Gamma-point Hamiltonian is a REAL SYMMETRIC matrix with exactly the eigenvalues -- and multiplicities -- of the
    complex Hermitian one, and plain real dot products of half-format vectors are the complex inner products."""
    Si = oracle.ElementPsp("Si", oracle.load_psp_hgh("Si", "lda"))
    lat = 10.26 / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
    model = oracle.Model(lat, [Si, Si], [np.ones(3) / 8, -np.ones(3) / 8], terms=("Kinetic", "AtomicLocal", "AtomicNonlocal"))
    basis = oracle.PlaneWaveBasis(model, 4, oracle.ExplicitKpoints([[0, 0, 0]], [1.0]), fft_size=(15, 15, 15))
    _, ham = oracle.energy_hamiltonian(basis, None, None)
    H = ham[0].to_dense()
    n = H.shape[0]
    nx, ny, nz = basis.fft_size
    m = np.ascontiguousarray(basis.kpoints[0].mapping, dtype=np.int64)
    cnt = C.c_int64()
    check(lib.dftk_mi_gamma_tables_host(nx, ny, nz, n, m.ctypes.data, C.byref(cnt), None, None))
    nh = cnt.value
    g, mg = np.zeros(nh, dtype=np.int32), np.zeros(nh, dtype=np.int32)
    check(lib.dftk_mi_gamma_tables_host(nx, ny, nz, n, m.ctypes.data, C.byref(cnt), g.ctypes.data, mg.ctypes.data))
    # E: real unknowns (x0; Re, Im of sqrt(2) x(G_j)) -> full complex vector
    E = np.zeros((n, 2 * nh - 1), dtype=complex)
    E[g[0], 0] = 1.0
    for j in range(1, nh):
        E[g[j], 2 * j - 1] = E[mg[j], 2 * j - 1] = 1 / np.sqrt(2)
        E[g[j], 2 * j], E[mg[j], 2 * j] = 1j / np.sqrt(2), -1j / np.sqrt(2)
    G = E.conj().T @ E
    assert np.abs(G - np.eye(2 * nh - 1)).max() < 1e-14                  # real dot products = complex inner products
    Hr = E.conj().T @ H @ E
    assert np.abs(Hr.imag).max() < 1e-12 * np.abs(Hr).max()              # H maps real fields to real fields
    np.testing.assert_allclose(np.linalg.eigvalsh(Hr.real), np.linalg.eigvalsh(H), atol=1e-11)


def test_default_diagtolalg_per_model_class():
    """``default_diagtolalg`` (scf_callbacks.jl:220-230) + ``determine_diagtol`` (:196-212), mirror and oracle:
    models with a nonlinear term (Hartree / Xc -- every DFT model) diagonalise the first TWO steps (``n_iter <= 1``)
    to ``min(6 * 0.005, 5 * 0.005) = 0.025``; only linear models (core Hamiltonian) take ``tol / 5`` there; an ``Xc``
    term without functionals is a ``TermNoop`` (xc.jl:33) and does not make the model nonlinear."""
    from types import SimpleNamespace
    from dftk_jl_amd.scf import default_diagtolalg, determine_diagtol
    from oracle.scf import default_diagtol_params, determine_diagtol as oracle_determine
    lat, atoms, pos = dftk.silicon_cell()
    olat, oatoms, opos = oracle.basis.silicon_primitive(a=10.26, functional="lda")
    tol = 1e-6
    cases = [
        ("dft", dftk.model_DFT(lat, atoms, pos), oracle.model_DFT(olat, oatoms, opos), 0.025),
        ("dft_T", dftk.model_DFT(lat, atoms, pos, temperature=1e-3, smearing="gaussian"),
         oracle.model_DFT(olat, oatoms, opos, temperature=1e-3, smearing="gaussian"), 0.025),
        ("hartree_only", dftk.model_atomic(lat, atoms, pos, extra_terms=("Hartree",)),
         oracle.model_atomic(olat, oatoms, opos, extra_terms=("Hartree",)), 0.025),
        ("core", dftk.model_atomic(lat, atoms, pos), oracle.model_atomic(olat, oatoms, opos), tol / 5),
        ("noop_xc", dftk.model_atomic(lat, atoms, pos, extra_terms=("Xc",), functionals=()),
         oracle.model_atomic(olat, oatoms, opos, extra_terms=("Xc",), functionals=()), tol / 5),
    ]
    hist = [3e-2, 4e-3, 7e-3]
    for name, m, om, first in cases:
        f = default_diagtolalg(SimpleNamespace(model=m), tol)
        okw = default_diagtol_params(om, tol)
        for n_iter in (0, 1):
            assert f(n_iter, hist[:n_iter]) == pytest.approx(first, rel=0, abs=0), name
            assert oracle_determine(n_iter, hist[:n_iter], **okw) == f(n_iter, hist[:n_iter]), name
        # afterwards: ratio_rhodiff * min(history), clamped to [100 eps, diagtol_max]; never grows again
        assert f(2, hist[:2]) == pytest.approx(0.2 * 4e-3) == oracle_determine(2, hist[:2], **okw)
        assert f(3, hist) == pytest.approx(0.2 * 4e-3)
        assert f(2, [1.0, 1.0]) == 0.005 and f(2, [1e-20, 1.0]) == 100 * np.finfo(float).eps
    assert determine_diagtol(0, [], diagtol_first=1.0) == 0.025          # min(diagtol_first, 5 diagtol_max)
    fx = default_diagtolalg(SimpleNamespace(model=SimpleNamespace(term_types=("Kinetic", "ExactExchange"),
                                                                  functionals=())), tol)
    assert fx(0, []) == 0.025 and fx(2, [1e-2, 1e-2]) == pytest.approx(5e-4 * 1e-2)
    assert default_diagtol_params(SimpleNamespace(terms=("Kinetic", "ExactExchange"), functionals=()), tol) == dict(ratio=5e-4)


def test_multi_k_entries_argument_checks_without_gpu(lib):
    """The multi-k entries (``dftk_mi_lobpcg_multi``, ``dftk_mi_density_accumulate_multi``, ``dftk_mi_band_kinetic_multi``,
    ``dftk_mi_kblocks_set_potential``): empty batches are no-ops, missing tables are argument errors -- decided on the
    host before any device call; the counters of the last batched call are readable at any time."""
    i64 = C.c_int64
    assert lib.dftk_mi_lobpcg_multi(0, None, 4, None, None, 1e-6, 1, 10, 0, 1, None, None, None, None, None, None, None) == 0
    assert lib.dftk_mi_lobpcg_multi(2, None, 4, None, None, 1e-6, 1, 10, 0, 1, None, None, None, None, None, None, None) != 0
    assert lib.dftk_mi_lobpcg_multi(-1, None, 4, None, None, 1e-6, 1, 10, 0, 1, None, None, None, None, None, None, None) != 0
    rho = (C.c_double * 8)()
    assert lib.dftk_mi_density_accumulate_multi(0, None, None, None, None, None, rho) == 0
    assert lib.dftk_mi_density_accumulate_multi(1, None, None, None, None, None, rho) != 0
    assert lib.dftk_mi_band_kinetic_multi(0, None, None, None, None, None) == 0
    assert lib.dftk_mi_band_kinetic_multi(3, None, None, None, None, None) != 0
    assert lib.dftk_mi_kblocks_set_potential(0, None, rho) == 0 and lib.dftk_mi_kblocks_set_potential(1, None, rho) != 0
    r, o, m, q = i64(-1), i64(-1), i64(-1), i64(-1)
    assert lib.dftk_mi_batch_stats(C.byref(r), C.byref(o), C.byref(m), C.byref(q)) == 0
    assert min(r.value, o.value, m.value, q.value) >= 0


def _reg_sizes():
    """The instantiated four-step factorisations n = (R1A R1B)(R2A R2B) of the register-resident z kernels, read from
    REG_SIZES in csrc/fft_kernels.hip."""
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "dftk.jl_amd", "csrc", "fft_kernels.hip")).read()
    block = src[src.index("#define REG_SIZES(X)"):src.index("static bool fft_reg_on")]
    return [tuple(int(v) for v in m) for m in re.findall(r"X\((\d+), (\d+), (\d+), (\d+), (\d+)\)", block)]


def _dft(x, sgn):
    n = len(x)
    k = np.arange(n)
    return np.exp(sgn * 2j * np.pi * np.outer(k, k) / n) @ x


def _dft_ct(x, RA, RB, sgn):
    """dft_ct of fft_kernels.hip: input index n = RB a + b, result x[RB c + d] = X[c + RA d] (in place)."""
    x = x.copy()
    N = RA * RB
    for b in range(RB):
        t = _dft(np.array([x[RB * a + b] for a in range(RA)]), sgn)
        for c in range(RA):
            x[RB * c + b] = t[c] * np.exp(sgn * 2j * np.pi * (b * c) / N)
    for c in range(RA):
        u = _dft(np.array([x[RB * c + b] for b in range(RB)]), sgn)
        for d in range(RB):
            x[RB * c + d] = u[d]
    return x


@pytest.mark.parametrize("size", _reg_sizes())
def test_four_step_z_pass_index_maps(size):
    """NumPy model of k_zpass_reg's data flow (FourStep::backward, the V multiply in natural order, FourStep::forward):
    which thread holds which element after each sub-transform, the twiddles between them, the two transposition images.
    Every instantiated length against numpy.fft, with a pruned input (sphere planes only) as the kernel sees it."""
    n, R1A, R1B, R2A, R2B = size
    R1, R2 = R1A * R1B, R2A * R2B
    # (round 6: the small cubes of the k-point workloads added lengths with R1 < R2 and single-factor radices, R?B = 1)
    assert R1 * R2 == n and all(2 <= r <= 6 for r in (R1A, R2A)) and all(1 <= r <= 6 for r in (R1B, R2B))
    assert 8 * max(R1, R2) <= 1024                                  # threads of a tile
    rng = np.random.default_rng(n)
    z_lo, nzx = n // 4 + 1, 2 * (n // 4) + 1                        # planes {0..z_lo-1} u {n-(nzx-z_lo)..n-1}
    x = np.zeros(n, complex)
    idx = np.r_[0:z_lo, n - (nzx - z_lo):n]
    x[idx] = rng.standard_normal(nzx) + 1j * rng.standard_normal(nzx)
    V = rng.standard_normal(n)
    tw = np.exp(2j * np.pi * np.arange(n) / n)
    img1 = np.zeros((R1, R2), complex)                              # LDS image [k1][n2]
    for j in range(R2):                                             # thread n2 = j
        a = _dft_ct(np.array([x[R2 * n1 + j] for n1 in range(R1)]), R1A, R1B, +1)
        for p in range(R1):
            c, d = divmod(p, R1B)
            k1 = c + R1A * d
            img1[k1, j] = a[p] * tw[j * k1]
    img2 = np.zeros((R2, R1), complex)                              # LDS image [k1'][k1]
    mid = np.zeros(n, complex)
    for j in range(R1):                                             # thread k1 = j
        a2 = _dft_ct(img1[j, :].copy(), R2A, R2B, +1)
        f = np.zeros(R2, complex)
        for p in range(R2):
            c, d = divmod(p, R2B)
            k2 = c + R2A * d
            mid[j + R1 * k2] = a2[p]
            f[k2] = a2[p] * V[j + R1 * k2]
        f = _dft_ct(f, R2A, R2B, -1)
        for p in range(R2):
            c, d = divmod(p, R2B)
            k1p = c + R2A * d
            img2[k1p, j] = f[p] * np.conj(tw[j * k1p])
    out = np.zeros(n, complex)
    for j in range(R2):                                             # thread k1' = j
        a = _dft_ct(img2[j, :].copy(), R1A, R1B, -1)
        for p in range(R1):
            c, d = divmod(p, R1B)
            out[j + R2 * (c + R1A * d)] = a[p]
    ref_mid = np.fft.ifft(x) * n
    ref = np.fft.fft(ref_mid * V)
    assert np.abs(mid - ref_mid).max() < 1e-11 * np.abs(ref_mid).max()
    assert np.abs(out - ref).max() < 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("m,k,flags", [(259, 135491, 1), (503, 132430, 9), (1006, 132430, 9), (1509, 132430, 9), (777, 67746, 1),
                                       (141, 4100, 1), (1300, 300, 1), (2000, 264859, 9)])
def test_zgemm_compact_upper_grid_enumerates_the_live_tiles_once(lib, m, k, flags):
    """Emulation of the `upper & 16` decode of k_zgemm_3m (compact UPPER launches): the grid 8 ceil(L ns / 8) holds every
    (k-chunk, live tile) pair exactly once, no tile strictly below the diagonal, and each XCD (id % 8) a contiguous block
    of at most ceil(L ns / 8) work items."""
    p = _gemm_plan(lib, "C", m, m, k, flags)
    assert p["zmI"] == 2
    bn = p["bn"]
    q = 128 // bn
    shift_r = 1 if (m % 128 and m >= 128) else 0
    gm, gn, ns = p["gmf"] + shift_r, p["gnf"] + p["shift"], p["nsI"]
    if gm == 0 or gn == 0:
        pytest.skip("no interior launch")
    L = sum(max(0, gn - r * q) for r in range(gm))
    N = L * ns
    per = (N + 7) // 8
    seen = {}
    for wg in range(8 * per):
        xcd, slot = wg & 7, wg >> 3
        item = xcd * per + slot
        if slot >= per or item >= N:
            continue
        z, t = divmod(item, L)
        r = 0
        while t >= gn - r * q:
            t -= gn - r * q
            r += 1
        row_t, col_t = r, r * q + t
        assert 0 <= row_t < gm and 0 <= col_t < gn and 0 <= z < ns
        jend = m if (p["shift"] and col_t == gn - 1) else (col_t + 1) * bn
        assert row_t * 128 < jend                                   # the tile intersects the upper triangle
        assert (z, row_t, col_t) not in seen
        seen[(z, row_t, col_t)] = xcd
    live = [(r, c) for r in range(gm) for c in range(gn) if r * 128 < (m if (p["shift"] and c == gn - 1) else (c + 1) * bn)]
    assert len(live) == L and len(seen) == L * ns
    assert {(r, c) for (_, r, c) in seen} == set(live)


def test_sphere_planes_wrap_around_contiguously_for_the_baseline_kpoints():
    """The register-resident z kernels address the sphere's z planes by index arithmetic: they must be {0 .. z_lo-1} u
    {nz-(nzx-z_lo) .. nz-1} (dftk_mi_kblock's z_lo, api.cpp).  Checked here on the host for every irreducible k-point of
    the k-point BASELINE configs (cfg 3: fcc Al 36^3, cfg 4: graphene 30x30x120) and for a Gamma-only supercell."""
    import ctypes as C
    from dftk_jl_amd._lib import check
    lib = dftk.load_library()

    def planes_ok(b):
        nx, ny, nz = b.fft_size
        for kp in b.kpoints:
            z = np.unique(np.asarray(kp.mapping) // (nx * ny))
            lo = 0
            while lo < len(z) and z[lo] == lo:
                lo += 1
            assert lo >= 1 and np.array_equal(z[lo:], nz - (len(z) - lo) + np.arange(len(z) - lo)), (kp.coordinate, z)

    a = 7.6324708938577865                                           # test/testcases.jl:74, as bench.py --system al
    lat = a / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
    Al = dftk.ElementPsp("Al", dftk.load_psp("Al", "pbe"))
    model = dftk.model_DFT(lat, [Al], [np.zeros(3)], functionals=("gga_x_pbe", "gga_c_pbe"), temperature=1e-3,
                           smearing="gaussian")
    planes_ok(dftk.PlaneWaveBasis(model, 40.0, dftk.MonkhorstPack((6, 6, 6)), device="cpu", build_terms=False))
    a, Lz = 4.66, 20.0                                               # examples/graphene.jl:15-30, as bench.py --system graphene
    lat = np.array([[a / 2, a / 2, 0.0], [-a * np.sqrt(3) / 2, a * np.sqrt(3) / 2, 0.0], [0.0, 0.0, Lz]])
    Cc = dftk.ElementPsp("C", dftk.load_psp("C", "pbe"))
    pos = [np.array([1 / 3, -1 / 3, 0.0]), np.array([-1 / 3, 1 / 3, 0.0])]
    model = dftk.model_DFT(lat, [Cc, Cc], pos, functionals=("gga_x_pbe", "gga_c_pbe"), temperature=1e-3,
                           smearing="fermi_dirac")
    b = dftk.PlaneWaveBasis(model, 40.0, dftk.MonkhorstPack((5, 5, 1)), device="cpu", build_terms=False)
    assert b.fft_size[2] == 120                                      # the length the multi-k z kernels run at cfg 4
    planes_ok(b)
    lat, atoms, pos = dftk.silicon_cell((2, 2, 2))
    planes_ok(dftk.PlaneWaveBasis(dftk.model_DFT(lat, atoms, pos), 30.0, dftk.ExplicitKpoints([[0, 0, 0]], [1.0]), device="cpu",
                                  build_terms=False))


def test_header_is_plain_c_and_the_c_consumer_builds_and_fails_loudly_without_a_gpu():
    """include/dftk_mi355x.h compiled as C99 (-pedantic -Werror) by gcc through tools/abi_c_check.c -- the C consumer a
    Julia `ccall` stands for (struct-by-value `dftk_mi_cplx`, `char id[128]`, callback typedefs).  Linked against the
    in-tree library; on a box without a GPU it must refuse to do anything (exit code 3, no CPU fallback)."""
    import subprocess
    from dftk_jl_amd import _build
    exe = _build.build_abi_check()
    assert os.path.exists(exe)
    import torch
    if not torch.cuda.is_available():
        res = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=120)
        assert res.returncode == 3, (res.returncode, res.stdout, res.stderr)
        assert "no CPU fallback" in res.stderr


def test_band_count_rules_fft_size_helper_and_kpoint_normalisation():
    """``default_n_bands`` / ``AdaptiveBands`` / ``FixedBands`` (src/scf/nbands_algorithm.jl:7-110) at the BASELINE
    configs (M = 7 / 259 / 6 / 8 / 503) and ``determine_n_bands`` on hand-made occupations;
    ``next_compatible_fft_size`` (src/fft.jl:277-287); ``normalize_kpoint_coordinate`` (src/bzmesh.jl:5-10: ties go UP,
    so 1/2 maps to -1/2)."""
    from dftk_jl_amd.basis import next_compatible_fft_size
    from dftk_jl_amd.scf import FixedBands, default_n_bands
    from dftk_jl_amd.symmetry import normalize_kpoint_coordinate
    lat, atoms, pos = dftk.silicon_cell()
    si = dftk.model_DFT(lat, atoms, pos)
    assert (default_n_bands(si), dftk.AdaptiveBands(si).n_bands_converge, dftk.AdaptiveBands(si).n_bands_compute) == (4, 4, 7)
    for ncell, M in ((4, 259), (5, 503)):
        big = dftk.model_DFT(*dftk.silicon_cell((ncell,) * 3))
        ab = dftk.AdaptiveBands(big)
        assert (ab.n_bands_converge, ab.n_bands_compute) == (M - 3, M)
    a = 7.6324708938577865
    al = dftk.model_DFT(a / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]]), [dftk.ElementPsp("Al", dftk.load_psp("Al", "pbe"))],
                        [np.zeros(3)], functionals=("gga_x_pbe", "gga_c_pbe"), temperature=1e-3, smearing="gaussian")
    ab = dftk.AdaptiveBands(al)                       # 3 electrons, T > 0: ceil(2 * 1.05) = 3 to converge, max(3 + 3, ceil(2.4)) = 6
    assert (default_n_bands(al, 1.0), ab.n_bands_converge, ab.n_bands_compute) == (2, 3, 6)
    fb = FixedBands.for_model(al)
    assert (fb.n_bands_converge, fb.n_bands_compute) == (3, 6)          # ceil(2 * 1.20) = 3, + 3
    assert FixedBands(8).n_bands_compute == 11
    # first step (no occupation yet): converge floor((3 + 6) / 2) = 4, compute 6 (or as many as the guess holds)
    assert ab.determine_n_bands(None, None, None) == (4, 6)
    assert ab.determine_n_bands(None, None, [np.zeros((9, 5))]) == (4, 9)
    # later steps: converge up to the last band occupied above the threshold; compute up to the last band within gap_min of it
    occ = [np.array([2.0, 2.0, 1.0, 1e-3, 1e-9, 0.0]), np.array([2.0, 1.5, 1e-7, 0.0, 0.0, 0.0])]
    eig = [np.array([0.0, 0.1, 0.2, 0.3, 0.305, 0.5]), np.array([0.0, 0.1, 0.2, 0.3, 0.4, 0.5])]
    assert ab.determine_n_bands(occ, eig, None) == (4, 7)               # band 4 occupied (1e-3); eps_5 - eps_4 < gap_min -> 5, but >= 4 + 3
    occ_full = [np.full(6, 2.0)]                                        # everything occupied: "one more than we have"
    assert ab.determine_n_bands(occ_full, [np.arange(6.0)], None) == (6, 9)
    assert ab.determine_n_bands(occ, None, [np.zeros((12, 5))])[1] == 12
    # FFT sizes: 2-3-5-smooth and divisible by the product of the factors
    assert [next_compatible_fft_size(n) for n in (27, 28, 31, 97, 150, 191)] == [27, 30, 32, 100, 150, 192]
    assert next_compatible_fft_size(28, factors=(2,)) == 30 and next_compatible_fft_size(25, factors=(2, 3)) == 30
    assert next_compatible_fft_size(27, factors=(4,)) == 32 and next_compatible_fft_size(31, smallprimes=()) == 31
    # k-point coordinates in [-1/2, 1/2)
    got = normalize_kpoint_coordinate([0.5, -0.5, 1.25, -0.75, 0.49999, 2.0])
    np.testing.assert_allclose(got, [-0.5, -0.5, 0.25, 0.25, 0.49999, 0.0], atol=1e-15)


def test_interpolate_kpoint_is_the_references_scatter_gather_through_the_cube():
    """``interpolate_kpoint`` (src/interpolation.jl:96-115): the reference scatters the coefficients into a cube-sized
    array by ``kpoint_in.mapping`` and gathers them by ``kpoint_out.mapping``; the mirror matches the two ascending
    mapping lists directly (no cube-sized temporary per band).  Same numbers, plane waves outside the input sphere
    get zero, identical k-points return a copy; the final ``ortho_qr`` of the reference is LOBPCG's own first step."""
    import torch
    from dftk_jl_amd.eigen import interpolate_kpoint
    lat, atoms, pos = dftk.silicon_cell()
    model = dftk.model_DFT(lat, atoms, pos)
    basis = dftk.PlaneWaveBasis(model, 7.0, dftk.MonkhorstPack((3, 3, 3)), device="cpu", build_terms=False)
    assert len(basis.kpoints) >= 3
    gen = torch.Generator().manual_seed(0)
    for k_in, k_out in [(basis.kpoints[0], basis.kpoints[1]), (basis.kpoints[2], basis.kpoints[0]),
                        (basis.kpoints[1], basis.kpoints[2])]:
        assert np.all(np.diff(k_in.mapping) > 0) and np.all(np.diff(k_out.mapping) > 0)     # the premise: ascending
        data = torch.complex(torch.randn((5, k_in.n_G), dtype=torch.float64, generator=gen),
                             torch.randn((5, k_in.n_G), dtype=torch.float64, generator=gen))
        got = interpolate_kpoint(data, k_in, k_out).numpy()
        cube = np.zeros((5, basis.N), dtype=complex)
        cube[:, k_in.mapping] = data.numpy()
        want = cube[:, k_out.mapping]
        assert got.shape == (5, k_out.n_G) and np.array_equal(got, want)
        shared = np.isin(k_out.mapping, k_in.mapping)
        assert 0 < shared.sum() < k_out.n_G and np.all(got[:, ~shared] == 0)
    same = interpolate_kpoint(data, k_in, k_in)
    assert torch.equal(same, data) and same.data_ptr() != data.data_ptr()


@pytest.mark.parametrize("n,real", [(33, False), (100, True), (259, False), (503, True), (512, False)])
def test_block_recurrences_of_the_cooperative_cholesky_and_inverse(n, real):
    """NumPy emulation of the block algebra of ``k_potrf_trtri_coop`` (dense_kernels.hip; safe_cholesky + inv(R),
    lobpcg_hyper_impl.jl:190-210), block column by block column exactly as the workgroups walk it: block column l of
    the lower factor = (its columns of O minus the contributions of the panels to its left) solved against its own
    16 x 16 diagonal factor; W_l = D_l^{-1}; block column l of X = L^{-1} from X_ll = W_l,
    X_kl = -W_k sum_{j=l}^{k-1} L_kj X_jl; the caller gets R = L^H and Z = X^H (upper triangular, zeros below).
    Padding rows / columns up to the next multiple of 16 carry the identity."""
    rng = np.random.default_rng(n)
    A = rng.standard_normal((3 * n, n)) + (0 if real else 1j) * rng.standard_normal((3 * n, n))
    O = A.conj().T @ A
    B = 16
    nb = -(-n // B)
    npad = B * nb
    Lp = np.eye(npad, dtype=O.dtype)
    Lp[:n, :n] = np.tril(O)                              # what the kernel loads: conj of the caller's upper triangle
    panels, W = [], []
    for l in range(nb):
        r0 = B * l
        col = Lp[r0:, r0:r0 + B].copy()                  # rows r0 .. npad of block column l
        for j in range(l):                               # consume the published panels to the left
            P = panels[j][r0 - B * j:, :]                # their rows from r0 on; the first 16 are L_{l,j}
            col -= P @ P[:B].conj().T
        D = np.linalg.cholesky(col[:B])                  # lower factor of the diagonal block
        col[B:] = col[B:] @ np.linalg.inv(D).conj().T    # y D^H = x, row by row
        col[:B] = D
        panels.append(col)
        W.append(np.linalg.inv(D))
    L = np.zeros((npad, npad), dtype=O.dtype)
    for l in range(nb):
        L[B * l:, B * l:B * (l + 1)] = panels[l]
    assert np.allclose(np.triu(L, 1), 0)
    X = np.zeros_like(L)
    for l in range(nb):                                  # block column l of X = L^{-1}
        X[B * l:B * (l + 1), B * l:B * (l + 1)] = W[l]
        for k in range(l + 1, nb):
            S = sum(L[B * k:B * (k + 1), B * j:B * (j + 1)] @ X[B * j:B * (j + 1), B * l:B * (l + 1)] for j in range(l, k))
            X[B * k:B * (k + 1), B * l:B * (l + 1)] = -W[k] @ S
    R = L.conj().T[:n, :n]
    Z = X.conj().T[:n, :n]
    scale = np.linalg.norm(O)
    assert np.linalg.norm(R.conj().T @ R - O) < 1e-13 * scale
    assert np.linalg.norm(Z @ R - np.eye(n)) < 1e-11 * np.linalg.cond(R)
    assert np.allclose(np.tril(Z, -1), 0) and np.allclose(np.tril(R, -1), 0)
    np.testing.assert_allclose(R, np.linalg.cholesky(O).conj().T, rtol=0, atol=1e-11 * np.sqrt(scale))
    if real:
        assert not np.iscomplexobj(R)


def test_host_side_orthogonalisation_of_small_ritz_coefficient_blocks():
    """``host_ortho_small`` of csrc/lobpcg.cpp -- ortho!(X, Y) (lobpcg_hyper_impl.jl:271-323) on the HOST for the Ritz
    coefficient blocks ``cP = cX - e`` of small k-blocks (four scheduling rounds fewer per iteration of the lock-step
    driver) -- through tools/host_ortho_check.cpp, which includes the driver's translation unit and runs without a
    GPU: X'X = I, Y'X = 0 and span((1 - Y Y') X_in) kept to round-off for complex / real blocks shaped like the
    driver's; without a generator a column inside span(Y) (``drop_small!``) and linearly dependent columns (Cholesky
    breakdown) are DECLINED with X untouched, i.e. handed to the device path and its fallbacks; with the driver's generator
    ``drop_small!`` runs on the host too (the reference's ``cP`` holds plain columns of ``cX`` whenever a vector was newly
    locked: complex and real block)."""
    import subprocess
    from dftk_jl_amd import _build
    exe = _build.build_host_ortho_check()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, (res.returncode, res.stdout[-3000:], res.stderr[-2000:])
    assert "host_ortho_check OK" in res.stdout and "FAILED" not in res.stdout
    assert res.stdout.count("done=1") == 9 and res.stdout.count("done=0") == 2


def test_memory_statistics_and_plan_for_the_literal_4096_electron_cell():
    """``estimate_memory_usage`` (src/memory_usage.jl:35-87: psi_k, P_k, rho bytes and the 1 P + 2 psi + 6 psi_k + 12 rho
    peak) and the per-GPU plan of the plane-wave-sharded Gamma block: the headline 1000-electron cell fits one
    MI355X several times over; the LITERAL BASELINE configs[4] string (Si 8x8x8, 1024 atoms, 4096 electrons) does not
    fit one GPU in any layout but fits 4 and 8 GPUs with the existing row-slab sharding and DENSE projector slabs
    (no on-the-fly projectors needed) -- the question SURVEY appendix B left open."""
    from dftk_jl_amd import memory_usage as mu
    lat, atoms, pos = dftk.silicon_cell((5, 5, 5))
    model = dftk.model_DFT(lat, atoms, pos)
    st = dftk.estimate_memory_usage(model, 30.0)
    assert (st.n_kpoints, st.n_Gk, st.n_bands, st.n_nonlocal_projectors) == (1, 264859, 503, 1250)
    assert st.psik_bytes == 16 * 264859 * 503 and st.nonlocal_Pk_bytes == 16 * 264859 * 1250
    assert st.rho_bytes == 8 * 192 ** 3
    assert st.scf_peak_bytes == st.nonlocal_P_bytes + 2 * st.psi_bytes + 6 * st.psik_bytes + 12 * st.rho_bytes
    one = dftk.plan_planewave_sharded(model, 30.0, 1)
    assert one["fits"] and one["fft_size"] == (192, 192, 192) and one["n_half"] == 132430
    assert one["bytes_per_rank"]["lobpcg_blocks"] == 14 * 132430 * 503 * 16          # 14 blocks
    assert 25e9 < one["total_bytes_per_rank"] < 45e9
    lat, atoms, pos = dftk.silicon_cell((8, 8, 8))
    big = dftk.model_DFT(lat, atoms, pos)
    assert big.n_electrons == 4096 and len(pos) == 1024
    plans = {p: dftk.plan_planewave_sharded(big, 30.0, p) for p in (1, 2, 4, 8)}
    assert plans[1]["fft_size"] == (300, 300, 300) and plans[1]["n_bands"] == 2051 and plans[1]["n_projectors"] == 5120
    assert not plans[1]["fits"] and plans[1]["reference_layout"]["scf_peak_bytes"] > 288e9
    assert plans[4]["fits"] and plans[8]["fits"]
    assert plans[8]["total_bytes_per_rank"] < 100e9
    assert plans[8]["communication"]["gram_allreduce_bytes"] == 16 * (3 * 2051) ** 2
    assert not plans[8]["limits"]["register_resident_z_kernels"] and plans[8]["limits"]["fft_axis_lds_ok"]
    assert "DOES NOT FIT" in mu.format_plan(plans[1]) and "fits" in mu.format_plan(plans[8])


def test_collinear_spin_descriptors_on_host():
    """Model / PlaneWaveBasis with collinear spin (Model.jl:29-39, :352-372; PlaneWaveBasis.jl:50-53, :218-232;
    Kpoint.jl:58-74): spin inferred from the magnetic moments, filled occupation 1, k-point list doubled (all up, then
    all down) with repeated weights summing to 2, band counts and Fermi level / occupations equal to the oracle's."""
    import oracle
    lat = 2.71176 * np.array([[-1, 1, 1], [1, -1, 1], [1, 1, -1.0]])
    Fe = dftk.ElementPsp("Fe", dftk.load_psp("Fe", "lda"))
    m = dftk.model_DFT(lat, [Fe], [np.zeros(3)], functionals=("lda_xc_teter93",), temperature=0.01, smearing="fermi_dirac",
                       magnetic_moments=(4.0,), symmetries=True)
    assert (m.spin_polarization, m.n_spin_components, m.filled_occupation, m.n_electrons) == ("collinear", 2, 1, 8)
    assert len(m.symmetries) == 48
    b = dftk.PlaneWaveBasis(m, 15, dftk.MonkhorstPack((4, 4, 4), (0.5, 0.5, 0.5)), fft_size=(20, 20, 20), device="cpu",
                            build_terms=False)
    assert len(b.kpoints) == 12 and [k.spin for k in b.kpoints] == [1] * 6 + [2] * 6
    assert abs(sum(b.kweights) - 2.0) < 1e-14 and b.kweights[:6] == b.kweights[6:]
    assert all(np.array_equal(b.kpoints[i].mapping, b.kpoints[i + 6].mapping) for i in range(6))
    assert dftk.AdaptiveBands(m).n_bands_compute == 8
    om = oracle.model_DFT(lat, [oracle.ElementPsp("Fe", oracle.load_psp_hgh("Fe", "lda"))], [np.zeros(3)],
                          functionals=("lda_xc_teter93",), temperature=0.01, smearing="fermi_dirac", magnetic_moments=(4.0,),
                          symmetries=True)
    ob = oracle.PlaneWaveBasis(om, 15, oracle.MonkhorstPack((4, 4, 4), (0.5, 0.5, 0.5)), fft_size=(20, 20, 20), build_terms=False)
    assert all(np.allclose(k.coordinate, q.coordinate) and k.spin == q.spin for k, q in zip(b.kpoints, ob.kpoints))
    rng = np.random.default_rng(1)
    ev = [np.sort(rng.standard_normal(8)) * 0.3 + (0.05 if k.spin == 2 else 0.0) for k in b.kpoints]
    occ, eF = dftk.compute_occupation(b, ev)
    oocc, oeF = oracle.compute_occupation(ob, ev)
    assert eF == pytest.approx(oeF, abs=1e-12) and all(np.allclose(a, c, atol=1e-12) for a, c in zip(occ, oocc))
    assert max(float(o.max()) for o in occ) <= 1.0 + 1e-12
    # a model without moments stays unpolarised; moments on an explicitly unpolarised model are refused by the guess
    m0 = dftk.model_DFT(lat, [Fe], [np.zeros(3)], functionals=("lda_xc_teter93",), temperature=0.01)
    assert (m0.spin_polarization, m0.n_spin_components, m0.filled_occupation) == ("none", 1, 2)
    with pytest.raises(NotImplementedError):
        dftk.model_DFT(lat, [Fe], [np.zeros(3)], spin_polarization="full")


def test_partial_eigensolver_shift_rule_and_numpy_model(lib):
    """dftk_mi_heev_lowest (csrc/eig_kernels.hip) on the host: the shift rule exported by the library equals the NumPy model's
    (tools/lab/heev_lowest_model.py), it always lies above the nev-th smallest diagonal entry (the interlacing argument), the
    number of held iterations follows the estimated gap, and the model -- the same schedule of scaled Newton-Schulz iterations,
    leverage-score column selection and Cholesky-QR passes as the kernels -- returns the lowest nev pairs of matrices with the
    structure of LOBPCG Rayleigh-Ritz matrices to round-off."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "lab"))
    import heev_lowest_model as model
    rng = np.random.default_rng(12)
    for n, nev in ((90, 30), (333, 111), (1006, 503)):
        d = np.concatenate([np.sort(rng.uniform(-0.2, 0.4, nev)), rng.uniform(0.45, 9.0, n - nev)])
        d = d[rng.permutation(n)]
        sigma, gap, hold = C.c_double(), C.c_double(), C.c_int()
        check(lib.dftk_mi_heev_sigma_host(n, d.ctypes.data, nev, C.byref(sigma), C.byref(gap), C.byref(hold), 12.0))
        s_model, g_model = model.choose_sigma(d, nev)
        assert abs(sigma.value - s_model) < 1e-15 and abs(gap.value - g_model) < 1e-15
        ds = np.sort(d)
        assert ds[nev - 1] < sigma.value < ds[nev]                 # above d_(nev); inside the diagonal gap when there is one
        want = int(np.ceil(np.log(max(model.L_HAT / (gap.value / 12.0), 1.0)) / np.log(model.GROW)))
        assert hold.value == min(want, 34)
    assert lib.dftk_mi_heev_sigma_host(5, None, 2, None, None, None, 1.0) < 0
    # no gap in the diagonal at all (equal entries): still strictly above d_(nev)
    d = np.zeros(40)
    check(lib.dftk_mi_heev_sigma_host(40, d.ctypes.data, 10, C.byref(sigma), C.byref(gap), None, 0.0))
    assert sigma.value > 0.0 and gap.value > 0.0

    def rr_like(n, nev, coupling):
        nlow = nev + nev // 3
        low = np.repeat(rng.uniform(-0.2, 0.6, nlow // 4 + 1), 4)[:nlow] + 1e-6 * rng.standard_normal(nlow)
        lam = np.concatenate([np.sort(low), rng.uniform(0.6, 8.0, n - nlow)])
        Q, _ = np.linalg.qr(np.eye(n) + coupling * rng.standard_normal((n, n)) / np.sqrt(n))
        A = (Q.T * lam) @ Q
        w, Z = np.linalg.eigh(A[:nev, :nev])
        T = np.eye(n)
        T[:nev, :nev] = Z
        A = T.T @ A @ T
        A = (A + A.T) / 2
        A[:nev, :nev] = np.diag(w)
        return A
    for n, nev, coupling in ((240, 80, 0.3), (300, 150, 1e-3)):
        A = rr_like(n, nev, coupling)
        log = []
        out = model.heev_lowest(A, nev, log)
        assert out is not None, log
        lam, V = out
        ref = np.linalg.eigvalsh(A)[:nev]
        assert np.abs(lam - ref).max() < 1e-12 and np.abs(V.T @ V - np.eye(nev)).max() < 1e-12
        assert np.abs(A @ V - V * lam).max() < 1e-11
        k = [e for e in log if e[0] == "summary"][0][2]
        assert nev <= k <= max(nev + 64, (3 * nev) // 2)


def test_first_wave_of_a_batched_k_mesh_and_nearest_neighbour_start():
    """The two-wave start of a batched k-mesh (eigen.py; the reference starts k-point ik from k-point ik - 1, diag.jl:39-42):
    the first wave is spread evenly over the list, every other k-point takes its start from the nearest first-wave k-point up
    to reciprocal lattice vectors; the default first-wave size is the whole mesh up to 24 k-points, else n_k / 16 (>= 4)."""
    from dftk_jl_amd.eigen import first_wave_indices, nearest_index, _chain_width
    assert first_wave_indices(72, 4, True) == [0, 18, 36, 54]
    assert first_wave_indices(72, 16, False) == list(range(16))
    assert first_wave_indices(8, 16, True) == list(range(8)) and first_wave_indices(5, 5, True) == list(range(5))
    for n_k, w in [(72, 4), (100, 6), (33, 7), (200, 12)]:
        idx = first_wave_indices(n_k, w, True)
        assert len(idx) == w and idx[0] == 0 and idx == sorted(set(idx)) and idx[-1] < n_k
        gaps = np.diff(idx + [n_k])
        assert gaps.max() - gaps.min() <= 1                                   # evenly spread
    coords = np.array([[0.0, 0.0, 0.0], [0.5, 0.0, 0.0], [0.0, 0.5, 0.25]])
    assert nearest_index(coords, [0.1, 0.0, 0.0]) == 0 and nearest_index(coords, [0.45, 0.05, 0.0]) == 1
    assert nearest_index(coords, [0.95, 0.0, 0.0]) == 0                      # periodic image of Gamma
    assert nearest_index(coords, [0.0, -0.45, 0.2]) == 2

    class _B:
        kbatch, n_lanes = True, 1

        def __init__(self, n):
            self.kpoints = [None] * n
    assert _chain_width(_B(8)) == 8 and _chain_width(_B(24)) == 24 and _chain_width(_B(72)) == 4 and _chain_width(_B(200)) == 12
