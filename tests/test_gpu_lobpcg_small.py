"""The small-block LOBPCG driver (lobpcg.cpp: lobpcg_run_small -- ONE host synchronisation per iteration; k_b_ortho: the
whole ortho!(X) / ortho!(X, Y) loops of lobpcg_hyper_impl.jl:216-323 in one kernel) against NumPy, the oracle's LOBPCG
trajectory and the general driver (``DFTK_MI_LOBPCG_SMALL=0`` in a subprocess: the switch is read once per process)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd._lib import check  # noqa: E402

import oracle  # noqa: E402
from oracle.lobpcg import LOBPCG, PreconditionerTPA  # noqa: E402
from test_gpu_kernels import Basis, dev, run_lobpcg  # noqa: E402
from test_gpu_lobpcg_blocks import _block, _tpa_setup  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EPS = float(np.finfo(float).eps)


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return dftk.load_library()


def _small_stats(lib):
    a, b = C.c_int64(), C.c_int64()
    check(lib.dftk_mi_lobpcg_small_stats(C.byref(a), C.byref(b)))
    return a.value, b.value


@pytest.mark.parametrize("n,m,ny,cond", [(1350, 6, 12, 1e2), (1350, 6, 6, 1e6), (4653, 8, 16, 1e3), (725, 7, 0, 1e4), (997, 1, 3, 1.0),
                                          (1350, 8, 0, 1e9)])
def test_fused_ortho_kernel(lib, n, m, ny, cond):
    """ortho!(X, Y): X comes back orthonormal, orthogonal to Y, spanning (1 - Y Y') span(X); ortho!(X) alone keeps the span.
    Badly conditioned blocks need several Cholesky-QR passes / safe_cholesky's shifts -- all inside the one kernel."""
    rng = np.random.default_rng(n + m + ny)
    bs = Basis(lib, 8, 8, 8)
    Y = np.linalg.qr(_block(rng, n, max(ny, 1)))[0][:, :ny]
    X = _block(rng, n, m) @ (np.linalg.qr(_block(rng, m, m))[0] * np.geomspace(1.0, 1.0 / cond, m)) @ np.linalg.qr(_block(rng, m, m))[0]
    if ny:
        X = X + 0.5 * Y @ _block(rng, ny, m)
    Xd = dev(X.T.copy())
    Yd = dev(Y.T.copy()) if ny else None
    res = np.zeros(4)
    check(lib.dftk_mi_ortho_small(bs.h, n, m, Xd.data_ptr(), n, ny, Yd.data_ptr() if ny else None, n, None, 2 * EPS,
                                  res.ctypes.data))
    assert res[0] == 0.0, res
    Q = Xd.cpu().numpy().T
    assert np.linalg.norm(Q.conj().T @ Q - np.eye(m)) < 50 * EPS * m
    ref = X - Y @ (Y.conj().T @ X) if ny else X
    if ny:
        assert np.linalg.norm(Y.conj().T @ Q) < 10 * EPS * np.sqrt(n)
    # same span: the component of the reference block outside span(Q) is round-off (relative to its smallest direction)
    out = ref - Q @ (Q.conj().T @ ref)
    assert np.linalg.norm(out) < (1e-13 * cond + 1e-12) * np.linalg.norm(ref), (np.linalg.norm(out) / np.linalg.norm(ref), res)
    print("rounds", res[1], "Cholesky count", res[2], "growth", res[3])


def test_fused_ortho_reports_the_branches_it_does_not_take(lib):
    """A column inside span(Y) (drop_small! would re-randomise it) -> status 1; a non-finite block -> status 2."""
    rng = np.random.default_rng(5)
    n, m, ny = 900, 4, 6
    bs = Basis(lib, 8, 8, 8)
    Y = np.linalg.qr(_block(rng, n, ny))[0]
    X = _block(rng, n, m)
    X[:, 2] = Y @ _block(rng, ny, 1)[:, 0]
    Xd, Yd = dev(X.T.copy()), dev(Y.T.copy())
    res = np.zeros(4)
    check(lib.dftk_mi_ortho_small(bs.h, n, m, Xd.data_ptr(), n, ny, Yd.data_ptr(), n, None, 2 * EPS, res.ctypes.data))
    assert res[0] == 1.0, res
    X[5, 1] = np.nan
    Xd = dev(X.T.copy())
    check(lib.dftk_mi_ortho_small(bs.h, n, m, Xd.data_ptr(), n, ny, Yd.data_ptr(), n, None, 2 * EPS, res.ctypes.data))
    assert res[0] == 2.0, res


@pytest.mark.parametrize("use_tpa,M,ncc", [(1, 8, 5), (0, 6, 6), (1, 3, 3)])
def test_small_driver_walks_the_oracle_trajectory(lib, use_tpa, M, ncc):
    """tests/test_gpu_lobpcg_blocks.py::test_lobpcg_residual_history_matches_oracle for a block the small-block driver
    takes (M <= 8): same locking pattern, residual norms per iteration, eigenvalues; no restart on the general driver."""
    _, H, bs, kb = _tpa_setup(lib, Ecut=12, fft=(24, 24, 24))
    rng = np.random.default_rng(21 + M)
    tol = 1e-7
    X0 = np.linalg.qr(_block(rng, H.n_G, M))[0]
    calls0, restarts0 = _small_stats(lib)
    lam, res, nit, conv, nmv, X = run_lobpcg(lib, kb, X0, tol, n_conv_check=ncc, use_tpa=use_tpa, maxiter=200)
    calls1, restarts1 = _small_stats(lib)
    assert calls1 == calls0 + 1 and restarts1 == restarts0, "the block must take the small-block driver without a restart"
    Mo, nio, nsvd = C.c_int(), C.c_int(), C.c_int()
    check(lib.dftk_mi_lobpcg_history(kb.h, C.byref(Mo), C.byref(nio), None, 0, C.byref(nsvd)))
    hist = np.zeros((nio.value + 1, Mo.value))
    check(lib.dftk_mi_lobpcg_history(kb.h, C.byref(Mo), C.byref(nio), hist.ctypes.data, hist.size, C.byref(nsvd)))
    hist = hist.T
    assert (Mo.value, nio.value) == (M, nit)
    np.testing.assert_array_equal(hist[:, -1], res)
    prec = PreconditionerTPA(H.kinetic) if use_tpa else None
    ores = LOBPCG(H.mul, X0, prec, tol, 200, miniter=1, n_conv_check=ncc)
    ohist = ores["residual_history"]
    assert conv == 1
    np.testing.assert_allclose(lam[:ncc], ores["λ"][:ncc], atol=1e-10)
    nit_o = ohist.shape[1] - 1
    ncmp = min(nit, nit_o, 8) + 1
    dev_rel = np.abs(hist[:, :ncmp] - ohist[:, :ncmp]) / np.maximum(ohist[:, :ncmp], 1e-300)
    print("max relative deviation of the residual norms per iteration:", np.array2string(dev_rel.max(axis=0), precision=2))
    np.testing.assert_array_equal(hist[:, :ncmp] == 0.0, ohist[:, :ncmp] == 0.0)
    assert dev_rel[:, :4].max() < 1e-9
    assert dev_rel.max() < 1e-4
    assert abs(nit - nit_o) <= 2 + nit_o // 10, (nit, nit_o)
    dense = np.linalg.eigvalsh(H.to_dense())[:ncc]
    np.testing.assert_allclose(lam[:ncc], dense, atol=1e-9)
    assert np.linalg.norm(X.conj().T @ X - np.eye(M)) < 1e-11
    # (the reported norm of a locked column is 0, as in the reference's history; the others are the true residual norms)
    true_res = np.linalg.norm(H.mul(X) - X * lam, axis=0)
    assert true_res[:ncc].max() < tol
    live = res > 0
    np.testing.assert_allclose(true_res[live], res[live], rtol=1e-6, atol=1e-12)


def test_small_driver_restarts_on_a_rank_deficient_guess(lib):
    """A start block with a duplicated column: safe_cholesky's shifts run inside the kernel; whatever it cannot finish
    (drop_small!) restarts the call on the general driver -- the caller sees the same converged eigenpairs either way."""
    _, H, bs, kb = _tpa_setup(lib)
    rng = np.random.default_rng(12)
    M = 6
    X0 = np.linalg.qr(_block(rng, H.n_G, M))[0]
    X0[:, 4] = X0[:, 2]
    X0[:, 5] = X0[:, 0] + 1e-9 * X0[:, 1]
    lam, res, nit, conv, nmv, X = run_lobpcg(lib, kb, X0, 1e-8, maxiter=200)
    assert conv == 1
    np.testing.assert_allclose(lam, np.linalg.eigvalsh(H.to_dense())[:M], atol=1e-8)
    assert np.linalg.norm(X.conj().T @ X - np.eye(M)) < 1e-10


SCRIPT = r'''
import json, os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch
import dftk_jl_amd as dftk
out = {}
lat, atoms, pos = dftk.silicon_cell()
model = dftk.model_DFT(lat, atoms, pos, temperature=1e-3, smearing="gaussian")
for kb_env in ("1", "0"):
    os.environ["DFTK_MI_KBATCH"] = kb_env
    basis = dftk.PlaneWaveBasis(model, 12, dftk.MonkhorstPack((3, 3, 3)), fft_size=(24, 24, 24))
    r = dftk.self_consistent_field(basis, tol=1e-9)
    out["E_kbatch" + kb_env] = r["energies"].total
    out["eig_kbatch" + kb_env] = [np.asarray(l)[:4].tolist() for l in r["eigenvalues"]]
    out["n_iter_kbatch" + kb_env] = int(r["n_iter"])
import ctypes as C
a, b = C.c_int64(), C.c_int64()
basis.lib.dftk_mi_lobpcg_small_stats(C.byref(a), C.byref(b))
out["small_calls"], out["small_restarts"] = a.value, b.value
print("RESULT " + json.dumps(out))
'''


def test_scf_small_driver_equals_general_driver(tmp_path):
    """A metallic SCF on a k-mesh (batched loop and stream lanes) with the small-block driver (default) and with
    DFTK_MI_LOBPCG_SMALL=0: energies to 1e-9 Ha, eigenvalues to 1e-7, and the small-block driver hardly ever restarts."""
    script = tmp_path / "scf.py"
    script.write_text(SCRIPT)
    outs = {}
    for tag, env_extra in (("small", {}), ("general", {"DFTK_MI_LOBPCG_SMALL": "0"})):
        env = dict(os.environ, REPO=ROOT, **env_extra)
        res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=900)
        assert res.returncode == 0, (tag, res.stdout[-1500:], res.stderr[-3000:])
        outs[tag] = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):])
    s, g = outs["small"], outs["general"]
    assert g["small_calls"] == 0 and s["small_calls"] > 50
    assert s["small_restarts"] <= 0.02 * s["small_calls"], s
    for k in ("kbatch1", "kbatch0"):
        assert abs(s["E_" + k] - g["E_" + k]) < 1e-9, (k, s["E_" + k], g["E_" + k])
        assert np.max(np.abs(np.array(s["eig_" + k]) - np.array(g["eig_" + k]))) < 1e-7
        assert abs(s["n_iter_" + k] - g["n_iter_" + k]) <= 2
    assert abs(s["E_kbatch1"] - s["E_kbatch0"]) < 1e-9
