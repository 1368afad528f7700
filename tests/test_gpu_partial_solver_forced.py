"""The partial-spectrum Rayleigh-Ritz solver (dftk_mi_heev_lowest: Newton-Schulz spectral split + Jacobi on the projected
matrix, eig_kernels.hip) takes over from n = 384 by default, i.e. only for the 1000-electron cells.  Here the eigen- /
LOBPCG / SCF parity suites are re-run in a subprocess with ``DFTK_MI_HEEV_PARTIAL_MIN=24``: every Rayleigh-Ritz step of
every un-batched LOBPCG call of those suites -- complex k-points, metals with smearing, PBE, the residual-history
comparisons with the oracle, LOBPCG driven to 1e-12 -- then goes through the split, against the same oracle numbers and
reference pins (test/silicon_lda.jl:47-51, test/lobpcg.jl:52-76).  (Round 5 covered this only by an offline log; the
switch is read once per process, hence the subprocess.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["tests/test_gpu_eig.py", "tests/test_gpu_lobpcg_blocks.py", "tests/test_gpu_scf.py", "tests/test_gpu_gamma_real.py",
          "tests/test_gpu_kernels.py::test_lobpcg_free_electron_golden",
          "tests/test_gpu_kernels.py::test_lobpcg_core_hamiltonian_vs_oracle_and_dense"]


def test_parity_suites_pass_with_the_partial_solver_forced_from_n24():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    env = dict(os.environ, DFTK_MI_HEEV_PARTIAL_MIN="24", DFTK_MI_HEEV_TRACE="1")
    res = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-s"] + SUITES,
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = res.stdout[-3000:] + res.stderr[-3000:]
    assert res.returncode == 0, tail
    # the split really ran (its trace line), and not only on the direct dftk_mi_heev_lowest calls of test_gpu_eig.py
    assert (res.stdout + res.stderr).count("[heev lowest") > 50, tail
