"""N > 1 code path on the real hot path with ONE GPU: two ranks share cuda:0 and reduce over gloo
(RCCL refuses two ranks on one device; the collective is the only thing swapped).  Covers what the
driver's multi-GPU bench exercises: k-point distribution, the density all-reduce, the Fermi level
over sharded eigenvalues, summed energies, and bench.py's aggregation over ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import dftk_jl_amd as dftk  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KCOORDS = [[0, 0, 0], [0.5, 0, 0], [0.25, 0.25, 0]]
KWEIGHTS = [0.25, 0.5, 0.25]

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch, torch.distributed as dist
import dftk_jl_amd as dftk
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["PORT"],
                        rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
comm = dftk.KptComm.from_torch()
lat, atoms, pos = dftk.silicon_cell()
model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_vwn"))
kg = dftk.ExplicitKpoints(json.loads(os.environ["KCOORDS"]), json.loads(os.environ["KWEIGHTS"]))
basis = dftk.PlaneWaveBasis(model, 8, kg, fft_size=(20, 20, 20), device="cuda:0", comm_kpts=comm)
assert len(basis.kpoints) == (2 if comm.rank == 0 else 1)          # split_evenly: 3 k-points over 2 ranks
res = dftk.self_consistent_field(basis, tol=1e-9, nbandsalg=dftk.AdaptiveBands(model, n_bands_converge=6))
lam = comm.gather_lists([l[:6].tolist() for l in res["eigenvalues"]])
if comm.rank == 0:
    print("RESULT " + json.dumps({"E": res["energies"].total, "terms": dict(res["energies"]),
                                  "lam": [l for sub in lam for l in sub], "converged": bool(res["converged"]),
                                  "rho_sum": float(res["rho"].sum()) * basis.dvol}))
dist.barrier(); dist.destroy_process_group()
'''


def _spawn(cmds_env):
    procs = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT)
             for cmd, env in cmds_env]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-4000:]}"
    return outs


def test_two_ranks_one_gpu_scf_equals_single_rank(tmp_path):
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(31000 + os.getpid() % 2000)
    base = dict(os.environ, WORLD_SIZE="2", PORT=port, REPO=ROOT, MASTER_ADDR="127.0.0.1",
                KCOORDS=json.dumps(KCOORDS), KWEIGHTS=json.dumps(KWEIGHTS))
    outs = _spawn([([sys.executable, str(script)], dict(base, RANK=str(r))) for r in range(2)])
    line = [ln for ln in outs[0].splitlines() if ln.startswith("RESULT ")][-1]
    got = json.loads(line[len("RESULT "):])
    assert got["converged"]
    # the same calculation on one rank
    lat, atoms, pos = dftk.silicon_cell()
    model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_vwn"))
    basis = dftk.PlaneWaveBasis(model, 8, dftk.ExplicitKpoints(KCOORDS, KWEIGHTS), fft_size=(20, 20, 20))
    ref = dftk.self_consistent_field(basis, tol=1e-9, nbandsalg=dftk.AdaptiveBands(model, n_bands_converge=6))
    assert ref["converged"]
    assert abs(got["E"] - ref["energies"].total) < 1e-8 * 2            # 1e-8 Ha/atom
    for name, v in ref["energies"].items():
        assert abs(got["terms"][name] - v) < 1e-7, name
    np.testing.assert_allclose(np.array(got["lam"]).reshape(3, 6), np.array([l[:6] for l in ref["eigenvalues"]]),
                               atol=1e-7)
    assert abs(got["rho_sum"] - 8.0) < 1e-9


def test_bench_two_ranks_one_gpu(tmp_path):
    """bench.py's N = 2 path (torch.distributed.run env contract, one k-point per rank, max-over-ranks
    timing, whole-job value) on a tiny cell; both ranks on cuda:0 over gloo."""
    port = str(33000 + os.getpid() % 2000)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--supercell", "1", "--ecut", "10", "--no-cpu-baseline"]
    base = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                DFTK_MI_BENCH_BACKEND="gloo", DFTK_MI_BENCH_DEVICE="0")
    outs = _spawn([(cmd, dict(base, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)])
    lines = [ln for ln in outs[0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, outs[0][-2000:]
    assert not [ln for ln in outs[1].splitlines() if ln.startswith("{")]        # only rank 0 prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == "kpt2"
    assert abs(out["value"] - 2 * 2 / (out["ms_per_step"] * 2 / 1e3)) < 1e-6 * out["value"]   # 2 k-blocks x steps / time
    assert out["roofline"]["achieved"] > 0 and np.isfinite(out["config"]["E_total"])


def test_rccl_c_abi_single_rank_allreduce():
    """dftk_mi_comm_* / dftk_mi_allreduce_sum_f64 (what a Julia shim calls instead of mpi_sum!): the run-time
    RCCL binding, communicator life cycle and an in-place fp64 sum on a one-rank communicator."""
    import ctypes as C
    from dftk_jl_amd._lib import check
    lib = dftk.load_library()
    uid = C.create_string_buffer(128)
    check(lib.dftk_mi_comm_get_unique_id(uid))
    comm = C.c_void_p()
    check(lib.dftk_mi_comm_init_rank(uid.raw, 1, 0, 0, C.byref(comm)))
    x = torch.arange(1 << 20, dtype=torch.float64, device="cuda") * 0.5
    want = x.clone()
    torch.cuda.synchronize()
    check(lib.dftk_mi_allreduce_sum_f64(comm, x.data_ptr(), x.numel(), None))
    torch.cuda.synchronize()
    assert torch.equal(x, want)
    assert lib.dftk_mi_allreduce_sum_f64(comm, None, 4, None) < 0          # invalid argument, reported not crashed
    check(lib.dftk_mi_comm_destroy(comm))
