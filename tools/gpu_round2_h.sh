R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02h
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_symmetry.py tests/test_gpu_scf.py tests/test_gpu_baseline_parity.py tests/test_gpu_multirank.py tests/test_gpu_mixing.py -q --tb=short --durations=8 -p no:cacheprovider 2>&1 \
  | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path' > $O/pytest.log
tail -40 $O/pytest.log
cat > /tmp/lanes.py <<'PY'
import sys, time, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
import dftk_jl_amd as dftk
def run(name, model, ecut, kg, lanes, **kw):
    b = dftk.PlaneWaveBasis(model, ecut, kg, n_lanes=lanes, **kw)
    dftk.self_consistent_field(b, tol=1e-6, maxiter=1)
    t = time.time(); r = dftk.self_consistent_field(b, tol=1e-6); dt = time.time() - t
    print(f"{name:28s} lanes={b.n_lanes} k={len(b.kpoints):3d} n_iter={r['n_iter']:2d} wall={dt:6.2f}s  {r['n_iter']/dt:6.2f} it/s  E={r['energies'].total:.10f}", flush=True)
lat, atoms, pos = dftk.silicon_cell()
for lanes in (1, 4, 8, 16):
    run("cfg1 Si 4x4x4 nosym", dftk.model_DFT(lat, atoms, pos), 15, dftk.MonkhorstPack((4, 4, 4)), lanes)
run("cfg1 Si 4x4x4 sym", dftk.model_DFT(lat, atoms, pos, symmetries=True), 15, dftk.MonkhorstPack((4, 4, 4)), 8)
a = 7.6324708938577865
latA = a / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
Al = dftk.ElementPsp("Al", dftk.load_psp("Al", "pbe"))
for lanes, sym, kg in ((1, False, 6), (8, False, 6), (16, False, 6), (8, True, 12), (16, True, 12)):
    m = dftk.model_DFT(latA, [Al], [np.zeros(3)], functionals=("gga_x_pbe", "gga_c_pbe"), temperature=1e-3, smearing="gaussian", symmetries=sym)
    run(f"cfg3 Al PBE {kg}^3 sym={sym}", m, 40, dftk.MonkhorstPack((kg,) * 3), lanes)
PY
timeout 900 python /tmp/lanes.py 2>&1 | grep -v amdgpu.ids | tee $O/lanes.log
