"""Where a LATE SCF step of the benchmark cell spends its time: family timers of the library over steps 4..8 only
(the first two steps diagonalise 59 + 15 LOBPCG iterations and dominate the whole-SCF profile).
python tools/late_step_profile.py [supercell = 5] [steps skipped = 8] [steps kept = 6] [--any]
With --any every step is kept whatever its LOBPCG iteration count: `late_step_profile.py 5 0 1 --any` is the profile of the
FIRST SCF step (15 LOBPCG iterations from random orbitals, 27 % of the driver's window)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd._lib import check  # noqa: E402

keep_any = "--any" in sys.argv
sys.argv = [a for a in sys.argv if a != "--any"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lib = dftk.load_library()
lat, atoms, pos = dftk.silicon_cell((n, n, n))
model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
basis = dftk.PlaneWaveBasis(model, 30.0, dftk.MonkhorstPack((1, 1, 1)))
st = dftk.ScfStepper(basis, tol=1e-6, phase_timers=True)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for _ in range(skip):
    st.step()
torch.cuda.synchronize()
# Only steps with ONE LOBPCG iteration enter the table (the typical late step; a step that needs 2-3 iterations would blur it):
# the profiler is switched on per step and a step is kept or thrown away as a whole.  With DFTK_MI_GEMM_SHAPES=1 the library
# prints the zgemm shape table of a step to stderr when the profiler is switched off: captured here and summed over the kept
# steps.
import re
import tempfile


def prof_off_capture():
    """dftk_mi_prof_enable(0) with fd 2 redirected to a file: returns the [zgemm-shape] lines it printed."""
    sys.stderr.flush()
    with tempfile.TemporaryFile(mode="w+b") as tmp:
        saved = os.dup(2)
        os.dup2(tmp.fileno(), 2)
        try:
            check(lib.dftk_mi_prof_enable(basis.handle, 0))
        finally:
            os.dup2(saved, 2)
            os.close(saved)
        tmp.seek(0)
        return [ln for ln in tmp.read().decode(errors="replace").splitlines() if ln.startswith("[zgemm-shape]")]


shape_tab = {}
names = {0: "zgemm", 11: "zgemm_struct", 1: "fftA", 2: "fftB", 3: "fftC", 4: "fftD", 5: "fftE", 6: "dens", 7: "heev", 8: "chol", 9: "applyH(total)",
         15: "elementwise", 16: "host waits"}
nst = int(sys.argv[3]) if len(sys.argv) > 3 else 6
fam_ms = {f: 0.0 for f in names}
fam_nl = {f: 0 for f in names}
timers = {}
wall = 0.0
kept, tried, iters_seen = 0, 0, []
while kept < nst and tried < 4 * nst:
    tried += 1
    check(lib.dftk_mi_prof_enable(basis.handle, 1))
    t0 = time.time()
    info = st.step()
    torch.cuda.synchronize()
    dt = time.time() - t0
    n_it = int(round(float(np.mean(info["diagonalization"]["n_iter"]))))
    iters_seen.append(n_it)
    if n_it != 1 and not keep_any:
        prof_off_capture()      # discard this step's records
        continue
    vals = {}
    for f in names:
        ms, work, nl = C.c_double(), C.c_double(), C.c_int64()
        check(lib.dftk_mi_prof_get(basis.handle, f, C.byref(ms), C.byref(work), C.byref(nl)))
        vals[f] = (ms.value, nl.value)
    for ln in prof_off_capture():
        m = re.match(r"\[zgemm-shape\] (\S) m=(\d+) n=(\d+) k=(\d+) flags=(\d+) calls=(\d+) ms=([0-9.]+) TF/s=([0-9.]+)", ln)
        if m:
            key = (m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)))
            e = shape_tab.setdefault(key, [0, 0.0, 0.0])
            e[0] += int(m.group(6))
            e[1] += float(m.group(7))
            e[2] += float(m.group(8)) * float(m.group(7))      # TF/s x ms = Gflop
    for f in names:
        fam_ms[f] += vals[f][0]
        fam_nl[f] += vals[f][1]
    for k, v in info["timers"].items():
        timers[k] = timers.get(k, 0.0) + v
    wall += dt
    kept += 1
    if info["converged"]:
        break
nst = max(kept, 1)
tot = 0.0
for f, nm in names.items():
    print(f"{nm:14s} {fam_ms[f] / nst:8.2f} ms/step  {fam_nl[f] / nst:7.1f} launches/step")
    if f not in (9, 16):      # (apply_H contains other families; the host waits overlap device time)
        tot += fam_ms[f]
print(f"booked {tot / nst:.1f} ms/step of wall {1e3 * wall / nst:.1f} ms/step over {kept} {'steps' if keep_any else 'one-iteration steps'} "
      f"(LOBPCG iterations of the steps tried: {iters_seen}); host timers/step:",
      {k: round(1e3 * v / nst, 1) for k, v in timers.items()})
for key, (calls, ms, gf) in sorted(shape_tab.items(), key=lambda kv: -kv[1][1])[:40 if keep_any else 14]:
    print(f"[zgemm-shape] {key[0]} m={key[1]} n={key[2]} k={key[3]} flags={key[4]} calls={calls} ms={ms:.3f} TF/s={gf / ms:.2f}")
