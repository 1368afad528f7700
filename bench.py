#!/usr/bin/env python
"""bench.py -- SCF-iteration throughput of the MI355X plane-wave hot path (BASELINE.json metric).

Workload (``config.workload``), default = the north-star cell: fcc silicon 5 x 5 x 5 supercell (250 atoms,
1000 electrons), LDA (lda_x + lda_c_pw), HGH pseudopotential, Ecut = 30 Ha, Gamma only, FFT cube from
``compute_fft_size`` (192^3), fp64 -- BASELINE configs[4]'s "~1000 electrons" cell, which fits one GPU
(``--supercell 4`` = configs[1]).

What is timed: ONE WHOLE ``self_consistent_field`` in the sense of the reference's own ``scf_full`` benchmark
(benchmark/cases/common.jl:70-71): from ``guess_density`` (Gaussian superposition) to ``tol = 1e-6`` in the
density change, ScfAndersonDensitySolver, damping 0.8, AdaptiveDiagtol / AdaptiveBands defaults, random start
orbitals.  One "step" = one SCF iteration (build V = V_loc + V_H + V_xc, LOBPCG with H psi on the device,
``compute_density`` + all-reduce, energies, Anderson mixing).  ``--warmup W`` runs W SCF steps on a THROW-AWAY
stepper first (allocator, plan caches, lazy initialisation); the timed run then starts again from the guess
density.  ``--steps K`` caps the timed run: it stops at convergence or after K steps, and ``steps`` in the
JSON line is the number of steps actually run.  ``value`` = steps run / wall time, ``hpsi_applies_per_s`` =
n_matvec / wall time.

N > 1 (``--gpus N``, one process per GPU, launched by torch.distributed.run):
  --mode gamma   (default) the SAME Gamma-only workload, plane waves of the single k-block sharded over the
                 N GPUs as row slabs (``comm_pw``) -> strong scaling, N = 1 comparable.
  --mode kpoints BASELINE configs[2], fixed for every N: fcc Al, PBE, Ecut 40, 12x12x12 Monkhorst-Pack mesh with
                 the crystal symmetries (72 irreducible k-points, as the reference builds it), Gaussian smearing
                 T = 1e-3, LDOS mixing; split by ``distribute_kpoints`` with ONE density all-reduce per step ->
                 strong scaling (9 k-points per GPU at N = 8).
  --mode weak    one k-point of the Si supercell per GPU (weak scaling; the round-1 behaviour).

Prints ONE JSON line on rank 0 with ``roofline`` (dominant kernel family, HIP-event timed inside the library
on its own stream) and ``cpu_baseline`` (the CPU oracle timed on this host's cores).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
F64_MFMA_PEAK_TF = 78.6      # MI355X fp64 matrix peak (64-cycle v_mfma_f64_16x16x4_f64)

FAMILIES = {0: "zgemm_f64_mfma", 1: "fft_A_xbwd_scatter", 2: "fft_B_ybwd", 3: "fft_C_z_fused_V", 4: "fft_D_yfwd",
            5: "fft_E_xfwd_gather", 6: "density_z", 7: "heev_jacobi", 8: "potrf_trtri", 9: "apply_H_total",
            11: "zgemm_f64_mfma_structured", 13: "collectives", 15: "elementwise_nG_sized", 16: "host_waits"}
XGMI_LINK_GBS = 153.0        # 7 links x ~153 GB/s per GPU, ring collectives are per-link bound
COLLECTIVE_LATENCY_US = 25.0 # ASSUMED per-collective launch + synchronisation latency of RCCL on one node (not measured here)
PARITY_TOL_HA_PER_ATOM = 1e-8   # north star; the reference accepts 1e-9 .. 1e-10 Ha CPU <-> GPU (test/gpu.jl:30-31,113-114)
PARITY_SCF_TOL = 1e-8        # density tolerance the parity legs are converged to (E error ~ drho^2, eigenvalues ~ drho)
GOLDEN = {   # supercell n -> (fixture, k-mesh-compatible cube edge, primitive cells): SURVEY appendix B identity
    5: ("baseline_cfg5_prim_5x5x5_ecut30_fft40.json", 200, 125),
    4: ("baseline_cfg2_prim_4x4x4_ecut30_fft40.json", 160, 64),
}
KERNEL_FAMS = (0, 1, 2, 3, 4, 5, 6)      # candidates for "the dominant kernel" (0 stands for 0 + 11)
ONE_RANK_ENERGIES = os.path.join(ROOT, "tests", "golden", "device_one_rank_energies.json")


def one_rank_energy(workload_key):
    """Converged total energy of the ONE-rank device run of a workload (written by ``--write-reference-energy``; a
    device-vs-device fixture, NOT an oracle number: the oracle parity of the one-rank run is the `golden` leg)."""
    try:
        with open(ONE_RANK_ENERGIES) as fh:
            return json.load(fh).get(workload_key)
    except (OSError, ValueError):
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="cap on the timed SCF steps (stops earlier at convergence)")
    ap.add_argument("--warmup", type=int, default=3,
                    help="SCF steps on a throw-away stepper before the timed run (3: the Anderson / mixing paths of the second and "
                         "third step are warm as well -- their first-use cost is 50 ms, visible on the 0.4 s k-point SCFs)")
    ap.add_argument("--mode", choices=("gamma", "kpoints", "weak"), default="gamma")
    ap.add_argument("--supercell", type=int, default=5, help="n for the n x n x n Si supercell (5 = 1000 e-, 4 = configs[1])")
    ap.add_argument("--ecut", type=float, default=None)
    ap.add_argument("--kgrid", type=int, default=None, help="--mode kpoints: n of the n x n x n (graphene: n x n x 1) mesh")
    ap.add_argument("--system", choices=("al", "si", "graphene"), default="al",
                    help="--mode kpoints: BASELINE configs[2] (fcc Al PBE, Ecut 40, 12^3 mesh, default), configs[0] (Si primitive "
                         "LDA, Ecut 15, 4^3 mesh) or configs[3] (graphene slab PBE, Ecut 40, 9x9x1 mesh)")
    ap.add_argument("--no-symmetries", action="store_true", help="--mode kpoints: unreduced mesh")
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--no-gamma-real", action="store_true",
                    help="Gamma-only modes: iterate general complex orbitals exactly as the reference does, instead of the "
                         "real-symmetric ones (psi(-G) = conj psi(G)) the library uses at k = 0 by default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-random-start-leg", action="store_true",
                    help="skip the leg that repeats the timed SCF without the two-level start (random_start_value)")
    ap.add_argument("--no-complex-leg", action="store_true",
                    help="skip the second timed SCF with general complex orbitals (config.complex_iteration)")
    ap.add_argument("--cpu-step-budget", type=float, default=100.0,
                    help="run the REAL timed CPU step only if the sampled model predicts fewer seconds than this")
    ap.add_argument("--prof-all", action="store_true",
                    help="count kernel-family launches from the warm-up on (lines the counts up with a whole-process "
                         "rocprofv3 --pmc pass; tools/pmc_traffic_bench.sh)")
    ap.add_argument("--cpu-sample-bands", type=int, default=0, help="0 = one band per host core (max 256)")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the untimed parity legs (continuation of the timed SCFs to convergence, golden-fixture run)")
    ap.add_argument("--write-reference-energy", action="store_true",
                    help="N = 1: store the converged total energy of the parity leg in tests/golden/device_one_rank_energies.json "
                         "(what the N > 1 runs of the same workload are compared with)")
    ap.add_argument("--no-amdahl-probe", action="store_true",
                    help="--mode kpoints, N = 1: skip the timed SCF steps on one rank's share of the k-points (Amdahl model)")
    return ap.parse_args()


def prof_get(lib, basis, fam):
    ms, work, n = C.c_double(), C.c_double(), C.c_int64()
    from dftk_jl_amd._lib import check
    check(lib.dftk_mi_prof_get(basis.handle, fam, C.byref(ms), C.byref(work), C.byref(n)))
    return ms.value, work.value, n.value


def library_counts(lib):
    """(kernel launches, host synchronisations) the library has issued in this process so far (dftk_mi_launch_count)."""
    a, b_ = C.c_int64(), C.c_int64()
    lib.dftk_mi_launch_count(C.byref(a), C.byref(b_))
    return a.value, b_.value


def latency_roofline(run, steps, roof):
    """--mode kpoints: the many-small-k workloads are launch-latency problems (n_G ~ 1e3, 6-8 bands: every kernel is
    microseconds on kilobytes), so the bound that applies is neither HBM nor MFMA: it is launches and host
    synchronisations per SCF step.  achieved = wall microseconds per library launch; peak = the back-to-back launch
    rate of small kernels on one HIP stream (~5 us of host time per hipLaunchKernel here, profiles/r04_kpoints_share_
    step_N8_torch_profile.txt); frac = peak / achieved.  The MFMA numbers of the zgemm family stay in `mfma_view`."""
    launches = max(1, int(run["library_launches"]))
    wall_us = 1e6 * run["elapsed"]
    us_per_launch = wall_us / launches
    peak_us = 5.0
    return {"bound": "latency", "achieved": round(us_per_launch, 2), "peak": peak_us, "unit": "us/launch",
            "frac": round(peak_us / us_per_launch, 4), "traffic": None,
            "launches_per_step": round(launches / steps, 1),
            "host_syncs_per_step": round(run["library_host_syncs"] / steps, 1),
            "us_per_step": round(wall_us / steps, 1),
            "note": ("launch-latency bound: achieved = wall time of the timed SCF / kernel launches the library issued in "
                     "it (torch launches of the host mirror are not counted: they are what the native SCF glue removes); "
                     "peak = host cost of one back-to-back small launch on a HIP stream; frac = peak / achieved (1 = the "
                     "step is nothing but back-to-back launches)"),
            "mfma_view": {k_: roof[k_] for k_ in ("achieved", "peak", "unit", "frac", "kernel", "launches", "avg_launch_ms",
                                                  "families_ms", "families_launches") if k_ in roof}}


def library_source_hash():
    """Hash of the library's sources (also compiled into dftk_mi_version()): ties a PMC traffic file to the build."""
    from dftk_jl_amd import _build
    return _build.source_hash()


# ------------------------------------------------------------------------------------------ CPU leg
class BandWorkers:
    """Band-parallel local part of H psi and of the density on the host, as the reference threads them (one band per
    thread, FFT threads = 1: src/terms/Hamiltonian.jl:155, src/common/threading.jl:12).  Every worker thread owns ONE
    preallocated cube: no allocation and no page fault inside the timed loops (round 5 allocated a fresh 113 MB cube per
    band -- 256 threads then spend their time in the kernel's address-space lock).  The scatter, the
    in-place pocketfft transforms and the multiplication release the GIL."""

    def __init__(self, threads, N, shape, mapping, pot):
        import threading
        from concurrent.futures import ThreadPoolExecutor
        self.threads, self.N, self.shape, self.mapping, self.pot = int(threads), N, shape, mapping, pot
        self.pool = ThreadPoolExecutor(max_workers=self.threads)
        self.tls = threading.local()

    def _cube(self):
        c = getattr(self.tls, "cube", None)
        if c is None:
            c = self.tls.cube = np.zeros(self.N, dtype=complex)
        return c

    def local_one(self, col):
        import scipy.fft as sfft
        cube = self._cube()
        cube.fill(0.0)
        cube[self.mapping] = col
        c3 = sfft.ifftn(cube.reshape(self.shape), workers=1, norm="forward", overwrite_x=True)
        c1 = c3.reshape(self.N)
        c1 *= self.pot
        c3 = sfft.fftn(c1.reshape(self.shape), workers=1, norm="backward", overwrite_x=True)
        return c3.reshape(self.N)[self.mapping]

    def dens_one(self, col):
        import scipy.fft as sfft
        cube = self._cube()
        cube.fill(0.0)
        cube[self.mapping] = col
        c3 = sfft.ifftn(cube.reshape(self.shape), workers=1, norm="forward", overwrite_x=True)
        return c3.real ** 2 + c3.imag ** 2

    def shutdown(self):
        self.pool.shutdown()


def sweep_band_threads(N, shape, mapping, pot, cols, cores, avail_bytes):
    """H psi applies per second of the band-parallel host leg for 32 / 64 / 128 / 256 threads (and the core count),
    two waves of bands each (bounded: a wave is one 3-D FFT pair per thread); returns ({threads: rate}, best)."""
    cand = sorted({t for t in (32, 64, 128, 256, cores) if 1 <= t <= cores and t * 16 * N * 3 < avail_bytes})
    if not cand:
        cand = [max(1, min(cores, int(avail_bytes // (3 * 16 * N))))]
    rates = {}
    for t in cand:
        w = BandWorkers(t, N, shape, mapping, pot)
        nb = min(t, 256)
        list(w.pool.map(w.local_one, (cols[:, j % cols.shape[1]] for j in range(nb))))        # warm: buffers touched
        t0 = time.time()
        list(w.pool.map(w.local_one, (cols[:, j % cols.shape[1]] for j in range(2 * nb))))
        rates[t] = 2 * nb / (time.time() - t0)
        w.shutdown()
    best = max(rates, key=rates.get)
    return {str(k_): round(v_, 2) for k_, v_ in rates.items()}, best


def rule_of_thumb(basis, cores):
    """The reference's own "very (very) rough" estimate of the time per SCF step (docs/src/tricks/parallelization.md:57-73:
    30 ms per FFT on a 128^3 grid x grid points x k-points x occupied states x 8 FFT steps per state, no
    parallelisation), and what ideal scaling over this host's cores would make of it."""
    n_occ = -(-basis.model.n_electrons // basis.model.filled_occupation)
    serial = 30e-3 / 128 ** 3 * float(np.prod(basis.fft_size)) * len(basis.kcoords_global) * n_occ * 8
    return {"rule_of_thumb_serial_s_per_step": round(serial, 1),
            "rule_of_thumb_it_per_s": round(cores / serial, 4),
            "rule_of_thumb_note": ("estimate_time_per_scf_step of docs/src/tricks/parallelization.md:61-73 (FFT-limited, "
                                   f"unparallelised) divided by the {cores} host cores, i.e. DFTK with IDEAL thread scaling")}

def cpu_baseline_gamma(basis, info, n_sample, per_step):
    """kind = "port": the NumPy/SciPy oracle's arithmetic timed on this host, BATCHED over bands so that every
    core is busy (the reference threads H psi over bands, src/terms/Hamiltonian.jl:155, src/common/threading.jl),
    on a bounded sample of the same workload; scaled to SCF iterations/s with the operation counts of the device
    run.  Not DFTK itself (no Julia in the image)."""
    import scipy.fft as sfft
    import oracle
    from oracle.terms import HamiltonianBlock
    t_all = time.time()
    cores = os.cpu_count()
    kpt = basis.kpoints[0]
    T = basis.terms
    nx, ny, nz = basis.fft_size
    N = basis.N

    class OB:   # the minimal basis interface oracle.HamiltonianBlock needs
        pass
    ob = OB()
    ob.fft_size, ob.N = basis.fft_size, basis.N
    ob.fft_normalization, ob.ifft_normalization = basis.fft_normalization, basis.ifft_normalization
    ob.ifft = oracle.PlaneWaveBasis.ifft.__get__(ob)
    ob.fft = oracle.PlaneWaveBasis.fft.__get__(ob)
    okpt = oracle.Kpoint(1, kpt.coordinate, kpt.G_vectors.cpu().numpy(), kpt.mapping)
    P = T.P[0].cpu().numpy().T if T.P is not None else None       # (n_G, n_p)
    V = info["ham"][0].potential.cpu().numpy()
    kin = kpt.kinetic.cpu().numpy()
    H = HamiltonianBlock(ob, okpt, kin, V, P, T.D)
    rng = np.random.default_rng(0)
    n_G, M = kpt.n_G, info["psi"][0].shape[0]
    psi = rng.standard_normal((n_G, n_sample)) + 1j * rng.standard_normal((n_G, n_sample))
    pot = V.reshape(-1) * (basis.fft_normalization * basis.ifft_normalization)

    def hpsi_batched(block):          # the oracle's H psi (oracle/terms.py:HamiltonianBlock.mul), all bands at once
        nb = block.shape[1]
        cube = np.zeros((nb, N), dtype=complex)
        cube[:, kpt.mapping] = block.T
        cube = sfft.ifftn(cube.reshape(nb, nz, ny, nx), axes=(1, 2, 3), workers=cores, norm="forward", overwrite_x=True)
        cube = cube.reshape(nb, N)
        cube *= pot[None, :]
        cube = sfft.fftn(cube.reshape(nb, nz, ny, nx), axes=(1, 2, 3), workers=cores, norm="backward", overwrite_x=True)
        out = cube.reshape(nb, N)[:, kpt.mapping].T + kin[:, None] * block
        if P is not None:
            out = out + P @ (T.D @ (P.conj().T @ block))
        return out

    chk = hpsi_batched(psi[:, :2])
    ref = H.mul(psi[:, :2])
    err = float(np.linalg.norm(chk - ref) / np.linalg.norm(ref))
    assert err < 1e-12, f"batched CPU H psi deviates from the oracle: {err}"
    # band-parallel like the reference (one band per thread, FFT threads = 1: src/common/threading.jl:12,
    # src/terms/Hamiltonian.jl:155): worker threads with one preallocated cube each, thread count chosen by a sweep
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 32 << 30
    shape = (nz, ny, nx)
    sweep, threads = sweep_band_threads(N, shape, kpt.mapping, pot, psi, cores, avail)
    P_h = P
    D_h = T.D
    W = BandWorkers(threads, N, shape, kpt.mapping, pot)
    list(W.pool.map(W.local_one, (psi[:, n] for n in range(min(threads, n_sample)))))          # warm
    t0 = time.time()
    loc = list(W.pool.map(W.local_one, (psi[:, n] for n in range(n_sample))))
    loc = [l_ + kin * psi[:, n] for n, l_ in enumerate(loc)]
    if P_h is not None:                                                 # nonlocal part: one threaded zgemm pair
        nl = P_h @ (D_h @ (P_h.conj().T @ psi))
    t_hpsi = (time.time() - t0) / n_sample
    one = np.stack(loc[:2], axis=1) + (nl[:, :2] if P_h is not None else 0.0)
    err2 = float(np.linalg.norm(one - ref) / np.linalg.norm(ref))
    assert err2 < 1e-12, f"band-parallel CPU H psi deviates from the oracle: {err2}"
    t0 = time.time()
    acc = None
    for d in W.pool.map(W.dens_one, (psi[:, n] for n in range(n_sample))):
        acc = d if acc is None else acc + d
    t_dens = (time.time() - t0) / n_sample
    W.shutdown()
    bsz = threads
    # dense algebra rate: Gram matrix and rotation of a panel (threaded OpenBLAS zgemm)
    mcols = M                      # panels of the true width (LOBPCG's are M .. 3M wide)
    Xs = rng.standard_normal((n_G, mcols)) + 1j * rng.standard_normal((n_G, mcols))
    G = Xs.conj().T @ Xs
    t0 = time.time()
    G = Xs.conj().T @ Xs
    Xs @ G
    t_blas = time.time() - t0
    rate = 2 * 8.0 * n_G * mcols * mcols / t_blas                  # flop/s of zgemm on this host
    n_occ = basis.model.n_electrons // 2
    t_step = (per_step["n_matvec"] * t_hpsi + per_step["zgemm_flops"] / rate + n_occ * t_dens)
    late_flops = 80.0 * n_G * M * M         # one 1-iteration LOBPCG call: Gram / update products of a late step
    return {"value": 1.0 / t_step, "unit": "SCF iterations/s", "cores": cores, "kind": "port",
            "hpsi_applies_per_s": 1.0 / t_hpsi, "threads": threads, "thread_sweep_hpsi_per_s": sweep,
            **rule_of_thumb(basis, cores),
            "model_terms": {"hpsi_s_per_band": t_hpsi, "density_s_per_band": t_dens, "zgemm_gflops": rate / 1e9,
                            "late_step_zgemm_s": late_flops / rate},
            "sample": (f"NumPy/SciPy oracle arithmetic (not DFTK: no Julia here), band-parallel as the reference: "
                       f"{bsz} threads x one band each (pocketfft workers=1, one preallocated cube per thread; best of the sweep "
                       f"{sweep} H psi/s) + threaded OpenBLAS: {n_sample} bands of H psi "
                       f"({t_hpsi * 1e3:.2f} ms/band; checked against oracle.HamiltonianBlock.mul to {err:.1e}), "
                       f"{n_sample} density bands ({t_dens * 1e3:.2f} ms/band), a {mcols}-column zgemm panel "
                       f"({rate / 1e9:.0f} GF/s); one SCF step modelled as the device run's per-step averages: "
                       f"n_matvec={per_step['n_matvec']:.0f} H psi + {per_step['zgemm_flops'] / 1e12:.2f} TF of zgemm + "
                       f"{n_occ} density bands; sample wall {time.time() - t_all:.1f} s")}


def cfg1_scf_3steps(device):
    """A REAL SCF on both sides (no model): BASELINE configs[0] (Si primitive, LDA, Ecut 15, unreduced 4x4x4 mesh)
    run as the reference's ``scf_3steps`` benchmark (benchmark/cases/common.jl:70: maxiter = 3) by the CPU oracle
    and by the device path; wall times in seconds."""
    import dftk_jl_amd as dftk
    import oracle
    lat, atoms, pos = oracle.basis.silicon_primitive(a=10.26, functional="lda")
    om = oracle.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
    t0 = time.time()
    ob = oracle.PlaneWaveBasis(om, 15, oracle.MonkhorstPack((4, 4, 4)))
    ores = oracle.self_consistent_field(ob, tol=1e-6, maxiter=3)
    t_cpu = time.time() - t0
    lat, atoms, pos = dftk.silicon_cell()
    dm = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
    db = dftk.PlaneWaveBasis(dm, 15, dftk.MonkhorstPack((4, 4, 4)), device=device)
    dftk.self_consistent_field(db, tol=1e-6, maxiter=1)      # warm
    t0 = time.time()
    dres = dftk.self_consistent_field(db, tol=1e-6, maxiter=3)
    t_dev = time.time() - t0
    return {"workload": "configs[0]: Si primitive LDA Ecut 15, unreduced 4x4x4 k-mesh, fft 27^3, scf_3steps (maxiter = 3)",
            "cpu_oracle_s": round(t_cpu, 2), "device_s": round(t_dev, 3),
            "cpu_it_per_s": round(3 / t_cpu, 4), "device_it_per_s": round(3 / t_dev, 3),
            "E_total_cpu": ores["energies"].total, "E_total_device": dres["energies"].total,
            "note": "64 tiny k-blocks (n_G ~ 725, 7 bands): launch-latency bound on the device"}


# ------------------------------------------------------------------------------------------ CPU leg: a timed step
def cpu_timed_late_step(basis, info, diagtol, budget_s, dinfo=None, threads_hint=None):
    """kind = "port, timed step": ONE real SCF step of the reference's algorithm on the host cores, with wall seconds,
    beside the device's time for the same step.  Input = the converged state of the device run (psi, rho): from there
    a step is what every late SCF step of this workload is -- H[rho] is rebuilt, LOBPCG (general complex orbitals,
    the oracle's restatement of lobpcg_hyper_impl.jl) starts from the previous orbitals and needs ONE iteration at
    the step's ``diagtol`` (2 M H psi applies, a 2M x 2M Rayleigh-Ritz, block updates of the true width), then
    ``compute_density`` over the occupied bands.  H psi and the density are band-parallel over all cores exactly like
    the sampled leg above (one band per thread, pocketfft workers = 1); the dense algebra is NumPy's threaded OpenBLAS.
    Returns None when the host lacks the memory (13 blocks of n_G x M complex) or the sampled model predicts more than
    ``budget_s`` seconds."""
    import scipy.fft as sfft
    import oracle
    from oracle.lobpcg import PreconditionerTPA, lobpcg_hyper
    from concurrent.futures import ThreadPoolExecutor
    cores = os.cpu_count()
    kpt = basis.kpoints[0]
    T = basis.terms
    nx, ny, nz = basis.fft_size
    N = basis.N
    n_G, M = kpt.n_G, info["psi"][0].shape[0]
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 64 << 30
    need = 16 * n_G * M * 16 + 16 * n_G * (T.D.shape[0] if T.D is not None else 0) * 2
    if avail < need + (8 << 30):
        return None
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max((p_.get("num_threads", 0) for p_ in threadpool_info() if p_.get("user_api") == "blas"),
                           default=0)
    except Exception:
        blas_threads = 0
    P = T.P[0].cpu().numpy().T.copy() if T.P is not None else None       # (n_G, n_p)
    D = T.D
    V = info["ham"][0].potential.cpu().numpy()
    kin = kpt.kinetic.cpu().numpy()
    pot = V.reshape(-1) * (basis.fft_normalization * basis.ifft_normalization)
    X0 = info["psi"][0].cpu().numpy().T.copy()                            # (n_G, M) general complex orbitals
    mapping = kpt.mapping
    threads = int(threads_hint) if threads_hint else int(max(1, min(cores, M, avail // (8 * 16 * N))))
    W = BandWorkers(threads, N, (nz, ny, nx), mapping, pot)
    pool = W.pool
    n_hpsi = [0]

    def A(block):                       # mul!(H psi, H, psi) (Hamiltonian.jl:137-192), band-parallel
        n_hpsi[0] += block.shape[1]
        out = np.empty_like(block)
        for j, col in enumerate(pool.map(W.local_one, (block[:, j] for j in range(block.shape[1])))):
            out[:, j] = col
        out += kin[:, None] * block
        if P is not None:
            out += P @ (D @ (P.conj().T @ block))
        return out

    dens_one = W.dens_one

    # check the band-parallel operator against the oracle's own (2 bands)
    class OB:
        pass
    ob = OB()
    ob.fft_size, ob.N = basis.fft_size, basis.N
    ob.fft_normalization, ob.ifft_normalization = basis.fft_normalization, basis.ifft_normalization
    ob.ifft = oracle.PlaneWaveBasis.ifft.__get__(ob)
    ob.fft = oracle.PlaneWaveBasis.fft.__get__(ob)
    okpt = oracle.Kpoint(1, kpt.coordinate, kpt.G_vectors.cpu().numpy(), kpt.mapping)
    ref = oracle.terms.HamiltonianBlock(ob, okpt, kin, V, P, D).mul(X0[:, :2])
    err = float(np.linalg.norm(A(X0[:, :2]) - ref) / np.linalg.norm(ref))
    assert err < 1e-12, f"band-parallel CPU H psi deviates from the oracle: {err}"
    n_hpsi[0] = 0
    n_conv = int(info["n_bands_converge"])
    t0 = time.time()
    res = lobpcg_hyper(A, X0, prec=PreconditionerTPA(kin), tol=diagtol, n_conv_check=n_conv, maxiter=100)
    t_lobpcg = time.time() - t0
    occ = np.asarray(info["occupation"][0], dtype=float)
    t0 = time.time()
    rho = np.zeros((nz, ny, nx))
    live = [j for j in range(M) if abs(occ[j]) > 1e-8]
    for j, d in zip(live, pool.map(dens_one, (res["X"][:, j] for j in live))):
        rho += occ[j] * d
    rho *= basis.kweights[0] / basis.model.unit_cell_volume
    t_dens = time.time() - t0
    pool.shutdown()
    # the device ran the SAME step (device_late_step: same potential, same start orbitals, same tolerance): its
    # density and eigenvalues are what the host's must agree with, to the step's diagonalisation tolerance
    cmp_ = dinfo if dinfo is not None else info
    drho = float(np.linalg.norm(rho - cmp_["rho"].cpu().numpy()) * np.sqrt(basis.dvol))
    dlam = float(np.max(np.abs(res["λ"][:n_conv] - np.asarray(cmp_["eigenvalues"][0])[:n_conv])))
    return {"cpu_step_s": round(t_lobpcg + t_dens, 2), "cpu_lobpcg_s": round(t_lobpcg, 2), "cpu_density_s": round(t_dens, 2),
            "lobpcg_iterations": int(res["n_iter"]), "n_matvec": int(res["n_matvec"]), "converged": bool(res["converged"]),
            "diagtol": diagtol, "fft_threads": threads, "blas_threads": blas_threads,
            "hpsi_check_vs_oracle": err, "drho_vs_device": drho, "max_eigenvalue_diff_vs_device": dlam}


def device_late_step(dftk, basis, info, tol, diagtol):
    """The device's wall time for the same step as ``cpu_timed_late_step``: one more SCF step from the converged state
    at the diagonalisation tolerance ``diagtol`` of a typical late step of the timed run."""
    st = dftk.ScfStepper(basis, rho=info["rho"], psi=info["psi"], tol=tol, determine_tol=lambda n_iter, hist: diagtol)
    st.info.update(eigenvalues=info["eigenvalues"], occupation=info["occupation"], eF=info["eF"], n_iter=2,
                   history_drho=list(info["history_drho"]))
    import torch
    torch.cuda.synchronize()
    t0 = time.time()
    out = st.step()
    torch.cuda.synchronize()
    return time.time() - t0, out


# ------------------------------------------------------------------------------------------ the timed SCF
def run_scf(dftk, lib, basis, args, barrier, world, dist, torch, coarse_start=True):
    """--warmup steps on a throw-away stepper, then ONE whole self_consistent_field capped at --steps, HIP-event
    family timings switched on for the timed part.  Returns the numbers of the JSON line."""
    from dftk_jl_amd._lib import check
    if args.prof_all:
        check(lib.dftk_mi_prof_enable(basis.handle, 1))
    if args.warmup > 0:
        warm = dftk.ScfStepper(basis, tol=args.tol, coarse_start=coarse_start)
        for _ in range(args.warmup):
            if warm.step()["converged"]:
                break
        del warm
    barrier()
    if not args.prof_all:
        check(lib.dftk_mi_prof_enable(basis.handle, 3))        # 3: families + the per-shape zgemm table (slab replay)
    iters, diagtols, step_s, nmv_steps = [], [], [], []
    host_timers = {}
    counts0 = library_counts(lib)
    t0 = time.time()
    # guess_density is part of self_consistent_field.  The stepper's own convergence test is the (tighter) PARITY
    # tolerance so that it can be continued, untimed, after the timed loop; the timed loop stops at --tol exactly as
    # self_consistent_field would (the only difference: the last timed step also mixes, ~4 ms inside the timed region)
    ptol = min(PARITY_SCF_TOL, args.tol)
    stepper = dftk.ScfStepper(basis, tol=args.tol, is_converged=lambda info_: info_["history_drho"][-1] < ptol,
                              phase_timers=(args.mode != "kpoints"), coarse_start=coarse_start)
    info = None
    nmv_coarse = 0
    for _ in range(max(args.steps, 1)):
        ts = time.time()
        info = stepper.step()
        step_s.append(time.time() - ts)
        iters.append(float(np.mean(info["diagonalization"]["n_iter"])))
        diagtols.append(info["diagtol"])
        nmv_steps.append(int(info["n_matvec_step"]))
        nmv_coarse += int(info.get("n_matvec_coarse", 0))
        for k_, v_ in info["timers"].items():
            host_timers[k_] = host_timers.get(k_, 0.0) + v_
        if os.environ.get("DFTK_MI_BENCH_STEP_TIMERS"):      # where a slow step spent its time (stderr, rank 0)
            if int(os.environ.get("RANK", "0")) == 0:
                print(f"[step {len(step_s)}] {step_s[-1] * 1e3:.1f} ms: "
                      + ", ".join(f"{k_} {v_ * 1e3:.1f}" for k_, v_ in info["timers"].items()), file=sys.stderr)
        if info["history_drho"][-1] < args.tol:
            break
    info = dict(stepper.finalize())                          # energies + Hamiltonian of the final state, as the reference
    info["converged"] = bool(info["history_drho"][-1] < args.tol)
    barrier()
    elapsed = time.time() - t0
    counts1 = library_counts(lib)
    check(lib.dftk_mi_prof_enable(basis.handle, 0))
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    fam = {f: prof_get(lib, basis, f) for f in list(FAMILIES) + [10, 12, 14, 17, 18]}
    shapes = []
    try:
        cap = 512
        rows = (C.c_int64 * (6 * cap))()
        sms = (C.c_double * cap)()
        cnt = C.c_int()
        check(lib.dftk_mi_prof_zgemm_shapes(basis.handle, cap, rows, sms, C.byref(cnt)))
        shapes = [dict(trans="NC"[rows[6 * i]], m=int(rows[6 * i + 1]), n=int(rows[6 * i + 2]), k=int(rows[6 * i + 3]),
                       flags=int(rows[6 * i + 4]), calls=int(rows[6 * i + 5]), ms=float(sms[i])) for i in range(min(cnt.value, cap))]
    except Exception:
        shapes = []
    return dict(zgemm_shapes=shapes, info=info, elapsed=elapsed, iters=iters, diagtols=diagtols, step_s=step_s, nmv_steps=nmv_steps,
                host_timers=host_timers, fam=fam, stepper=stepper, n_matvec_coarse=nmv_coarse,
                library_launches=counts1[0] - counts0[0], library_host_syncs=counts1[1] - counts0[1])


def continue_to_parity(stepper, max_extra=80):
    """UNTIMED: carry the stepper of a timed (possibly capped) SCF on to the parity tolerance and return the converged
    total energy / spectrum -- what the parity block of the JSON line is computed from."""
    t0 = time.time()
    extra = 0
    info = stepper.info
    while not info.get("converged") and extra < max_extra:
        info = stepper.step()
        extra += 1
    info = stepper.finalize()
    nconv = int(info["n_bands_converge"])
    return {"converged": bool(info["converged"]), "E_total": float(info["energies"].total),
            "eigenvalues": [np.asarray(l_, dtype=float)[:nconv].copy() for l_ in info["eigenvalues"]],
            "extra_steps": extra, "steps_total": int(info["n_iter"]), "drho": float(info["history_drho"][-1]),
            "wall_s": round(time.time() - t0, 2)}


def golden_leg(dftk, model, ecut, n_super, device):
    """Parity against the ORACLE at the headline size: the supercell at Gamma with the cube that makes it identical to
    the oracle's primitive cell on the n x n x n k-mesh at 40^3 (cell_to_supercell, src/supercell.jl:27-53).  The
    production cube of the 5x5x5 cell is 192 = 5 x 38.4: no integer primitive cube exists for it, 200 = 5 x 40 is the
    nearest cube for which the identity is exact; the fixture was produced by tools/make_golden_baseline.py."""
    name, cube, n_prim = GOLDEN[n_super]
    with open(os.path.join(ROOT, "tests", "golden", name)) as fh:
        g = json.load(fh)
    assert abs(g["Ecut"] - ecut) < 1e-12 and tuple(g["functionals"]) == ("lda_x", "lda_c_pw")
    t0 = time.time()
    basis = dftk.PlaneWaveBasis(model, ecut, dftk.MonkhorstPack((1, 1, 1)), device=device, fft_size=(cube,) * 3)
    res = dftk.self_consistent_field(basis, tol=PARITY_SCF_TOL)
    n_atoms = len(model.positions)
    n_occ = model.n_electrons // 2
    union = np.sort(np.concatenate([np.array(lam)[:4] for lam in g["eigenvalues"]]))[:n_occ]
    got = np.sort(np.asarray(res["eigenvalues"][0]))[:n_occ]
    out = {"fixture": f"tests/golden/{name}", "cube": cube, "converged": bool(res["converged"]), "steps": int(res["n_iter"]),
           "wall_s": round(time.time() - t0, 2),
           "dE_total_vs_golden_per_atom": float(res["energies"].total - n_prim * g["E_total"]) / n_atoms,
           "max_deigenvalue_vs_golden": float(np.max(np.abs(got - union))),
           "max_dterm_per_primitive_cell": float(max(abs(res["energies"][k_] / n_prim - v_) for k_, v_ in g["energies"].items())),
           "why_this_cube": (f"the oracle restates the supercell as the primitive cell on the {n_super}^3 k-mesh at 40^3, an identity "
                             f"that needs a cube of {n_super} x 40 = {cube}; the production cube ({'x'.join(map(str, (192,) * 3)) if n_super == 5 else '150^3'}) "
                             f"has no integer primitive counterpart, so it is tied to this leg through "
                             f"dE_total_real_vs_complex (same cube, the reference's own iteration)")}
    del res, basis
    return out


def roofline_of(fam, n_gpus, workload):
    """``roofline`` object of the JSON line from the HIP-event family timings of one timed SCF."""
    zg_ms = fam[0][0] + fam[11][0]
    zg_useful = fam[0][1] + fam[11][1]
    zg_launch = fam[0][2] + fam[11][2]
    fam_ms = {f: (zg_ms if f == 0 else fam[f][0]) for f in KERNEL_FAMS}
    dom = max((f for f in KERNEL_FAMS if fam_ms[f] > 0), key=lambda f: fam_ms[f])
    if dom == 0:
        ms, work, launches = zg_ms, zg_useful, zg_launch
        roof = {"bound": "mfma", "achieved": work / (ms * 1e-3) / 1e12, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s"}
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["achieved_unstructured"] = (fam[0][1] / (fam[0][0] * 1e-3) / 1e12) if fam[0][0] > 0 else None
        roof["mfma_executed_tflops"] = fam[12][1] / (ms * 1e-3) / 1e12
        roof["mfma_busy_frac"] = roof["mfma_executed_tflops"] / F64_MFMA_PEAK_TF
        roof["note"] = ("achieved = USEFUL flops / time over all zgemm launches: 8mnk for an unstructured complex "
                        "call, 4mnk for a REAL call (real-symmetric Gamma orbitals: the product IS a real GEMM of "
                        "that many flops, nothing is saved by a trick); for UPPER (Gram, only i <= j needed) and "
                        "B_UPPER (X inv(R), k <= j) only the mathematically needed part.  achieved_unstructured = "
                        "the same for the unstructured calls alone.  mfma_executed_tflops = real flops the launched "
                        "tiles run on the matrix pipe (3M complex product: 6 per complex multiply-add, REAL: 4; "
                        "whole tiles incl. shifted / border recompute) / time; mfma_busy_frac = that / dense f64 "
                        "MFMA peak.  Call mix: a LOBPCG call of an SCF step starts from the kept A X (DESIGN 3.8c), i.e. "
                        "without the two projector products of an H X -- the launches that ran at the highest rate (56-59 "
                        "TF/s); the same kernels on the call mix with a full H X per call (DFTK_MI_AX_REUSE=0): 0.62-0.64")
    else:
        ms, work, launches = fam[dom]
        roof = {"bound": "hbm", "achieved": work / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        roof["frac"] = roof["achieved"] / roof["peak"]
    # HBM bytes per launch of the dominant family from PMC passes over THIS command and THIS build
    # (tools/pmc_traffic_bench.sh writes profiles/<round>_pmc_traffic.json with the library's source hash)
    roof["traffic"] = None
    import glob
    pmc_files = sorted((os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json"))),
                       reverse=True)       # newest round first; only a file of THIS build (lib_hash) is taken
    for pmc_file in pmc_files:
        try:
            with open(os.path.join(ROOT, "profiles", pmc_file)) as fh:
                pmc = json.load(fh)
            key = FAMILIES[dom]
            if (n_gpus == 1 and pmc.get("lib_hash") == library_source_hash() and pmc.get("workload") == workload
                    and key in pmc["families"]):       # (collected on one GPU: says nothing about a sharded launch mix)
                roof["traffic"] = pmc["families"][key]["bytes_per_launch"]
                roof["traffic_unit"] = "B/launch"
                roof["traffic_source"] = f"profiles/{pmc_file} (" + pmc["collected"] + ")"
                break
        except (OSError, KeyError, ValueError):
            pass
    roof["algorithmic_bytes_per_launch"] = (fam[10][1] / max(launches, 1)) if dom == 0 else work / max(launches, 1)
    roof["kernel"] = FAMILIES[dom]
    roof["launches"] = launches
    roof["avg_launch_ms"] = ms / max(launches, 1)
    roof["families_ms"] = {FAMILIES[f]: round(fam[f][0], 3) for f in FAMILIES}
    roof["families_launches"] = {FAMILIES[f]: int(fam[f][2]) for f in FAMILIES}
    roof["families_work"] = {FAMILIES[f]: fam[f][1] for f in FAMILIES}   # flops (zgemm) / algorithmic bytes (FFT)
    roof["families_work"]["zgemm_executed_real_flops"] = fam[12][1]
    roof["families_work"]["zgemm_operand_bytes"] = fam[10][1]
    roof["families_rate"] = {
        FAMILIES[f]: (round(fam[f][1] / (fam[f][0] * 1e-3) / (1e12 if f in (0, 11) else 1e9), 2)
                      if fam[f][0] > 0 and (f < 7 or f == 11) else None) for f in FAMILIES}
    return roof


def step_roofline(fam, elapsed_s, steps, step_s, iters):
    """Flat scalars for ``config`` (the driver's record keeps only scalars there): what the timed window would take at
    the rooflines of its OWN useful work -- zgemm useful flops at the dense f64 MFMA peak plus the algorithmic bytes of
    the FFT stages, the density pass and the n_G-sized element-wise kernels at the HBM peak -- over its wall time.
    Everything else (eigensolver, Cholesky, host glue, launch gaps) counts as pure overhead."""
    flops = fam[0][1] + fam[11][1]
    nbytes = sum(fam[f][1] for f in (1, 2, 3, 4, 5, 6, 15))
    t_roof = flops / (F64_MFMA_PEAK_TF * 1e12) + nbytes / (HBM_PEAK_GBS * 1e9)
    t_roof63 = flops / (F64_MFMA_PEAK_TF * 1e12) + nbytes / 6.3e12        # 6.3 TB/s: the achievable copy rate
    late = [1e3 * s_ for s_, i_ in zip(step_s, iters) if abs(i_ - 1.0) < 1e-9]
    return {"step_roofline_frac": round(t_roof / elapsed_s, 4),
            "step_roofline_frac_hbm_6p3": round(t_roof63 / elapsed_s, 4),
            "step_roofline_ms": round(1e3 * t_roof / steps, 2),
            "late_step_ms": round(float(np.median(late)), 2) if late else None,
            "late_steps_counted": len(late)}


def sharded_self_check(dftk, basis, comm_size):
    """--mode gamma, N > 1, before the SCF: one sharded H psi and one Gram matrix of a seeded random block against the
    UNSHARDED result computed from the gathered block on every rank -- a broken slab <-> band all-to-all plan or
    all-reduce shows up here, in seconds, instead of as a non-converging SCF.  Returns the relative deviations."""
    import torch
    kpt = basis.kpoints[0]
    nb = 8
    gen = torch.Generator(device="cpu").manual_seed(1234)
    full = torch.randn((nb, kpt.n_G, 2), dtype=torch.float64, generator=gen)
    full = torch.view_as_complex(full).to(basis.device)
    _, ham = dftk.energy_hamiltonian(basis, None, None, rho=dftk.guess_density(basis))
    H = ham[0]
    loc = full[:, kpt.row0:kpt.row1].contiguous()
    got = H @ loc                                         # sharded apply (this rank's rows of H psi)
    # reference: gather the rows of H psi over the ranks, and compare norms / Gram with a replicated evaluation
    parts = basis.comm_pw.gather_lists(torch.view_as_real(got).cpu().numpy().tolist()) if comm_size > 1 else None
    gram_loc = loc.conj() @ got.T                         # partial Gram psi' H psi from this rank's rows
    g = torch.view_as_real(gram_loc.contiguous()).reshape(-1).clone()
    basis.comm_pw.sum_(g, basis.stream_ptr)
    basis.sync()
    gram = torch.view_as_complex(g.reshape(nb, nb, 2))
    herm = float((gram - gram.conj().T).abs().max() / gram.abs().max())
    out = {"gram_hermiticity": herm}
    if parts is not None:
        Hfull = torch.cat([torch.view_as_complex(torch.tensor(p_, dtype=torch.float64).reshape(nb, -1, 2).contiguous())
                           for p_ in parts], dim=1).to(basis.device)
        ref = full.conj() @ Hfull.T
        out["gram_vs_gathered"] = float((gram - ref).abs().max() / ref.abs().max())
    return out


def kpoints_self_check(basis, comm, world):
    """--mode kpoints / weak, N > 1, before the SCF: the density all-reduce (mpi_sum!(rho, comm_kpts), src/densities.jl:46)
    on a rank-stamped cube through the SAME communicator the SCF uses -- every entry must come back as
    sum_r (r + 1) (i + 1 mod 7) exactly (small integers: exact in fp64 whatever the reduction order)."""
    import torch
    n = int(basis.N)
    i = torch.arange(n, dtype=torch.float64, device=basis.device) % 7 + 1.0
    x = (comm.rank + 1.0) * i
    comm.sum_(x, basis.stream_ptr)
    basis.sync()
    want = (world * (world + 1) / 2.0) * i
    return {"allreduce_rank_stamped_cube_max_abs_error": float((x - want).abs().max()), "cube_entries": n}


# ------------------------------------------------------------------------------------------ Amdahl model
def _t_allreduce_ms(n_gpus, calls, nbytes):
    """Ring all-reduce on point-to-point xGMI (SURVEY section 2.4 / 8d Unit E): 2 (p - 1) / p x bytes per link."""
    return calls * COLLECTIVE_LATENCY_US * 1e-3 + 2.0 * (n_gpus - 1) / n_gpus * nbytes / (XGMI_LINK_GBS * 1e9) * 1e3


def _t_alltoall_ms(n_gpus, calls, nbytes_total):
    """Slab <-> band transposition of blocks totalling `nbytes_total` (all ranks together): every GPU sends 1 / p^2 of
    it to each peer over that peer's own link."""
    return calls * COLLECTIVE_LATENCY_US * 1e-3 + nbytes_total / n_gpus ** 2 / (XGMI_LINK_GBS * 1e9) * 1e3


def amdahl_gamma(run, steps, n_cube):
    """--mode gamma measured on ONE GPU -> predicted plane-wave-sharded step on N GPUs.  Row-local work (GEMMs, FFT
    pipeline over this rank's bands, density, n_G-sized element-wise kernels: families timed with HIP events) divides by
    N; everything else of the step's wall time is REPLICATED (Rayleigh-Ritz heev, Cholesky + inverse, the host
    synchronisations of the LOBPCG control flow, the cube-sized potential / mixing / energy work, Python glue); the
    collectives a sharded run performs are counted by the library on one rank too (families 17 / 18) and priced with
    the xGMI link bandwidth plus an assumed latency per call."""
    fam = run["fam"]
    wall = 1e3 * run["elapsed"] / steps
    sharded_f = {FAMILIES.get(f, str(f)): fam[f][0] / steps for f in (0, 11, 1, 2, 3, 4, 5, 6, 15)}
    sharded = sum(sharded_f.values())
    named = {"heev_jacobi": fam[7][0] / steps, "potrf_trtri": fam[8][0] / steps,
             "host_waits_wall": fam[16][0] / steps, "host_waits_count": fam[16][2] / steps}
    for k_ in ("energy_hamiltonian", "energies", "mixing"):
        named["host_timer_" + k_] = 1e3 * run["host_timers"].get(k_, 0.0) / steps
    replicated = wall - sharded
    ar_calls, ar_bytes = fam[17][2] / steps, fam[17][1] / steps
    a2a_calls, a2a_bytes = fam[18][2] / steps, fam[18][1] / steps
    pred, sp, comm = {}, {}, {}
    for n in (2, 4, 8):
        c = (_t_allreduce_ms(n, ar_calls + 1, ar_bytes + 8.0 * n_cube)        # + the density all-reduce of the step
             + _t_alltoall_ms(n, a2a_calls, a2a_bytes))
        comm[str(n)] = round(c, 2)
        pred[str(n)] = round(replicated + sharded / n + c, 2)
        sp[str(n)] = round(wall / pred[str(n)], 2)
    return {"mode": "gamma (plane-wave row slabs of the one k-block)", "measured_on_gpus": 1,
            "per_step_ms": {"wall": round(wall, 2), "sharded": round(sharded, 2), "replicated": round(replicated, 2)},
            "sharded_families_ms": {k_: round(v_, 2) for k_, v_ in sharded_f.items()},
            "replicated_named_ms": {k_: round(v_, 2) for k_, v_ in named.items()},
            "collectives_per_step": {"allreduce_calls": round(ar_calls + 1, 1), "allreduce_MB": round((ar_bytes + 8.0 * n_cube) / 1e6, 1),
                                     "alltoall_calls": round(a2a_calls, 1), "alltoall_MB_all_ranks": round(a2a_bytes / 1e6, 1)},
            "comm_ms_per_step": comm, "predicted_ms_per_step": pred, "predicted_speedup": sp,
            "amdahl_limit": round(wall / replicated, 2),
            "assumptions": (f"xGMI {XGMI_LINK_GBS:g} GB/s per link (SURVEY 2.4), {COLLECTIVE_LATENCY_US:g} us per collective (assumed, not "
                            "measured), ring all-reduce 2(p-1)/p bytes per link, all-to-all 1/p^2 of the block per link; no "
                            "overlap of communication with compute; per-rank kernels keep their one-GPU efficiency at 1/N "
                            "of the rows (optimistic for N = 8: tiles of 128 rows, 16 554 rows per rank)")}


def measured_slab_step(dftk, lib, basis, run, steps, base):
    """--mode gamma on ONE GPU: the row-sharded families of the timed SCF MEASURED at the size one of N ranks runs them,
    instead of dividing their one-GPU times by N:
      * every zgemm shape of the timed run (per-shape table of the library, dftk_mi_prof_zgemm_shapes) replayed with its
        long dimension (rows of an 'N' product, inner dimension of a 'C' product) cut to 1 / N, same flags, same number
        of calls per step;
      * the FFT pipeline (H psi local part) and the density pass on ceil(M / N) bands per call -- a rank transforms its
        share of the BANDS after the slab -> band transposition --, scaled to the bands per step of the timed run / N;
      * the n_G-sized element-wise kernels divided by N (streaming kernels: 133 MB per block at N = 8, far above the
        size where they lose bandwidth).
    Replicated work (eigensolver, Cholesky, host glue, cube-sized work) is taken in full from the timed run, the
    collectives from the model above.  Returns {N: predicted step ms} and the measured pieces."""
    import torch
    from dftk_jl_amd._lib import check, dftk_mi_cplx
    kpt = basis.kpoints[0]
    if not getattr(kpt, "gamma_real", False):
        return None
    dev = basis.device
    h = basis.handle
    M = int(run["info"]["psi"][0].shape[0])
    n_half = C.c_int64()
    check(lib.dftk_mi_gamma_half_size(kpt.handle, C.byref(n_half)))
    n_half = int(n_half.value)
    nmv_step = float(run["info"]["n_matvec"]) / steps
    n_occ = basis.model.n_electrons // 2
    one, zero = dftk_mi_cplx(1.0, 0.0), dftk_mi_cplx(0.0, 0.0)
    gen = torch.Generator(device=dev).manual_seed(4321)

    def fam_ms(fs):
        return sum(prof_get(lib, basis, f)[0] for f in fs)

    def time_zgemm(sh, n_div):
        long_rows = sh["trans"] == "N" and sh["m"] >= 10000
        long_k = sh["trans"] == "C" and sh["k"] >= 10000
        m, n, k = sh["m"], sh["n"], sh["k"]
        if long_rows:
            m = -(-m // n_div)
        if long_k:
            k = -(-k // n_div)
        flags = sh["flags"]                                              # UPPER / B_UPPER / REAL exactly as the timed call had them
        if sh["trans"] == "N":
            A = torch.randn((k, m, 2), dtype=torch.float64, device=dev, generator=gen)      # column-major m x k
            B = torch.randn((n, k, 2), dtype=torch.float64, device=dev, generator=gen)
            lda, ldb = m, k
        else:
            A = torch.randn((m, k, 2), dtype=torch.float64, device=dev, generator=gen)      # column-major k x m
            B = torch.randn((n, k, 2), dtype=torch.float64, device=dev, generator=gen)
            lda, ldb = k, k
        Cm = torch.empty((n, m, 2), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        reps = 3
        check(lib.dftk_mi_zgemm_ex(h, sh["trans"].encode(), m, n, k, one, A.data_ptr(), lda, B.data_ptr(), ldb, zero,
                                   Cm.data_ptr(), m, flags))                                    # warm (plan cache)
        check(lib.dftk_mi_prof_enable(h, 1))
        for _ in range(reps):
            check(lib.dftk_mi_zgemm_ex(h, sh["trans"].encode(), m, n, k, one, A.data_ptr(), lda, B.data_ptr(), ldb, zero,
                                       Cm.data_ptr(), m, flags))
        t = fam_ms((0, 11)) / reps
        check(lib.dftk_mi_prof_enable(h, 0))
        return t, bool(long_rows or long_k)

    out = {"measured_slab_step_ms": {}, "slab_pieces_ms": {}}
    psi_h = torch.randn((M, n_half, 2), dtype=torch.float64, device=dev, generator=gen)
    psi_full = torch.view_as_real(run["info"]["psi"][0]).contiguous()
    rho = torch.zeros(tuple(reversed(basis.fft_size)), dtype=torch.float64, device=dev)
    for n_div in (2, 4, 8):
        zg = 0.0
        for sh in run["zgemm_shapes"]:
            t, sharded = time_zgemm(sh, n_div)
            zg += t * sh["calls"] / steps if sharded else sh["ms"] / steps
        nb = -(-M // n_div)
        Hh = torch.empty((nb, n_half, 2), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        check(lib.dftk_mi_gamma_apply_H(kpt.handle, 3, nb, psi_h.data_ptr(), n_half, Hh.data_ptr(), n_half))   # warm
        check(lib.dftk_mi_prof_enable(h, 1))
        for _ in range(3):
            check(lib.dftk_mi_gamma_apply_H(kpt.handle, 3, nb, psi_h.data_ptr(), n_half, Hh.data_ptr(), n_half))
        t_fft_band = fam_ms((1, 2, 3, 4, 5, 15)) / 3 / nb
        check(lib.dftk_mi_prof_enable(h, 0))
        nbd = max(2, -(-n_occ // n_div))
        w = np.full(nbd, 2.0)
        check(lib.dftk_mi_density_accumulate_real(kpt.handle, nbd, psi_full.data_ptr(), int(psi_full.shape[1]), w.ctypes.data,
                                                  rho.data_ptr()))
        check(lib.dftk_mi_prof_enable(h, 1))
        for _ in range(3):
            check(lib.dftk_mi_density_accumulate_real(kpt.handle, nbd, psi_full.data_ptr(), int(psi_full.shape[1]),
                                                      w.ctypes.data, rho.data_ptr()))
        t_dens_band = fam_ms((1, 2, 6, 15)) / 3 / nbd
        check(lib.dftk_mi_prof_enable(h, 0))
        fft = t_fft_band * nmv_step / n_div
        dens = t_dens_band * n_occ / n_div
        # element-wise family of the timed run minus what the FFT replay above already contains (pack / unpack passes are
        # booked as element-wise inside gamma_apply_H): only the LOBPCG driver's share is divided
        ew_total = base["sharded_families_ms"].get(FAMILIES[15], 0.0)
        pieces = {"zgemm": round(zg, 2), "fft_pipeline": round(fft, 2), "density": round(dens, 2),
                  "elementwise_div_n": round(ew_total / n_div, 2), "bands_per_apply": nb}
        sharded = zg + fft + dens + ew_total / n_div
        out["slab_pieces_ms"][str(n_div)] = pieces
        out["measured_slab_step_ms"][str(n_div)] = round(base["per_step_ms"]["replicated"] + sharded
                                                         + base["comm_ms_per_step"][str(n_div)], 2)
    wall = base["per_step_ms"]["wall"]
    out["measured_speedup"] = {k_: round(wall / v_, 2) for k_, v_ in out["measured_slab_step_ms"].items()}
    out["note"] = ("sharded families measured at slab size on this one GPU (zgemm shapes of the timed run replayed with 1/N of "
                   "the long dimension, FFT pipeline / density on ceil(M/N) bands per call), replicated part and collectives "
                   "as in predicted_ms_per_step; pack / unpack passes inside the H psi replay are part of fft_pipeline, so "
                   "elementwise_div_n counts them a second time (conservative)")
    return out


def amdahl_kpoints(dftk, basis, model, ecut, device, run, steps, args):
    """--mode kpoints measured on ONE GPU: the step time of ONE RANK'S SHARE is measured directly -- a whole SCF on the
    first ceil(n_k / N) irreducible k-points (weights renormalised: a legitimate, smaller k-mesh problem that does all
    the replicated cube work of a step: PBE potential, LDOS pass over its k-points, Fermi level, mixing, energies) --
    so that the latency-bound lock-step batching is priced at the local k-point count, not divided by N."""
    wall = 1e3 * float(np.median(run["step_s"][2:] or run["step_s"]))
    n_k = len(basis.kcoords_global)
    n_cube = basis.N
    saved = os.environ.get("DFTK_MI_KBATCH")
    if basis.kbatch:
        os.environ["DFTK_MI_KBATCH"] = "1"        # a real N-rank run decides from the GLOBAL count (basis.py): stays batched
    share, pred, sp, comm, nloc, share_timers = {}, {}, {}, {}, {}, {}
    try:
        for n in (2, 4, 8):
            n_loc = -(-n_k // n)
            kc = [np.asarray(k_) for k_ in basis.kcoords_global[:n_loc]]
            kw = np.asarray(basis.kweights_global[:n_loc], dtype=float)
            sub = dftk.PlaneWaveBasis(model, ecut, dftk.ExplicitKpoints(kc, list(kw / kw.sum())), device=device,
                                      fft_size=basis.fft_size)
            st = dftk.ScfStepper(sub, tol=args.tol)
            ts = []
            for i in range(8):
                t0 = time.time()
                info = st.step()
                ts.append(time.time() - t0)
                if info["converged"]:
                    break
            t_share = 1e3 * float(np.median(ts[2:] or ts))
            # the phase breakdown needs a device synchronisation per phase: one more step with them switched on (not in t_share)
            st.phase_timers = True
            info = st.step()
            share_timers[str(n)] = {k_: round(1e3 * v_, 2) for k_, v_ in info["timers"].items()}
            # one density all-reduce per step (+ one for the LDOS of a metal) and the eigenvalue gather
            c = _t_allreduce_ms(n, 2 if model.temperature > 0 else 1, (2 if model.temperature > 0 else 1) * 8.0 * n_cube) \
                + COLLECTIVE_LATENCY_US * 1e-3
            nloc[str(n)], share[str(n)], comm[str(n)] = n_loc, round(t_share, 2), round(c, 3)
            pred[str(n)] = round(t_share + c, 2)
            sp[str(n)] = round(wall / pred[str(n)], 2)
            del st, sub
    finally:
        if saved is None:
            os.environ.pop("DFTK_MI_KBATCH", None)
        else:
            os.environ["DFTK_MI_KBATCH"] = saved
    timers = {k_: round(1e3 * v_ / steps, 2) for k_, v_ in run["host_timers"].items()}
    return {"mode": "kpoints (k-point sharding, one density all-reduce per step)", "measured_on_gpus": 1,
            "per_step_ms": {"wall_median_late_step": round(wall, 2), "host_timers": timers},
            "k_points": n_k, "k_points_per_rank": nloc, "measured_share_step_ms": share,
            "share_step_host_timers_ms": share_timers, "comm_ms_per_step": comm,
            "predicted_ms_per_step": pred, "predicted_speedup": sp,
            "replicated_floor_ms": share.get("8"),
            "assumptions": (f"measured_share_step_ms[N] = median late SCF step of a self-consistent run on the first ceil(n_k/N) "
                            f"irreducible k-points (renormalised weights), same cube, same functional, lock-step batching as in the "
                            f"N-rank run; + all-reduce of rho ({8.0 * n_cube / 1e6:.2f} MB) at {XGMI_LINK_GBS:g} GB/s per link and "
                            f"{COLLECTIVE_LATENCY_US:g} us per collective (assumed)")}


# ------------------------------------------------------------------------------------------ main
def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import dftk_jl_amd as dftk

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # test hooks (tests/test_gpu_multirank.py: the N > 1 code path on a ONE-GPU box): ranks share a device and
    # reduce over gloo.  Never set by the driver; the measured configuration is one rank per GPU over RCCL.
    backend = os.environ.get("DFTK_MI_BENCH_BACKEND", "nccl")
    if "DFTK_MI_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["DFTK_MI_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        comm = dftk.KptComm.from_torch()
    else:
        comm = dftk.KptComm.single()
    n_gpus = world
    device = f"cuda:{local_rank}"

    lib = dftk.load_library()
    t0 = time.time()
    model = None
    if args.mode == "kpoints":
        sym = not args.no_symmetries
        if args.system == "al":
            a = 7.6324708938577865                                       # test/testcases.jl:74
            lat = a / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
            Al = dftk.ElementPsp("Al", dftk.load_psp("Al", "pbe"))
            model = dftk.model_DFT(lat, [Al], [np.zeros(3)], functionals=("gga_x_pbe", "gga_c_pbe"), temperature=1e-3,
                                   smearing="gaussian", symmetries=sym)
            ecut, kg_n, what = args.ecut or 40.0, (args.kgrid or 12,) * 3, "Al fcc (1 atom, 3 e-) PBE HGH, Gaussian smearing T=1e-3"
        elif args.system == "si":
            lat, atoms, pos = dftk.silicon_cell()
            model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"), symmetries=sym)
            ecut, kg_n, what = args.ecut or 15.0, (args.kgrid or 4,) * 3, "Si primitive (2 atoms, 8 e-) LDA HGH"
        else:                                                            # examples/graphene.jl:15-30
            a, L = 4.66, 20.0
            lat = np.array([[a / 2, a / 2, 0.0], [-a * np.sqrt(3) / 2, a * np.sqrt(3) / 2, 0.0], [0.0, 0.0, L]])
            Cc = dftk.ElementPsp("C", dftk.load_psp("C", "pbe"))
            pos = [np.array([1 / 3, -1 / 3, 0.0]), np.array([-1 / 3, 1 / 3, 0.0])]
            model = dftk.model_DFT(lat, [Cc, Cc], pos, functionals=("gga_x_pbe", "gga_c_pbe"), temperature=1e-3,
                                   smearing="fermi_dirac", symmetries=sym)
            n_ = args.kgrid or 9
            ecut, kg_n, what = args.ecut or 40.0, (n_, n_, 1), "graphene slab (2 atoms, 8 e-) PBE HGH, Fermi-Dirac T=1e-3"
        kg = dftk.MonkhorstPack(kg_n)
        basis = dftk.PlaneWaveBasis(model, ecut, kg, device=device, comm_kpts=comm)
        n_kblocks_total = len(basis.kcoords_global)
        workload = (f"{what}, Ecut={ecut:g} Ha, fft={'x'.join(map(str, basis.fft_size))}, "
                    f"{'x'.join(map(str, kg_n))} k-mesh, {len(basis.symmetries)} symmetries -> "
                    f"{n_kblocks_total} k-points ({len(basis.kpoints)} on rank 0, "
                    f"{'lock-step batched (dftk_mi_lobpcg_multi)' if basis.kbatch else str(basis.n_lanes) + ' stream lanes'})")
        parallelism, scaling = f"kpt{n_gpus}", "strong"
    else:
        n = args.supercell
        ecut = args.ecut or 30.0
        lat, atoms, pos = dftk.silicon_cell((n, n, n))
        model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
        if args.mode == "weak" and n_gpus > 1:
            allk = dftk.MonkhorstPack((2, 2, 2)).reducible().kcoords
            kgrid = dftk.ExplicitKpoints(allk[:n_gpus], [1.0 / n_gpus] * n_gpus)
            basis = dftk.PlaneWaveBasis(model, ecut, kgrid, device=device, comm_kpts=comm)
            parallelism, scaling, ktxt = f"kpt{n_gpus}", "weak", f"{n_gpus} k-points (1 per GPU)"
        else:
            basis = dftk.PlaneWaveBasis(model, ecut, dftk.MonkhorstPack((1, 1, 1)), device=device, comm_pw=comm,
                                        gamma_real=False if args.no_gamma_real else None)
            parallelism = "single" if n_gpus == 1 else f"pw{n_gpus} (plane-wave row slabs of the one k-block)"
            scaling, ktxt = "strong", "Gamma-only"
        workload = (f"Si {n}x{n}x{n} supercell ({len(atoms)} atoms, {model.n_electrons} e-) LDA HGH, "
                    f"Ecut={ecut:g} Ha, fft={'x'.join(map(str, basis.fft_size))}, {ktxt}")
    t_setup = time.time() - t0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # what the library's communicator (the data path of every collective) really is: proves N ranks met over RCCL
    comm_info = comm.describe() if hasattr(comm, "describe") else {"n_ranks": comm.size}
    if int(comm_info.get("n_ranks", -1)) != world:
        raise SystemExit(f"rank {rank}: the library's communicator reports {comm_info} but WORLD_SIZE is {world}")
    self_check = None
    if world > 1 and args.mode == "gamma":
        self_check = sharded_self_check(dftk, basis, world)
        bad = [k_ for k_, v_ in self_check.items() if not (v_ < 1e-10)]
        if bad:
            raise SystemExit(f"sharded self-check failed on rank {rank}: {self_check}")

    if world > 1 and args.mode in ("kpoints", "weak"):
        self_check = kpoints_self_check(basis, comm, world)
        if not (self_check["allreduce_rank_stamped_cube_max_abs_error"] == 0.0):
            raise SystemExit(f"k-point self-check failed on rank {rank}: {self_check}")

    run = run_scf(dftk, lib, basis, args, barrier, world, dist, torch)
    info, elapsed, fam = run["info"], run["elapsed"], run["fam"]
    n_atoms = len(model.positions)
    parity, late_info = None, info
    if not args.no_parity:
        # UNTIMED: the timed stepper carried on to the parity tolerance (all ranks take part; its energy is what the
        # 1 / 2 / 4 / 8 GPU lines of a scaling run are compared through)
        par = continue_to_parity(run["stepper"])
        late_info = run["stepper"].info           # the converged state: what the CPU leg's "late step" starts from
        parity = {"tolerance_Ha_per_atom": PARITY_TOL_HA_PER_ATOM, "scf_tol": min(PARITY_SCF_TOL, args.tol), "n_atoms": n_atoms,
                  "timed_leg": {k_: par[k_] for k_ in ("converged", "E_total", "extra_steps", "steps_total", "drho", "wall_s")}}
    amdahl = None
    steps_run = info["n_iter"]
    n_matvec = info["n_matvec"]                  # already summed over the k-point ranks
    kblocks = n_gpus if (args.mode == "weak" and n_gpus > 1) else 1
    value = kblocks * steps_run / elapsed

    if rank == 0:
        gamma_real = bool(getattr(basis.kpoints[0], "gamma_real", False))
        roof = roofline_of(fam, n_gpus, workload)
        if args.mode == "kpoints":
            roof = latency_roofline(run, steps_run, roof)
        booked = sum(fam[f][0] for f in (0, 11, 1, 2, 3, 4, 5, 6, 7, 8, 13))
        kp0 = basis.kpoints[0]
        out = {
            "metric": "SCF iterations/sec (Hψ applies/sec) at fixed Ecut·atoms",
            "value": value, "unit": "SCF iterations/s", "n_gpus": n_gpus, "steps": steps_run,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / steps_run, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "hpsi_applies_per_s": n_matvec / elapsed,
            "config": {"workload": workload, "timed": "whole self_consistent_field (scf_full): guess_density -> "
                                                      f"tol={args.tol:g}, capped at --steps={args.steps}",
                       "converged": bool(info["converged"]), "scf_wall_s": round(elapsed, 3),
                       "scf_wall_s_to_convergence": round(elapsed, 3) if info["converged"] else None,
                       "n_G": kp0.n_G, "n_bands": int(info["psi"][0].shape[0]),
                       "n_proj": int(basis.terms.D.shape[0]) if basis.terms.D is not None else 0,
                       "parallelism": parallelism, "rccl": comm_info, "sharded_self_check": self_check,
                       "setup_s": round(t_setup, 2),
                       "orbitals": ("real-symmetric at Gamma (psi(-G) = conj psi(G)): half-sphere real GEMMs, two bands "
                                    "per FFT pass; same eigenvalues / density / energies as the reference's complex "
                                    "iteration, which is timed in the same run: config.complex_iteration")
                       if gamma_real else "general complex",
                       "n_matvec": int(n_matvec), "lobpcg_iters_per_step": run["iters"],
                       "n_matvec_per_step": run["nmv_steps"],
                       "diagtol_per_step": [float(f"{d:.3g}") for d in run["diagtols"]],
                       "step_wall_s": [round(s_, 3) for s_ in run["step_s"]],
                       "host_timers_ms_per_step": {k_: round(1e3 * v_ / steps_run, 2)
                                                   for k_, v_ in run["host_timers"].items()},
                       "host_timers_synced": args.mode != "kpoints",   # (k-point steps: no per-phase device synchronisation)
                       "library_booked_ms": round(booked, 1), "lib_hash": library_source_hash(),
                       "E_total": info["energies"].total, "drho": info["history_drho"][-1],
                       "library_launches_per_step": round(run["library_launches"] / steps_run, 1),
                       "library_host_syncs_per_step": round(run["library_host_syncs"] / steps_run, 1)},
            "roofline": roof,
        }
        # flat scalars (the driver's BENCH record keeps scalars of `config`, not nested objects)
        out["config"].update(step_roofline(fam, elapsed, steps_run, run["step_s"], run["iters"]))
        out["config"]["roofline_frac_zgemm"] = round(roof["frac"], 4) if roof.get("bound") == "mfma" else None
        out["config"]["heev_ms_per_step"] = round(fam[7][0] / steps_run, 2)
    # ---- the same iteration WITHOUT the two-level start (random orbitals in the first diagonalisation, as the reference and as
    # rounds 1-5 of this repo): same basis, same run, same box
    if (world == 1 and args.mode == "gamma" and not args.no_random_start_leg and getattr(basis, "coarse", None) is not None):
        rrun = run_scf(dftk, lib, basis, args, barrier, world, dist, torch, coarse_start=False)
        ri = rrun["info"]
        rrun.pop("stepper", None)
        if rank == 0:
            out["random_start_value"] = round(ri["n_iter"] / rrun["elapsed"], 4)
            out["random_start_unit"] = "SCF iterations/s"
            out["config"]["random_start_value"] = out["random_start_value"]
            out["config"]["random_start"] = {
                "what": "the same self_consistent_field with the first diagonalisation started from random orbitals (the "
                        "reference's start; rounds 1-5 of this repo) instead of the two-level start",
                "value": ri["n_iter"] / rrun["elapsed"], "steps": int(ri["n_iter"]), "converged": bool(ri["converged"]),
                "scf_wall_s": round(rrun["elapsed"], 3), "E_total": ri["energies"].total,
                "lobpcg_iters_per_step": rrun["iters"], "step_wall_s": [round(s_, 3) for s_ in rrun["step_s"]],
                "n_matvec": int(ri["n_matvec"])}
            out["config"]["first_step_start"] = (
                "two-level: LOBPCG from random orbitals on the companion basis at Ecut / 4 (its H psi applies: "
                f"{run['n_matvec_coarse']}, counted in the timed region but not in n_matvec), zero-padded eigenvectors as start "
                "vectors; the companion basis is built with the basis (config.setup_s)")
        del rrun, ri
        torch.cuda.empty_cache()
    # ---- the reference's own iteration (general complex orbitals) on the same cell, same run, same box
    if (world == 1 and args.mode == "gamma" and not args.no_gamma_real and not args.no_complex_leg
            and bool(getattr(basis.kpoints[0], "gamma_real", False))):
        cbasis = dftk.PlaneWaveBasis(model, ecut, dftk.MonkhorstPack((1, 1, 1)), device=device, gamma_real=False,
                                     coarse_start=False)       # the reference's algorithm in full: random start orbitals
        run.pop("stepper", None)                     # (the real leg's orbitals and LOBPCG workspace are not needed any more)
        torch.cuda.empty_cache()
        crun = run_scf(dftk, lib, cbasis, args, barrier, world, dist, torch, coarse_start=False)
        ci = crun["info"]
        if parity is not None:
            cpar = continue_to_parity(crun["stepper"])
            parity["complex_leg"] = {k_: cpar[k_] for k_ in ("converged", "E_total", "extra_steps", "steps_total", "drho", "wall_s")}
            parity["dE_total_real_vs_complex_per_atom"] = (par["E_total"] - cpar["E_total"]) / n_atoms
            parity["max_deigenvalue_real_vs_complex"] = float(max(
                np.max(np.abs(a_ - b_[:len(a_)])) for a_, b_ in zip(par["eigenvalues"], cpar["eigenvalues"])))
        crun.pop("stepper", None)
        croof = roofline_of(crun["fam"], 1, workload)
        out["config"]["complex_iteration"] = {
            "what": "the same self_consistent_field with gamma_real=False: LOBPCG on general complex orbitals exactly as "
                    "the reference iterates them (3M complex zgemm, one band per FFT pass)",
            "value": ci["n_iter"] / crun["elapsed"], "unit": "SCF iterations/s", "steps": ci["n_iter"],
            "converged": bool(ci["converged"]), "scf_wall_s": round(crun["elapsed"], 3),
            "scf_wall_s_to_convergence": round(crun["elapsed"], 3) if ci["converged"] else None,
            "hpsi_applies_per_s": ci["n_matvec"] / crun["elapsed"], "n_matvec": int(ci["n_matvec"]),
            "E_total": ci["energies"].total,
            # (meaningful when both runs converged; capped runs stop at different points of their trajectories)
            "dE_total_vs_real": (ci["energies"].total - info["energies"].total)
            if (ci["converged"] and info["converged"]) else None,
            "lobpcg_iters_per_step": crun["iters"], "step_wall_s": [round(s_, 3) for s_ in crun["step_s"]],
            "roofline": {k_: croof[k_] for k_ in ("bound", "achieved", "peak", "unit", "frac", "achieved_unstructured",
                                                  "mfma_busy_frac", "kernel", "launches", "avg_launch_ms", "families_ms",
                                                  "families_rate") if k_ in croof}}
        out["config"]["complex_iteration_value"] = round(ci["n_iter"] / crun["elapsed"], 4)
        # top level, next to `value`: the like-for-like number for the REFERENCE's own algorithm (general complex orbitals);
        # `value` is the real-symmetric Gamma iteration, an extension the reference does not have
        out["complex_iteration_value"] = round(ci["n_iter"] / crun["elapsed"], 4)
        out["complex_iteration_unit"] = "SCF iterations/s"
        out["value_algorithm"] = ("real-symmetric Gamma orbitals" + (" + two-level start of the first diagonalisation" if
                                  getattr(basis, "coarse", None) is not None else "") + " (extensions); random_start_value = "
                                  "the same from random orbitals; complex_iteration_value = the reference's iteration "
                                  "(general complex orbitals, random start)")
        out["config"]["complex_iteration_frac"] = round(croof["frac"], 4)
        out["config"]["complex_iteration_steps"] = int(ci["n_iter"])
        del crun, ci, cbasis
        torch.cuda.empty_cache()

    run.pop("stepper", None)
    torch.cuda.empty_cache()
    # ---- parity against the ORACLE's golden fixture (untimed), Amdahl model (measured on this one GPU)
    if (world == 1 and parity is not None and args.mode == "gamma" and args.supercell in GOLDEN and ecut == 30.0
            and not args.no_gamma_real):
        try:
            parity["golden"] = golden_leg(dftk, model, ecut, args.supercell, device)
        except Exception as e:       # a leg that cannot run is a FAILED check (below), never a lost measurement
            parity["golden"] = {"error": repr(e), "converged": False, "dE_total_vs_golden_per_atom": 1e300,
                                "max_deigenvalue_vs_golden": 1e300}       # (finite: the line must stay strict JSON)
        torch.cuda.empty_cache()
    try:
        if world == 1 and args.mode == "gamma":
            amdahl = amdahl_gamma(run, steps_run, basis.N)
            if not args.no_amdahl_probe:
                try:
                    slab = measured_slab_step(dftk, lib, basis, run, steps_run, amdahl)
                    if slab is not None:
                        amdahl.update(slab)
                except Exception as e:
                    amdahl["measured_slab_step_ms"] = {"error": repr(e)}
        elif world == 1 and args.mode == "kpoints" and not args.no_amdahl_probe:
            amdahl = amdahl_kpoints(dftk, basis, model, ecut, device, run, steps_run, args)
    except Exception as e:           # the model is reporting only
        amdahl = {"error": repr(e)}
    parity_failed = []
    if rank == 0:
        if parity is not None:
            checks = {"timed_leg converged": parity["timed_leg"]["converged"]}
            if "complex_leg" in parity:
                checks["complex_leg converged"] = parity["complex_leg"]["converged"]
                checks["|dE real vs complex| per atom"] = abs(parity["dE_total_real_vs_complex_per_atom"]) < PARITY_TOL_HA_PER_ATOM
                checks["eigenvalues real vs complex"] = parity["max_deigenvalue_real_vs_complex"] < 1e-7
            wkey = workload.split(", Gamma-only")[0] if args.mode == "gamma" else workload.split(" k-points (")[0]
            parity["workload_key"] = wkey
            ref_e = one_rank_energy(wkey)
            if n_gpus > 1 and ref_e is not None:
                # the N-rank run against the ONE-rank run of the same workload (sharding must not move the fixed point)
                parity["E_total_one_rank"] = ref_e["E_total"]
                parity["dE_total_vs_one_rank_per_atom"] = (parity["timed_leg"]["E_total"] - ref_e["E_total"]) / n_atoms
                checks["|dE N ranks vs one rank| per atom"] = abs(parity["dE_total_vs_one_rank_per_atom"]) < PARITY_TOL_HA_PER_ATOM
                # a device-vs-device fixture: say which build wrote it (a stale entry is reported, not hidden)
                parity["one_rank_reference_lib_hash"] = ref_e.get("lib_hash")
                parity["one_rank_reference_is_this_build"] = ref_e.get("lib_hash") == library_source_hash()
            elif n_gpus > 1:
                parity["one_rank_reference"] = "MISSING"
                print(f"bench.py: WARNING: no one-rank reference energy for workload key {wkey!r} in {ONE_RANK_ENERGIES}: the "
                      f"N = {n_gpus} run is NOT compared with the one-rank run (write it with --gpus 1 --write-reference-energy)",
                      file=sys.stderr, flush=True)
            elif n_gpus == 1 and args.write_reference_energy and parity["timed_leg"]["converged"]:
                try:
                    with open(ONE_RANK_ENERGIES) as fh:
                        table = json.load(fh)
                except (OSError, ValueError):
                    table = {}
                table[wkey] = {"E_total": parity["timed_leg"]["E_total"], "scf_tol": parity["scf_tol"], "n_atoms": n_atoms,
                               "lib_hash": library_source_hash()}
                with open(ONE_RANK_ENERGIES, "w") as fh:
                    json.dump(table, fh, indent=1, sort_keys=True)
            if "golden" in parity:
                checks["golden leg converged"] = parity["golden"]["converged"]
                checks["|dE vs golden| per atom"] = abs(parity["golden"]["dE_total_vs_golden_per_atom"]) < PARITY_TOL_HA_PER_ATOM
                checks["eigenvalues vs golden"] = parity["golden"]["max_deigenvalue_vs_golden"] < 1e-7
            parity_failed = [k_ for k_, ok_ in checks.items() if not ok_]
            parity["checks"] = {k_: bool(v_) for k_, v_ in checks.items()}
            parity["pass"] = not parity_failed
            parity["note"] = ("untimed legs run after the timed region: the timed steppers continued to scf_tol (both the "
                              "real-symmetric and the reference's complex iteration, production cube), and -- N = 1 -- a whole SCF at "
                              "the k-mesh-compatible cube against the oracle fixture; the run exits non-zero when a check fails")
        out["config"]["parity"] = parity
        if parity is not None:
            out["config"]["parity_pass"] = bool(parity["pass"])
            out["config"]["parity_E_total_converged"] = parity["timed_leg"]["E_total"]
            out["config"]["parity_dE_real_vs_complex_per_atom"] = parity.get("dE_total_real_vs_complex_per_atom")
            out["config"]["parity_max_deigenvalue_real_vs_complex"] = parity.get("max_deigenvalue_real_vs_complex")
            if "golden" in parity:
                out["config"]["parity_dE_vs_golden_per_atom"] = parity["golden"]["dE_total_vs_golden_per_atom"]
                out["config"]["parity_max_deigenvalue_vs_golden"] = parity["golden"]["max_deigenvalue_vs_golden"]
            if "dE_total_vs_one_rank_per_atom" in parity:
                out["config"]["parity_dE_vs_one_rank_per_atom"] = parity["dE_total_vs_one_rank_per_atom"]
        out["amdahl"] = amdahl
        if isinstance(amdahl, dict) and "predicted_speedup" in amdahl:
            for n_, v_ in amdahl["predicted_speedup"].items():
                out["config"][f"amdahl_predicted_speedup_{n_}gpu"] = v_
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                if args.mode == "kpoints":
                    out["cpu_baseline"] = {"value": None, "unit": "SCF iterations/s", "cores": os.cpu_count(),
                                           "kind": "port", "sample": "see cfg1_scf_3steps"}
                else:
                    # the CPU leg is the REFERENCE's iteration: general complex orbitals, complex zgemm flops
                    per_step = {"n_matvec": n_matvec / steps_run, "zgemm_flops": fam[14][1] / steps_run}
                    n_smp = args.cpu_sample_bands or min(os.cpu_count(), 256)
                    model_leg = cpu_baseline_gamma(basis, late_info, n_smp, per_step)
                    out["cpu_baseline"] = model_leg
                    # a REAL timed step when the sampled model says it fits the budget
                    # tolerance of a TYPICAL late step: the median diagtol of the timed run (from the converged
                    # orbitals LOBPCG then needs the one iteration that most steps of this SCF take)
                    tol_mid = float(np.median(run["diagtols"][2:] or run["diagtols"]))
                    t_dev, dinfo = device_late_step(dftk, basis, late_info, args.tol, tol_mid)
                    late_model = (2 * late_info["psi"][0].shape[0] / model_leg["hpsi_applies_per_s"]
                                  + model_leg["model_terms"]["density_s_per_band"] * (basis.model.n_electrons // 2)
                                  + model_leg["model_terms"]["late_step_zgemm_s"])
                    timed = None
                    if late_model < args.cpu_step_budget:
                        timed = cpu_timed_late_step(basis, late_info, float(dinfo["diagtol"]), args.cpu_step_budget, dinfo,
                                                    threads_hint=model_leg.get("threads"))
                    if timed is not None:
                        timed["device_step_s"] = round(t_dev, 4)
                        timed["device_lobpcg_iterations"] = float(np.mean(dinfo["diagonalization"]["n_iter"]))
                        timed["speedup_same_step"] = round(timed["cpu_step_s"] / t_dev, 1)
                        out["cpu_baseline"] = {
                            "value": 1.0 / timed["cpu_step_s"], "unit": "SCF iterations/s", "cores": os.cpu_count(),
                            "kind": "port, timed step",
                            "sample": (f"ONE real late SCF step of this workload on the host, wall-clocked: the oracle's LOBPCG "
                                       f"(general complex orbitals, restatement of lobpcg_hyper_impl.jl) from the converged "
                                       f"orbitals at diagtol={timed['diagtol']:.2g} -> {timed['lobpcg_iterations']} iteration(s), "
                                       f"{timed['n_matvec']} H psi applies ({timed['cpu_lobpcg_s']} s) + compute_density "
                                       f"({timed['cpu_density_s']} s); H psi / density band-parallel on {timed['fft_threads']} "
                                       f"threads (pocketfft workers=1 each), dense algebra = NumPy OpenBLAS "
                                       f"({timed['blas_threads']} threads) on panels of the true width; the device ran the "
                                       f"same step in {t_dev:.3f} s.  Not DFTK (no Julia in the image)"),
                            "timed_step": timed, "sampled_model": model_leg}
                    else:
                        out["cpu_baseline"]["timed_step"] = (f"skipped: the sampled model predicts {late_model:.0f} s "
                                                             f"(budget {args.cpu_step_budget:.0f} s) or host memory is short")
                        out["cpu_baseline"]["device_late_step_s"] = round(t_dev, 4)
                out["cpu_baseline"]["cfg1_scf_3steps"] = cfg1_scf_3steps(device)
            except Exception as e:  # the baseline is reporting only; never lose the measurement
                out["cpu_baseline"] = {"value": None, "unit": "SCF iterations/s", "cores": os.cpu_count(),
                                       "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit(f"PARITY FAILED: {parity_failed} (config.parity of the JSON line above)")


if __name__ == "__main__":
    main()
