#!/usr/bin/env python
"""bench.py -- SCF-iteration throughput of the MI355X plane-wave hot path (BASELINE.json metric).

Workload (``config.workload``): fcc silicon n x n x n supercell (default 4x4x4 = 128 atoms, BASELINE
configs[1]), LDA (lda_x + lda_c_pw), HGH pseudopotential, Ecut = 30 Ha, FFT cube from
``compute_fft_size`` (150^3), fp64.  One "step" = one SCF iteration of DFTK's
``self_consistent_field``: build V = V_loc + V_H + V_xc, LOBPCG (AdaptiveDiagtol / AdaptiveBands)
with H psi on the device, ``compute_density`` (+ RCCL all-reduce), energies, Anderson mixing.
N = 1: Gamma only (exactly configs[1]).  N > 1: one k-point per GPU (the first N points of the
unshifted 2x2x2 Monkhorst-Pack mesh, weight 1/N each) so that per-GPU work is fixed (weak
scaling) and the only data-path collective is the density all-reduce; ``value`` then counts
k-block SCF iterations per second summed over ranks.

Prints ONE JSON line on rank 0 (contract in the task statement) with ``roofline`` (dominant
kernel family, HIP-event timed inside the library on its own stream) and ``cpu_baseline`` (the
CPU oracle timed on a bounded sample of the same workload on this host's cores).
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
            5: "fft_E_xfwd_gather", 6: "density_z", 7: "heev_jacobi", 8: "potrf_trtri", 9: "apply_H_total"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--supercell", type=int, default=4, help="n for the n x n x n Si supercell (4 = configs[1])")
    ap.add_argument("--ecut", type=float, default=30.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-bands", type=int, default=192)
    ap.add_argument("--prof-all", action="store_true",
                    help="count kernel-family launches from the first warm-up step on (to line the counts up with a "
                         "whole-process rocprofv3 --pmc run; see tools/pmc_traffic_bench.sh)")
    return ap.parse_args()


def prof_get(lib, basis, fam):
    ms, work, n = C.c_double(), C.c_double(), C.c_int64()
    from dftk_jl_amd._lib import check
    check(lib.dftk_mi_prof_get(basis.handle, fam, C.byref(ms), C.byref(work), C.byref(n)))
    return ms.value, work.value, n.value


def cpu_baseline(basis, info, n_sample, n_lobpcg_iters_per_step, n_matvec_per_step):
    """Time the NumPy/SciPy oracle (kind = "port") on a bounded sample of the same workload and
    scale to SCF iterations/s with the per-step operation counts measured on the device run."""
    import oracle
    from oracle.terms import HamiltonianBlock
    t_all = time.time()
    kpt = basis.kpoints[0]
    T = basis.terms

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
    H = HamiltonianBlock(ob, okpt, kpt.kinetic.cpu().numpy(), V, P, T.D)
    rng = np.random.default_rng(0)
    n_G, M = kpt.n_G, info["psi"][0].shape[0]
    psi = rng.standard_normal((n_G, n_sample)) + 1j * rng.standard_normal((n_G, n_sample))
    H.mul(psi[:, :1])                                              # warm FFT plans / BLAS threads
    t0 = time.time()
    H.mul(psi)
    t_hpsi = (time.time() - t0) / n_sample
    t0 = time.time()
    for n in range(n_sample):
        np.abs(ob.ifft(okpt, psi[:, n], normalize=False)) ** 2
    t_dens = (time.time() - t0) / n_sample
    # dense algebra rate: Gram matrix and rotation of a 64-column panel
    mcols = 64
    Xs = rng.standard_normal((n_G, mcols)) + 1j * rng.standard_normal((n_G, mcols))
    t0 = time.time()
    G = Xs.conj().T @ Xs
    Xs @ G
    t_blas = time.time() - t0
    rate = 2 * 8.0 * n_G * mcols * mcols / t_blas                  # flop/s of zgemm on this host
    flops_dense = 224.0 * n_G * M * M                              # SURVEY section 8(d), Unit C
    n_occ = basis.model.n_electrons // 2
    t_step = n_matvec_per_step * t_hpsi + n_lobpcg_iters_per_step * flops_dense / rate + n_occ * t_dens
    return {"value": 1.0 / t_step, "unit": "SCF iterations/s", "cores": os.cpu_count(), "kind": "port",
            "hpsi_applies_per_s": 1.0 / t_hpsi,
            "sample": (f"oracle (NumPy/SciPy restatement, scipy.fft workers=all, OpenBLAS threads=all) timed on "
                       f"{n_sample} bands of H psi ({t_hpsi * 1e3:.1f} ms/band), {n_sample} density bands "
                       f"({t_dens * 1e3:.1f} ms/band) and a {mcols}-column zgemm panel ({rate / 1e9:.1f} GF/s); "
                       f"scaled with the device run's per-step counts (n_matvec={n_matvec_per_step:.0f}, "
                       f"LOBPCG iterations={n_lobpcg_iters_per_step:.1f}, dense flops/iter=224 n_G M^2); "
                       f"sample wall {time.time() - t_all:.1f} s")}


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

    lib = dftk.load_library()
    n = args.supercell
    lat, atoms, pos = dftk.silicon_cell((n, n, n))
    model = dftk.model_DFT(lat, atoms, pos, functionals=("lda_x", "lda_c_pw"))
    allk = dftk.MonkhorstPack((2, 2, 2)).reducible().kcoords
    kgrid = dftk.ExplicitKpoints(allk[:n_gpus], [1.0 / n_gpus] * n_gpus)
    t0 = time.time()
    basis = dftk.PlaneWaveBasis(model, args.ecut, kgrid, device=f"cuda:{local_rank}", comm_kpts=comm)
    t_setup = time.time() - t0
    stepper = dftk.ScfStepper(basis, tol=1e-6)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from dftk_jl_amd._lib import check
    if args.prof_all:
        check(lib.dftk_mi_prof_enable(basis.handle, 1))
    for _ in range(args.warmup):
        stepper.step()
    barrier()
    if not args.prof_all:
        check(lib.dftk_mi_prof_enable(basis.handle, 1))
    nmv0 = stepper.info["n_matvec"]
    iters = []
    host_timers = {}
    t0 = time.time()
    for _ in range(args.steps):
        info = stepper.step()
        iters.append(float(np.mean(info["diagonalization"]["n_iter"])))
        for k_, v_ in info["timers"].items():
            host_timers[k_] = host_timers.get(k_, 0.0) + v_
    barrier()
    elapsed = time.time() - t0
    check(lib.dftk_mi_prof_enable(basis.handle, 0))
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    n_matvec = info["n_matvec"] - nmv0          # already summed over ranks
    kblocks = n_gpus                             # one k-block per rank
    value = kblocks * args.steps / elapsed

    if rank == 0:
        fam = {f: prof_get(lib, basis, f) for f in FAMILIES}
        kernel_fams = [f for f in range(0, 7) if fam[f][2] > 0]
        dom = max(kernel_fams, key=lambda f: fam[f][0])
        ms, work, launches = fam[dom]
        if dom == 0:
            roof = {"bound": "mfma", "achieved": work / (ms * 1e-3) / 1e12, "peak": F64_MFMA_PEAK_TF,
                    "unit": "TFLOP/s"}
        else:
            roof = {"bound": "hbm", "achieved": work / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        roof["frac"] = roof["achieved"] / roof["peak"]
        if dom == 0:
            roof["note"] = ("achieved = algorithmic 8mnk flop / time; full tiles run the 3M (Karatsuba) complex product, "
                            "i.e. 6mnk executed MFMA flop, so the executed-flop rate of those launches is 3/4 of it "
                            "(an MFMA-saturated 3M kernel would read 104.8 TFLOP/s here); peak = dense f64 MFMA spec")
        roof["traffic"] = None
        # HBM bytes per launch of the dominant family from the committed PMC passes (rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE over this same command, tools/pmc_traffic_bench.sh): the counters
        # cannot be read from inside the process, so the last measured value is reported with its source
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")) as fh:
                pmc = json.load(fh)
            if FAMILIES[dom] in pmc["families"] and pmc["workload"] == f"si{n}x{n}x{n}_ecut{args.ecut:g}":
                roof["traffic"] = pmc["families"][FAMILIES[dom]]["bytes_per_launch"]
                roof["traffic_unit"] = "B/launch"
                roof["traffic_source"] = ("profiles/r01_pmc_traffic.json (" + pmc["collected"] + ")"
                                          + ("; " + pmc["note"] if pmc.get("note") else ""))
                roof["algorithmic_bytes_per_launch"] = (prof_get(lib, basis, 10)[1] / max(launches, 1) if dom == 0
                                                        else work / max(launches, 1))
        except (OSError, KeyError, ValueError):
            pass
        roof["kernel"] = FAMILIES[dom]
        roof["launches"] = launches
        roof["avg_launch_ms"] = ms / max(launches, 1)
        roof["families_ms"] = {FAMILIES[f]: round(fam[f][0], 3) for f in FAMILIES}
        roof["families_launches"] = {FAMILIES[f]: int(fam[f][2]) for f in FAMILIES}
        roof["families_work"] = {FAMILIES[f]: fam[f][1] for f in FAMILIES}   # flops (zgemm) / algorithmic bytes (FFT)
        roof["families_rate"] = {
            FAMILIES[f]: (round(fam[f][1] / (fam[f][0] * 1e-3) / (1e12 if f == 0 else 1e9), 2)
                          if fam[f][0] > 0 and f < 7 else None) for f in FAMILIES}
        out = {
            "metric": "SCF iterations/sec (Hψ applies/sec) at fixed Ecut·atoms",
            "value": value, "unit": "SCF iterations/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "hpsi_applies_per_s": n_matvec / elapsed,
            "config": {"workload": f"Si {n}x{n}x{n} supercell ({len(atoms)} atoms, {model.n_electrons} e-) LDA "
                                   f"HGH, Ecut={args.ecut:g} Ha, fft={'x'.join(map(str, basis.fft_size))}, "
                                   f"{'Gamma-only' if n_gpus == 1 else f'{n_gpus} k-points (1 per GPU)'}",
                       "n_G": basis.kpoints[0].n_G, "n_bands": int(info["psi"][0].shape[0]),
                       "n_proj": int(basis.terms.D.shape[0]) if basis.terms.D is not None else 0,
                       "parallelism": f"kpt{n_gpus}", "setup_s": round(t_setup, 2),
                       "lobpcg_iters_per_step": iters,
                       "host_timers_ms_per_step": {k_: round(1e3 * v_ / args.steps, 2) for k_, v_ in host_timers.items()}, "E_total": info["energies"].total,
                       "drho": info["history_drho"][-1]},
            "roofline": roof,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(basis, info, args.cpu_sample_bands, float(np.mean(iters)),
                                                   n_matvec / args.steps)
            except Exception as e:  # the baseline is reporting only; never lose the measurement
                out["cpu_baseline"] = {"value": None, "unit": "SCF iterations/s", "cores": os.cpu_count(),
                                       "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
