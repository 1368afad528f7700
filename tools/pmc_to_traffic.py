#!/usr/bin/env python
"""Turn the two PMC passes of tools/pmc_traffic_bench.sh into profiles/r02_pmc_traffic.json.

usage: pmc_to_traffic.py fetch_counter_collection.csv write_counter_collection.csv bench_line.json out.json

Kernels are grouped into the library's profiling families by name; bytes per launch = family bytes / launches the
library counted in the SAME whole-process pass (bench.py --prof-all; a zgemm call = one launch).
FETCH_SIZE is doubled (gfx950 under-reports 16 B/lane streaming reads by 2x: /opt/skills/guides/MI355X_MICROARCH.md
"HBM", calibrated with a 1 GiB device copy by tools/pmc_traffic.sh); WRITE_SIZE is taken as is.  CSV units: KiB.
The output carries the bench line's workload string and library source hash: bench.py only quotes it as
``roofline.traffic`` for exactly that build and workload."""
import collections
import csv
import json
import sys

FAMILY_OF = [("k_zgemm", "zgemm_f64_mfma"), ("k_xbwd_scatter", "fft_A_xbwd_scatter"), ("k_ybwd", "fft_B_ybwd"),
             ("k_zpass<0", "fft_C_z_fused_V"), ("k_zpass_reg", "fft_C_z_fused_V"), ("k_yfwd", "fft_D_yfwd"), ("k_xfwd_gather", "fft_E_xfwd_gather"),
             ("k_zdensity", "density_z"), ("k_jacobi", "heev_jacobi")]


def family(name):
    for key, fam in FAMILY_OF:
        if key in name:
            return fam
    return None


def totals(path, counter):
    per_dispatch = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per_dispatch[(r["Dispatch_Id"], r["Kernel_Name"])] = float(r["Counter_Value"])
    out, n = collections.defaultdict(float), collections.defaultdict(int)
    for (_, name), v in per_dispatch.items():
        f = family(name)
        if f and "k_zgemm_reduce" not in name:
            out[f] += v
            n[f] += 1
        elif f:
            out[f] += v          # the split-K reduction belongs to its zgemm call, not a launch of its own
    return out, n


def main():
    (fetch, nf), (write, nw) = totals(sys.argv[1], "FETCH_SIZE"), totals(sys.argv[2], "WRITE_SIZE")
    line = json.loads([ln for ln in open(sys.argv[3]).read().strip().splitlines() if ln.startswith("{")][-1])
    roof = line["roofline"]
    work, launches = roof.get("families_work", {}), roof.get("families_launches", {})
    fams = {}
    for f in sorted(set(fetch) | set(write)):
        # launches as the LIBRARY counts them in the same whole-process pass (bench.py --prof-all): one per FFT-stage
        # launch, one per zgemm CALL (interior + border + split-K reduction kernels together)
        n = launches.get(f, 0) + (launches.get("zgemm_f64_mfma_structured", 0) if f == "zgemm_f64_mfma" else 0)
        if not n:
            continue
        fams[f] = {"fetch_KiB_raw": fetch[f], "write_KiB_raw": write[f], "kernel_dispatches_fetch_pass": nf[f],
                   "kernel_dispatches_write_pass": nw[f], "launches": n,
                   "bytes_per_launch": (2.0 * fetch[f] + write[f]) * 1024 / n}
        if f == "zgemm_f64_mfma":
            fams[f]["algorithmic_bytes_per_launch"] = work.get("zgemm_operand_bytes", 0.0) / n
        elif f in work:
            fams[f]["algorithmic_bytes_per_launch"] = work[f] / n
        fams[f]["ratio"] = fams[f]["bytes_per_launch"] / max(fams[f].get("algorithmic_bytes_per_launch", 0.0), 1e-300)
    json.dump({"workload": line["config"]["workload"], "lib_hash": line["config"]["lib_hash"],
               "collected": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `python bench.py "
                            "--no-cpu-baseline --prof-all` (tools/pmc_traffic_bench.sh)" + __import__("os").environ.get("PMC_NOTE", ""),
               "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE x1, KiB -> B; per launch = family bytes / family "
                             "dispatches in the same pass",
               "families": fams}, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
