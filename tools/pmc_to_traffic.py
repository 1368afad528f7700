#!/usr/bin/env python
"""Turn the two PMC passes of tools/pmc_traffic_bench.sh into profiles/r01_pmc_traffic.json.

usage: pmc_to_traffic.py fetch_counter_collection.csv write_counter_collection.csv bench_line.json out.json

Kernels are grouped into the library's profiling families by name; FETCH_SIZE is doubled (gfx950
under-reports 16 B/lane streaming reads by 2x: calibrated with a 1 GiB device copy, tools/pmc_traffic.sh,
and /opt/skills/guides/MI355X_MICROARCH.md "HBM"); WRITE_SIZE is taken as is.  Units in the CSV: KiB."""
import collections
import csv
import json
import sys

FAMILY_OF = [("k_zgemm", "zgemm_f64_mfma"), ("k_xbwd_scatter", "fft_A_xbwd_scatter"), ("k_ybwd", "fft_B_ybwd"),
             ("k_zpass<0", "fft_C_z_fused_V"), ("k_yfwd", "fft_D_yfwd"), ("k_xfwd_gather", "fft_E_xfwd_gather"),
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
    out = collections.defaultdict(float)
    for (_, name), v in per_dispatch.items():
        f = family(name)
        if f:
            out[f] += v
    return out


def main():
    fetch, write = totals(sys.argv[1], "FETCH_SIZE"), totals(sys.argv[2], "WRITE_SIZE")
    line = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    launches = line["roofline"]["families_launches"]
    work = line["roofline"].get("families_work", {})
    fams = {}
    for f in sorted(set(fetch) | set(write)):
        n = launches.get(f, 0)
        if not n:
            continue
        b = 2.0 * fetch[f] * 1024 + write[f] * 1024
        fams[f] = {"fetch_KiB_raw": fetch[f], "write_KiB_raw": write[f], "hbm_bytes": b, "launches": n,
                   "bytes_per_launch": b / n}
        if f in work and f != "zgemm_f64_mfma":
            fams[f]["algorithmic_bytes_per_launch"] = work[f] / n
        elif f == "zgemm_f64_mfma" and line["roofline"].get("kernel") == f:
            # operand bytes per launch of THIS pass (bench.py --prof-all counts from the first warm-up step on)
            fams[f]["algorithmic_bytes_per_launch"] = line["roofline"].get("algorithmic_bytes_per_launch")
    json.dump({"workload": sys.argv[5] if len(sys.argv) > 5 else "si4x4x4_ecut30",
               "collected": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, python bench.py --prof-all",
               "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE x1, KiB -> B", "families": fams},
              open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
