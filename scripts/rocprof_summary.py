#!/usr/bin/env python
"""Condense the rocprofv3 outputs of scripts/profile_bench.sh into the small CSVs kept under profiles/.

  python scripts/rocprof_summary.py gpurun_out profiles/r01_v2

writes <prefix>_kernel_stats.csv  (rocprofv3 --kernel-trace --stats: calls, total, average per kernel) and
       <prefix>_hbm_pmc.csv       (FETCH_SIZE / WRITE_SIZE per dispatch, separate --pmc passes, in bytes after the
                                   unit (KB) and gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md,
                                   calibrated on k_h_av whose traffic is known exactly).
"""
import collections
import csv
import json
import re
import sys


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    return n.strip()


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    ni, nj, nk = (int(x) for x in (sys.argv[3:6] if len(sys.argv) > 5 else (1440, 1080, 75)))
    import os
    stats = f"{src}/prof_stats/stats_kernel_stats.csv"
    rows = list(csv.DictReader(open(stats))) if os.path.exists(stats) else []   # (PMC-only re-runs keep the old stats file)
    with open(f"{prefix}_kernel_stats.csv", "w") if rows else open(os.devnull, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,percent\n")
        for r in rows:
            n = short(r["Name"])
            if not n.startswith("k_"):
                continue   # torch set-up kernels (synthetic state), fills and copies
            f.write(f"\"{n}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e6:.3f},{float(r['AverageNs']) / 1e3:.1f},{r['Percentage']}\n")
    agg = {}
    for tag in ("fetch", "write"):
        a = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f"{src}/prof_{tag}/{tag}_counter_collection.csv")):
            n = short(r["Kernel_Name"])
            a[n][0] += 1
            a[n][1] += float(r["Counter_Value"]) * 1024.0     # FETCH_SIZE / WRITE_SIZE are in KB
        agg[tag] = a
    # calibration on k_h_av: per step three calls on (ni+4)x(nj+4)xnk cells reading 2+1+2 and writing 1+1+1 words
    cells = (ni + 4) * (nj + 4) * nk * 8.0
    n_hav = agg["fetch"]["k_h_av"][0]
    exp_fetch = cells * 5.0 / 3.0
    exp_write = cells * 1.0
    cal_f = exp_fetch / (agg["fetch"]["k_h_av"][1] / n_hav)
    cal_w = exp_write / (agg["write"]["k_h_av"][1] / agg["write"]["k_h_av"][0])
    with open(f"{prefix}_hbm_pmc.csv", "w") as f:
        f.write(f"# FETCH_SIZE x{cal_f:.3f}, WRITE_SIZE x{cal_w:.3f} (calibration on k_h_av: known {exp_fetch / 1e6:.1f} MB read, "
                f"{exp_write / 1e6:.1f} MB written per launch on average)\n")
        f.write("kernel,dispatches,fetch_bytes_per_launch,write_bytes_per_launch,total_MB_per_launch\n")
        names = sorted((n for n in agg["fetch"] if n.startswith("k_")), key=lambda n: -(agg["fetch"][n][1] + agg["write"][n][1]))
        out = {}
        total = 0.0
        for n in names:
            cf, vf = agg["fetch"][n]
            cw, vw = agg["write"][n]
            fb = cal_f * vf / cf
            wb = cal_w * vw / max(cw, 1)
            out[n] = fb + wb
            f.write(f"\"{n}\",{cf},{fb:.0f},{wb:.0f},{(fb + wb) / 1e6:.1f}\n")
            total += cf * (fb + wb)
    print(f"all dycore kernels, all dispatches of the PMC run: {total / 1e9:.1f} GB")
    json.dump({"fetch_cal": cal_f, "write_cal": cal_w, "traffic_bytes_per_launch": out, "total_bytes_all_dispatches": total}, open(f"{prefix}_hbm_pmc.json", "w"), indent=1)
    print(open(f"{prefix}_hbm_pmc.csv").read()[:3000])


if __name__ == "__main__":
    main()
