#!/usr/bin/env python
"""Condense the rocprofv3 outputs of scripts/profile_bench.sh into the small CSVs kept under profiles/.

  python scripts/rocprof_summary.py gpurun_out profiles/r02_v1

writes <prefix>_kernel_stats.csv  (rocprofv3 --kernel-trace --stats: calls, total, average per kernel) and
       <prefix>_hbm_pmc.csv       (FETCH_SIZE / WRITE_SIZE per dispatch, separate --pmc passes, in bytes after the
                                   unit (KB) and gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md,
                                   calibrated on k_uhtr whose traffic is known exactly).
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
    # calibration on k_uhtr (uhtr = uhtr + dt*uh, vhtr = vhtr + dt*vh: one launch per step that reads four 3-D arrays and writes
    # two, nothing else): the guide's "calibrate on a known byte count in your own access pattern" (FETCH_SIZE counts 64 B
    # per 128-B request on gfx950: a factor close to 2 on streaming reads; WRITE_SIZE close to 1)
    cells = (ni + 1) * nj * nk * 8.0
    cal_k = "k_uhtr"
    n_cal = agg["fetch"][cal_k][0]
    exp_fetch = cells * 4.0
    exp_write = cells * 2.0
    cal_f = exp_fetch / (agg["fetch"][cal_k][1] / n_cal)
    cal_w = exp_write / (agg["write"][cal_k][1] / agg["write"][cal_k][0])
    nsteps = n_cal     # k_uhtr runs once per step
    with open(f"{prefix}_hbm_pmc.csv", "w") as f:
        f.write(f"# FETCH_SIZE x{cal_f:.3f}, WRITE_SIZE x{cal_w:.3f} (calibration on k_uhtr: known {exp_fetch / 1e6:.1f} MB read, "
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
    per_step = {n: agg["fetch"][n][0] / nsteps for n in out}
    json.dump({"fetch_cal": cal_f, "write_cal": cal_w, "steps_in_run": nsteps, "traffic_bytes_per_launch": out,
               "launches_per_step": per_step, "bytes_per_step": total / nsteps, "total_bytes_all_dispatches": total,
               "note": "dynamics only (bench.py --tracers -1): every k_* dispatch of the run / the number of steps in it"},
              open(f"{prefix}_hbm_pmc.json", "w"), indent=1)
    print(open(f"{prefix}_hbm_pmc.csv").read()[:3000])


if __name__ == "__main__":
    main()
