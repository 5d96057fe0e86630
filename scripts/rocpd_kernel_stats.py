"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace) into a per-kernel stats CSV
(name, calls, total_ns, avg_ns, min_ns, max_ns, pct) -- the same table `rocprofv3 --stats` prints."""
import csv
import sqlite3
import sys


def main(db_path, out_csv):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in kcols else ("kernel_name" if "kernel_name" in kcols else kcols[-1])
    q = (f"select s.{name_col}, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by 1 order by 3 desc")
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(100.0 * r[2] / tot, 3)])
    print(f"{len(rows)} kernels, total {tot/1e6:.2f} ms -> {out_csv}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
