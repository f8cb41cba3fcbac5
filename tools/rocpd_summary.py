"""Export the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd SQLite output of ROCm 7.2) as CSV.
usage: python tools/rocpd_summary.py <results.db> <out.csv> [steps_in_run]"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent", "ms_per_step"])
    for name, calls, tot, avg, pct in rows:
        if pct < 0.01:
            continue
        w.writerow([name, calls, f"{tot:.1f}", f"{avg:.2f}", f"{pct:.2f}", f"{tot / 1000.0 / steps:.2f}"])
print(f"{len(rows)} kernels -> {out}")
