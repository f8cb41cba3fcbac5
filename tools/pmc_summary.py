"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (rocpd SQLite) per kernel.
usage: python tools/pmc_summary.py <fetch.db> <write.db> <out.csv> <out.json> <steps_in_run>
HBM traffic per the MI355X guide: FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) coalesced reads on
gfx950, so reads = 2 x FETCH_SIZE; WRITE_SIZE is taken as is (uncalibrated).  Values are KB in the DB."""
import csv
import json
import sqlite3
import sys

fdb, wdb, out_csv, out_json, steps = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], float(sys.argv[5])


def per_kernel(db):
    cur = sqlite3.connect(db).cursor()
    return {k: (n, s) for k, n, s in cur.execute("select kernel_name, count(*), sum(value) from counters_collection group by kernel_name")}


def short(k: str):
    """rocprofv3 kernel name -> the name me_last_kernel() / bench.py use (first template arguments only)."""
    import re
    m = re.search(r"(attn2_kernel)<(\d+), (\d+), (\d+), \d+, \d+, \d+, (true|false)", k)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)},{m.group(4)},{'fold' if m.group(5) == 'true' else 'classic'}>"
    m = re.search(r"(gemm8p_kernel)<(\d+), (\d+), (true|false)", k)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)},{m.group(4)}>"
    m = re.search(r"(gemm_kernel)<(\d+), (\d+)", k)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)}>"
    if "conv3_halo_kernel" in k:
        return "conv3_halo_kernel"
    m = re.search(r"(attn_kernel)<(\d+), (\d+)", k)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)},general-dual>"
    return None


F, W = per_kernel(fdb), per_kernel(wdb)
rows = []
for k in sorted(set(F) | set(W), key=lambda k: -(F.get(k, (0, 0))[1] + W.get(k, (0, 0))[1])):
    n = F.get(k, W.get(k))[0]
    f_kb, w_kb = F.get(k, (0, 0))[1], W.get(k, (0, 0))[1]
    rows.append((k, n, f_kb, w_kb))
fam = {}
with open(out_csv, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "dispatches", "FETCH_SIZE_KB_sum", "WRITE_SIZE_KB_sum", "hbm_bytes_per_launch_corrected(2*fetch+write)", "GB_per_step_corrected"])
    for k, n, f_kb, w_kb in rows[:40]:
        per = (2 * f_kb + w_kb) * 1024 / n
        w.writerow([k, n, f"{f_kb:.0f}", f"{w_kb:.0f}", f"{per:.0f}", f"{(2 * f_kb + w_kb) * 1024 / steps / 1e9:.2f}"])
        for name in (short(k), "gemm" if ("gemm_kernel" in k or "gemm8p_kernel" in k or "conv3_halo_kernel" in k) else None):
            if name:
                d = fam.setdefault(name, [0, 0.0])
                d[0] += n
                d[1] += (2 * f_kb + w_kb) * 1024
json.dump({k: {"launches": v[0], "hbm_bytes_per_launch": v[1] / v[0], "gb_per_step": v[1] / steps / 1e9} for k, v in fam.items()}, open(out_json, "w"), indent=1)
print(open(out_json).read())
