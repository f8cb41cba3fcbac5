"""Per-kernel sums of the SQ counters of one or more rocprofv3 --pmc passes (rocpd SQLite):
python tools/pmc_sq.py <kernel-substring> <db> [<db> ...]"""
import sqlite3
import sys

key = sys.argv[1]
for db in sys.argv[2:]:
    cur = sqlite3.connect(db).cursor()
    q = "select counter_name, count(*), sum(value) from counters_collection where kernel_name like ? group by counter_name"
    for name, n, v in cur.execute(q, (f"%{key}%",)):
        print(f"{name:34s} dispatches {n:5d}  sum {v:18.0f}  per-dispatch {v / n:16.0f}")
