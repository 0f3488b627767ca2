#!/usr/bin/env python3
"""Per-kernel averages of the counters in one or more rocprofv3 --pmc result databases (rocpd .db).
usage: pmc_summary.py out.txt a.db [b.db ...]"""
import sqlite3
import sys

out = open(sys.argv[1], "w")
for db in sys.argv[2:]:
    c = sqlite3.connect(db)
    try:
        rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                         "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    except sqlite3.Error as e:
        out.write(f"{db}: {e}\n")
        continue
    cur = None
    for k, n, v, cnt in rows:
        if k != cur:
            out.write(f"\n{k[:150]}  ({cnt} dispatches)\n")
            cur = k
        out.write(f"    {n:32s} {v:18.1f}\n")
out.close()
print(open(sys.argv[1]).read()[:6000])
