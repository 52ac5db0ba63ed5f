#!/usr/bin/env python3
"""Per-kernel sums of every counter in rocprofv3 --pmc result databases (rocpd sqlite).

  python tools/pmc_dump.py gpurun_out/prof/sq/*_results.db [...] > profiles/rNN_pmc.txt"""
import sqlite3
import sys


def short(name):
    return name.replace("gi::", "").split("(")[0].replace("void ", "")


for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
    # a counter is reported once per dimension instance (XCC, SE, ...): dispatches = rows / instances is not needed for sums
    print(f"# {path}")
    for k, c, n, v in sorted(rows, key=lambda r: (short(r[0]), r[1])):
        print(f"{short(k):44s} {c:32s} rows {n:8d} sum {v:.6g}")
