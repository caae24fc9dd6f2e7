"""Prints per-kernel sums of the counters in rocprofv3 result databases (rocpd sqlite):  python tools/pmc_summary.py DB [DB ...]"""
import sqlite3, sys
for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    for name, ctr, val, n in rows:
        short = name.split("(")[0].replace("rnnt::(anonymous namespace)::", "").replace("void ", "")
        print(f"{path.split('/')[-2]:14s} {short[:70]:70s} {ctr:34s} {val / n:16.1f} per launch ({n})")
