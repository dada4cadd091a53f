#!/usr/bin/env python3
"""Dump per-kernel averages of every PMC counter found in the rocprofv3 databases of a directory (matinv kernels only).
usage: pmc_dump.py <dir with *_results.db> <out.txt> <tag>"""
import glob
import os
import sqlite3
import sys

d, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
lines = [f"# {tag}: rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --workload W --steps 3 --warmup 1 (separate passes)",
         "# per-dispatch averages over the dispatches of each matinv kernel; FETCH_SIZE / WRITE_SIZE are KiB as reported",
         f"{'pass':<18} {'counter':<26} {'avg':>18} {'n':>4}  kernel"]
for f in sorted(glob.glob(os.path.join(d, "*_results.db"))):
    cur = sqlite3.connect(f).cursor()
    try:
        rows = list(cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                "where kernel_name like '%matinv%' group by kernel_name, counter_name"))
    except sqlite3.Error:
        continue
    for k, c, v, n in rows:
        short = k.split("(")[0].replace("void matinv::", "")
        lines.append(f"{os.path.basename(f).replace('_results.db',''):<18} {c:<26} {v:>18.1f} {n:>4}  {short}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:3]), f"\n... {len(lines)-3} rows")
