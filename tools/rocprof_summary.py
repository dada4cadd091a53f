#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (…_results.db) into the compact per-kernel summary committed under profiles/.
usage: rocprof_summary.py results.db [title] > profiles/<name>.txt"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    print(f"# {title}")
    print("# source: rocprofv3 --kernel-trace --stats (rocpd sqlite, view top_kernels); durations in microseconds")
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>12} {'pct':>7}  kernel")
    for name, calls, total, avg, pct in db.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels"):
        short = name if len(name) < 150 else name[:147] + "..."
        print(f"{calls:>6} {total:>12.3f} {avg:>12.3f} {pct:>7.2f}  {short}")
    try:
        rows = list(db.execute("select name, count(*), avg(value), sum(value) from counters_collection group by name"))
        if rows:
            print("\n# PMC counters (per dispatch average, sum)")
            for r in rows:
                print(r)
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main()
