#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd (SQLite) output.

    python tools/rocpd_pmc.py gpurun_out/pmc_FETCH_SIZE/mb_results.db [name-filter ...]
"""
import re
import sqlite3
import sys


def main(path, filters):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
                       "from counters_collection group by kernel_name, counter_name order by 4 desc").fetchall()
    print("| kernel | counter | dispatches | avg | min | max | avg us |")
    print("|---|---|---:|---:|---:|---:|---:|")
    for name, ctr, n, avg, mn, mx, dur in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        if "at::native" in short:
            continue
        if filters and not any(f in short for f in filters):
            continue
        print(f"| `{short[:90]}` | {ctr} | {n} | {avg:.1f} | {mn:.1f} | {mx:.1f} | {dur / 1e3:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
