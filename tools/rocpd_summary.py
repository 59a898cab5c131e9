#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd/SQLite) kernel trace into a per-kernel stats table (the `--stats` view).

    python tools/rocpd_summary.py gpurun_out/prof_r01/bench_results.db > profiles/r01_bench_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\[clone .kd\]", "", name).strip()
    return name if len(name) <= 120 else name[:117] + "..."


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(
        f"select {name_col}, count(*), sum(end - start), min(end - start), max(end - start) "
        f"from kernels group by {name_col} order by 3 desc"
    ).fetchall()
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary ({path})\n")
    print(f"total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, tot, mn, mx in rows:
        print(f"| `{short(name)}` | {n} | {tot / 1e6:.3f} | {tot / n / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100.0 * tot / total:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
