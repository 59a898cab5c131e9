#!/usr/bin/env python3
"""Timeline of the LAST graph replay in a rocprofv3 kernel trace of `bench.py --lanes 1 --no-breakdown ...` (rocpd SQLite): every dispatch
from the last style_affine_kernel on, in start order, with its start offset, duration and the gap to the previous dispatch's end.

    python tools/rocpd_timeline.py trace/bench_results.db [n_replays_back]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\[clone .kd\]", "", name).strip()
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:70]


def main(path, back=1):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    starts = [i for i, r in enumerate(rows) if "style_affine_kernel" in r[0]]
    i0 = starts[-back]
    i1 = starts[-back + 1] if back > 1 else len(rows)
    t0 = rows[i0][1]
    prev_end = t0
    print(f"# dispatches {i0}..{i1} of {len(rows)} ({path})\n")
    print("| # | kernel | start us | dur us | gap us |")
    print("|---:|---|---:|---:|---:|")
    for k, (name, s, e) in enumerate(rows[i0:i1]):
        print(f"| {k} | `{short(name)}` | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {(s - prev_end) / 1e3:.1f} |")
        prev_end = max(prev_end, e)
    print(f"\nspan {(prev_end - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
