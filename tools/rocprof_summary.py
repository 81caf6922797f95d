#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (``--kernel-trace --stats``) as a per-kernel table.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db [--top 40] > profiles/rNN_xxx.txt
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(
        f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
        f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':<78} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'%':>6}")
    for name, calls, tot, avg, mn, mx in rows[: a.top]:
        short = name if len(name) <= 76 else name[:73] + "..."
        print(f"{short:<78} {calls:>6} {tot / 1e6:>10.3f} {avg / 1e3:>10.1f} {mn / 1e3:>10.1f} {mx / 1e3:>10.1f} {100 * tot / total:>6.2f}")
    print(f"{'TOTAL GPU kernel time':<78} {sum(r[1] for r in rows):>6} {total / 1e6:>10.3f}")


if __name__ == "__main__":
    main()
