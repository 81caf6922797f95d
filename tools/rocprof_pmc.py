#!/usr/bin/env python3
"""Per-kernel PMC totals from a rocprofv3 rocpd database collected with ``--pmc <counter> --kernel-trace``.

    python tools/rocprof_pmc.py gpurun_out/pmc_fetch/f_results.db [--top 25]
Values are summed over all dispatches of a kernel and also given per dispatch.  FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads (MI355X_MICROARCH.md §HBM): double it
before comparing with byte counts of 16 B/lane streams.
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" in tables:
        cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        name = "kernel_name" if "kernel_name" in cols else "name"
        cname = "counter_name" if "counter_name" in cols else "pmc_name"
        val = "value" if "value" in cols else "counter_value"
        rows = db.execute(f"select {name}, {cname}, count(distinct dispatch_id), sum({val}) from counters_collection "
                          f"group by {name}, {cname} order by 4 desc").fetchall()
    else:
        raise SystemExit(f"no counters_collection view; tables: {tables}")
    print(f"{'kernel':<70} {'counter':<12} {'disp':>5} {'sum':>16} {'per_dispatch':>16}")
    for k, c, n, s in rows[: a.top]:
        print(f"{k[:68]:<70} {c:<12} {n:>5} {s:>16.0f} {s / max(n, 1):>16.1f}")


if __name__ == "__main__":
    main()
