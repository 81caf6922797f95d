#!/usr/bin/env python3
"""Per-kernel PMC totals from a rocprofv3 rocpd database collected with ``--pmc <counter> --kernel-trace``.

    python tools/rocprof_pmc.py gpurun_out/pmc_fetch/f_results.db [--top 25] [--json out.json]
Values are summed over all dispatches of a kernel and also given per dispatch.  A kernel launched on graphs of very different size
(the 10^7-row higher-order graph and the 5*10^5-row first-order graph) is split into size CLUSTERS: dispatches are sorted by value and cut
where the value jumps by more than 1.6x, so every (kernel, graph) pair gets its own per-dispatch figure.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads (MI355X_MICROARCH.md §HBM):
double it before comparing with byte counts of 16 B/lane streams.
"""
import argparse
import json
import sqlite3


def clusters(values):
    vals = sorted(values)
    out, cur = [], [vals[0]]
    for v in vals[1:]:
        if cur[-1] > 0 and v > 1.6 * cur[-1]:
            out.append(cur)
            cur = []
        cur.append(v)
    out.append(cur)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" not in tables:
        raise SystemExit(f"no counters_collection view; tables: {tables}")
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    name = "kernel_name" if "kernel_name" in cols else "name"
    cname = "counter_name" if "counter_name" in cols else "pmc_name"
    val = "value" if "value" in cols else "counter_value"
    rows = db.execute(f"select {name}, {cname}, dispatch_id, sum({val}) from counters_collection group by {name}, {cname}, dispatch_id "
                      f"order by dispatch_id").fetchall()
    per = {}
    for k, c, _, v in rows:
        per.setdefault((k, c), []).append(float(v))
    table = sorted(per.items(), key=lambda kv: -sum(kv[1]))
    print(f"{'kernel':<70} {'counter':<12} {'disp':>5} {'sum':>16} {'per_dispatch':>16}  clusters (count x mean)")
    dump = {}
    for (k, c), vals in table[: a.top]:
        cl = clusters(vals)
        text = ", ".join(f"{len(g)} x {sum(g) / len(g):.1f}" for g in cl)
        print(f"{k[:68]:<70} {c:<12} {len(vals):>5} {sum(vals):>16.0f} {sum(vals) / len(vals):>16.1f}  {text}")
        dump[f"{k}|{c}"] = {"dispatches": len(vals), "per_dispatch_mean": sum(vals) / len(vals), "values_in_dispatch_order": vals,
                            "clusters": [{"count": len(g), "mean": sum(g) / len(g)} for g in cl]}
    if a.json:
        with open(a.json, "w") as fh:
            json.dump(dump, fh, indent=1)


if __name__ == "__main__":
    main()
