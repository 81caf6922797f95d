#!/usr/bin/env python3
"""Ordered kernel list of ONE steady-state step of a traced bench.py run (rocpd database of `rocprofv3 --kernel-trace`): start offset,
duration, gap to the previous kernel.  Steps are delimited by the k_tail_keys launches (first kernel of the lift).

    python tools/rocprof_timeline.py x_results.db [step_from_end=2]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "k_db2_keys" in r[0]] or [i for i, r in enumerate(rows) if "k_tail_keys" in r[0]]
# (steps of the fused builder start with k_db2_keys; the generic kernels' k_tail_keys only delimits runs made with --builder generic)
a, b = marks[-back - 1], marks[-back]
seg = rows[a:b]
t0 = seg[0][1]
prev_end = t0
busy = 0
for name, s, e in seg:
    short = name.replace("void ", "").replace("pp::", "pp::")[:90]
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {short}")
    prev_end = max(prev_end, e)
    busy += e - s
print(f"# {len(seg)} kernels, wall {(seg[-1][2] - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms")
