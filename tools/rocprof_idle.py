#!/usr/bin/env python3
"""GPU idle time inside the steady-state steps of a profiled bench.py run (rocpd database of `rocprofv3 --kernel-trace`).
Steps are delimited by the k_temporal_count launches (one per step); prints busy / idle time per step and the longest gaps."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "k_temporal_count" in r[0]]
if len(marks) < 6:
    sys.exit("need at least 6 steps in the trace")
for a, b in zip(marks[-5:-2], marks[-4:-1]):
    seg = rows[a:b]
    wall = seg[-1][2] - seg[0][1]
    busy = sum(e - s for _, s, e in seg)
    gaps = sorted(((seg[i + 1][1] - seg[i][2]) / 1e3, seg[i][0][:50], seg[i + 1][0][:50]) for i in range(len(seg) - 1))
    print(f"step: {len(seg)} kernels, wall {wall / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(wall - busy) / 1e6:.3f} ms")
    for g, x, y in gaps[-6:][::-1]:
        print(f"    gap {g:8.1f} us  after {x}  before {y}")
