#!/usr/bin/env python3
"""Registers / spills / occupancy / LDS per kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
usage: kernel_resources.py pathpyg_amd/csrc/pp_gcn_wide.hip [name filter] [-DX=..]"""
import re, subprocess, sys
src = sys.argv[1]
filt = [a for a in sys.argv[2:] if not a.startswith("-")]
defs = [a for a in sys.argv[2:] if a.startswith("-")]
out = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage", *defs,
                      "-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    k, _, v = m.group(1).partition(":")
    k, v = k.strip(), v.strip()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    if filt and not any(f in name for f in filt):
        continue
    print(f"{name[:70]:70s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>4} spill {r.get('VGPRs Spill','?'):>4} "
          f"occ {r.get('Occupancy [waves/SIMD]','?'):>2} lds {r.get('LDS Size [bytes/block]','?'):>7}")
