#!/bin/bash
# Run ON the GPU box: HBM traffic of the level-by-level multi-order builder (separate FETCH_SIZE / WRITE_SIZE passes, never combined with other trace
# domains) on the headline stream, K = 5, 3 builds -> gpurun_out/multi_order_pmc.txt.  gfx950: FETCH_SIZE tallies 64 B per 128-B request — doubled below.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/multi_order_pmc.txt
{
for C in FETCH_SIZE WRITE_SIZE; do
  echo "######## rocprofv3 --pmc $C --kernel-trace -- python tools/probes/multi_order_k.py 5 3   (KiB per dispatch-size cluster; FETCH_SIZE: double it for bytes)"
  rm -rf /tmp/p_m; rocprofv3 --pmc $C --kernel-trace -d /tmp/p_m -o x -- python $R/tools/probes/multi_order_k.py 5 3 > /dev/null 2>&1
  python $R/tools/rocprof_pmc.py $(find /tmp/p_m -name "*.db" | head -1) --top 14 --json /tmp/pmc_m_$C.json | cut -c1-230
done
python - <<PY
import json
f, w = json.load(open("/tmp/pmc_m_FETCH_SIZE.json")), json.load(open("/tmp/pmc_m_WRITE_SIZE.json"))
print("######## HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB, dispatch by dispatch (same program, same order), last build (levels 1..4 -> layers 2..5)")
for key in f:
    name = key.split("|")[0]
    if not name.startswith(("pp::k_mo_", "void pp::k_mo_")):
        continue
    wk = name + "|WRITE_SIZE"
    if wk not in w or w[wk]["dispatches"] != f[key]["dispatches"]:
        continue
    fv, wv = f[key]["values_in_dispatch_order"], w[wk]["values_in_dispatch_order"]
    per = [(2.0 * a + b) * 1024 / 1e9 for a, b in zip(fv, wv)]
    n = len(per) // 3
    print(f"{name[:90]:90s} GB per launch, last build: " + " ".join(f"{x:7.3f}" for x in per[-n:]))
PY
} > $O 2>&1
cat $O | cut -c1-200
