#!/bin/bash
# Run ON the GPU box (gpurun -- bash tools/collect_profiles.sh): rocprofv3 kernel statistics and the two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE) of the default bench command; only text summaries go to gpurun_out/profiles/ (the rocpd databases
# stay in /tmp).  Copy the summaries into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rm -rf /tmp/p_stats; rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o x -- $CMD > /tmp/log_stats.txt 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/p_stats -name "*.db" | head -1) --top 60 > $OUT/bench_kernel_stats.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C; rocprofv3 --pmc $C --kernel-trace -d /tmp/p_$C -o x -- $CMD > /tmp/log_$C.txt 2>&1
  python $R/tools/rocprof_pmc.py $(find /tmp/p_$C -name "*.db" | head -1) --top 40 > $OUT/pmc_$C.txt 2>&1
done
python $R/bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.txt
python - <<PY
import json, re
out = {}
def grab(path):
    rows = {}
    for line in open(path):
        m = re.match(r"(.{68,70}?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+(\d+)\s+([\d.]+)\s*$", line)
        if m:
            rows[m.group(1).strip()] = (int(m.group(3)), float(m.group(5)))
    return rows
f, w = grab("$OUT/pmc_FETCH_SIZE.txt"), grab("$OUT/pmc_WRITE_SIZE.txt")
for key, pat in (("k_gcn_forward", "k_gcn_forward<64, 64"), ("k_gcn_backward", "k_gcn_backward<64, 64"), ("k_spmm_v4", "k_spmm_v4<16, 2"),
                 ("k_expand", "k_expand<true>")):
    fk = [k for k in f if pat in k]
    wk = [k for k in w if pat in k]
    if fk and wk:
        fe, wr = f[fk[0]], w[wk[0]]
        out[key] = {"fetch_kib_per_dispatch": fe[1], "write_kib_per_dispatch": wr[1], "fetch_correction": 2.0,
                    "hbm_bytes_per_dispatch": (2.0 * fe[1] + wr[1]) * 1024, "dispatches_in_profile": fe[0]}
json.dump({"workload": "bench.py defaults (m=10^7, N=5*10^5, delta=10^6, F=64)",
           "source": "pmc_FETCH_SIZE.txt + pmc_WRITE_SIZE.txt (separate rocprofv3 --pmc passes); FETCH_SIZE doubled (gfx950)", **out},
          open("$OUT/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -c 1500 $OUT/bench_line.json
