#!/bin/bash
# Run ON the GPU box (gpurun -- bash tools/collect_profiles.sh [TAG]): rocprofv3 kernel statistics and the two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains) of the default bench command; only text / json summaries go to
# gpurun_out/profiles/ (the rocpd databases stay in /tmp).  Copy the summaries into profiles/ afterwards (named per round).
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
# (--no-multi-order / --no-hub-streams: those sections run other streams through other kernels after the timed region; the per-dispatch means of the
# PMC passes are taken over the headline step's launches)
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-multi-order --no-hub-streams"
rm -rf /tmp/p_stats; rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o x -- $CMD > /tmp/log_stats.txt 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/p_stats -name "*.db" | head -1) --top 70 > $OUT/${TAG}_bench_kernel_stats.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C; rocprofv3 --pmc $C --kernel-trace -d /tmp/p_$C -o x -- $CMD > /tmp/log_$C.txt 2>&1
  python $R/tools/rocprof_pmc.py $(find /tmp/p_$C -name "*.db" | head -1) --top 40 --json /tmp/pmc_$C.json > $OUT/${TAG}_pmc_$C.txt 2>&1
done
python - <<PY
import json
f, w = json.load(open("/tmp/pmc_FETCH_SIZE.json")), json.load(open("/tmp/pmc_WRITE_SIZE.json"))
out = {}
def pick(table, pat, counter):
    for k, v in table.items():
        if pat in k and k.endswith("|" + counter):
            return v
    return None
# (the backward kernel runs as two instantiations: the register-capped one on the higher-order graph, the plain one on the first-order graph)
for key, pat in (("k_gcn_forward", "k_gcn_forward<64, 64"), ("k_gcn_backward@ho", "k_gcn_backward<64, 64, false, false, false, true>"),
                 ("k_gcn_backward@fo", "k_gcn_backward<64, 64, false, false, false, false>"), ("k_spmm_v4", "k_spmm_v4<16, 2"),
                 ("k_expand", "k_expand<true>"), ("k_temporal_count", "k_temporal_count"), ("k_spmm_act_backward", "k_spmm_act_backward<16>"),
                 ("k_weight_grad64", "k_weight_grad64"), ("k_db2_mid@count", "k_db2_mid<long, 0, false"), ("k_db2_mid@fill", "k_db2_mid<long, 0, true"),
                 ("k_db2_out", "k_db2_out<false>"), ("k_db2_gather_out", "k_db2_gather_out"), ("k_db2_out_ids", "k_db2_out_ids"),
                 ("k_db2_keys", "k_db2_keys"), ("k_db2_unzip", "k_db2_unzip")):
    fe, wr = pick(f, pat, "FETCH_SIZE"), pick(w, pat, "WRITE_SIZE")
    if not fe or not wr or fe["dispatches"] != wr["dispatches"]:
        continue
    # the two passes run the same program: dispatch i of one pass is dispatch i of the other.  HBM bytes = 2 * FETCH + WRITE (KiB -> bytes)
    fv, wv = fe["values_in_dispatch_order"], wr["values_in_dispatch_order"]
    tot = [(2.0 * a + b) * 1024 for a, b in zip(fv, wv)]
    tot = [t_ for t_ in tot]
    if key in ("k_expand", "k_temporal_count") and len(tot) > 3:               # (generic lift kernels: only the 3 untimed passes of the bench run them)
        pass
    big = [i for i, t_ in enumerate(tot) if t_ >= 0.4 * max(tot)]            # launches on the 10^7-row higher-order graph
    small = [i for i, t_ in enumerate(tot) if t_ < 0.4 * max(tot)]           # launches on the 5*10^5-row first-order graph
    for suffix, sel in ((("@ho", big), ("@fo", small)) if (small and "@" not in key) else (("", big + small),)):
        out[key + suffix] = {"fetch_kib_per_dispatch": sum(fv[i] for i in sel) / len(sel), "write_kib_per_dispatch": sum(wv[i] for i in sel) / len(sel),
                             "fetch_correction": 2.0, "hbm_bytes_per_dispatch": sum(tot[i] for i in sel) / len(sel), "dispatches_in_profile": len(sel)}
json.dump({"workload": "bench.py defaults (m=10^7, N=5*10^5, delta=10^6, F=64), partition mode at 1 GPU",
           "source": "${TAG}_pmc_FETCH_SIZE.txt + ${TAG}_pmc_WRITE_SIZE.txt (separate rocprofv3 --pmc passes); FETCH_SIZE doubled (gfx950); "
                     "@ho / @fo = the dispatch-size clusters of the 10^7-row higher-order and the 5*10^5-row first-order graph", **out},
          open("$OUT/${TAG}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
# the bench line LAST, with the table of THIS profile in place (bench.py reads roofline.traffic from profiles/<TAG>_pmc_traffic.json: the kept line and
# the table it cites then describe the same tree — VERDICT r5 #10)
cp $OUT/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
timeout 900 python $R/bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_stderr.txt
timeout 300 python $R/tools/bench_kernels.py --ops lift,agg,plan,spmm,dense,gcn > $OUT/${TAG}_per_op_timings.txt 2>&1
tail -c 600 $OUT/${TAG}_bench_line.json
