#!/bin/bash
# Run HERE after `gpurun -- 'bash tools/collect_profiles.sh r06; bash tools/shapes_r06.sh; bash tools/probes/multi_order_pmc.sh'`: copies the summaries
# from gpurun_out/ into profiles/r06_* (+ profiles/r06_shapes/) with the command and the tree they come from in their first line.
C=$(git rev-parse --short HEAD); O=gpurun_out/profiles; CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-multi-order --no-hub-streams"
{ echo "# round 6, tools/collect_profiles.sh r06 (rocprofv3 --kernel-trace --stats -- $CMD; MI355X, tree $C)"; cat $O/r06_bench_kernel_stats.txt; } > profiles/r06_bench_kernel_stats.txt
{ echo "# round 6: rocprofv3 --pmc FETCH_SIZE --kernel-trace -- $CMD   (separate pass; KiB; gfx950: FETCH_SIZE tallies 64 B per 128-B request -> doubled in r06_pmc_traffic.json; tree $C)"; cat $O/r06_pmc_FETCH_SIZE.txt; } > profiles/r06_pmc_FETCH_SIZE.txt
{ echo "# round 6: rocprofv3 --pmc WRITE_SIZE --kernel-trace -- $CMD   (separate pass; KiB; tree $C)"; cat $O/r06_pmc_WRITE_SIZE.txt; } > profiles/r06_pmc_WRITE_SIZE.txt
cp $O/r06_pmc_traffic.json profiles/r06_pmc_traffic.json; cp $O/r06_bench_line.json profiles/r06_bench_line.json; cp $O/r06_bench_stderr.txt profiles/r06_bench_stderr.txt
{ echo "# python tools/bench_kernels.py --ops lift,agg,plan,spmm,dense,gcn  (per-op medians, HIP events; headline workload m=10^7, N=5*10^5, F=64; MI355X, tree $C)"; cat $O/r06_per_op_timings.txt; } > profiles/r06_per_op_timings.txt
mkdir -p profiles/r06_shapes; S=gpurun_out/shapes
for f in multi_order.txt config2_scale_free.txt multi_order_shapes.txt events_1e8_k5.txt multi_order_kernel_stats.txt config2_k3_kernel_stats.txt emulate8.json streams_mode.json config1.json f128.json f256.json config3_per_gpu.json hub_streams.txt; do cp $S/$f profiles/r06_shapes/$f; done
{ echo "# round 6: HBM traffic of the level-by-level multi-order builder (tools/probes/multi_order_pmc.sh; MI355X, tree $C)"; cat gpurun_out/multi_order_pmc.txt; } > profiles/r06_multi_order_pmc.txt
python - <<'PY'
import json
l = json.loads(open('profiles/r06_bench_line.json').read().strip().splitlines()[-1])
print("step", round(l['ms_per_step'], 3), "ms, value", l['value'], "roofline.frac", round(l['roofline']['frac'], 3), "traffic from", l['roofline'].get('traffic_source'))
mo = l['multi_order']
for name in list(mo)[2:]:
    print(name, {k: (round(v['ms'], 2), v['level_by_level']) for k, v in mo[name].items() if k.startswith('K=')}, [round(r['ms'], 2) for r in mo[name]['layers']])
PY
