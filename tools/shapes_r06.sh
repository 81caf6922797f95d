#!/bin/bash
# Run ON the GPU box: every file of profiles/r06_shapes/ — the multi-order builds (headline stream K = 2..5, configs[2] generator, contact / ties
# streams, configs[4]'s 10^8-event stream), their kernel table, and the non-headline bench lines.  -> gpurun_out/shapes/  (copy to profiles/r06_shapes/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/shapes; mkdir -p $O
cd $R
timeout 600 python tools/probes/multi_order.py > $O/multi_order.txt 2> $O/multi_order.err
timeout 900 python tools/probes/config2_scale_free.py > $O/config2_scale_free.txt 2> $O/config2.err
timeout 600 python tools/probes/multi_order_shapes.py > $O/multi_order_shapes.txt 2> $O/multi_order_shapes.err
{ timeout 600 python tools/probes/multi_order_k.py 5 3 1e8 5e6 1e8 5e6; timeout 600 python tools/probes/multi_order_k.py 5 2 1e8 5e6 1e8 1e7; } > $O/events_1e8_k5.txt 2> $O/events_1e8_k5.err
bash tools/probes/prof_script.sh mo5 tools/probes/multi_order_k.py 5 3 > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/probes/multi_order_k.py 5 3   (3 builds of MultiOrderModel.from_temporal_graph(max_order=5) on the headline stream + the stream's set-up; MI355X)"; cat $R/gpurun_out/stats_mo5.txt; echo; cat $R/gpurun_out/out_mo5.txt; } > $O/multi_order_kernel_stats.txt
bash tools/probes/prof_script.sh c2k3 tools/probes/config2_k3.py 1500000 3 > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/probes/config2_k3.py 1500000 3   (3 builds at max_order=3 on the configs[2] generator; MI355X)"; cat $R/gpurun_out/stats_c2k3.txt; echo; cat $R/gpurun_out/out_c2k3.txt; } > $O/config2_k3_kernel_stats.txt
B="timeout 600 python bench.py --no-cpu-baseline --warmup 3 --no-multi-order --no-hub-streams"
$B --emulate-ranks 8 --steps 5 > $O/emulate8.json 2> $O/emulate.err
$B --steps 10 --mode streams > $O/streams_mode.json 2> $O/streams.err
$B --steps 10 --events 2000000 --nodes 100000 --span 1000000 --delta 100000 > $O/config1.json 2> $O/config1.err
$B --steps 5 --features 128 > $O/f128.json 2> $O/f128.err
$B --steps 5 --features 256 > $O/f256.json 2> $O/f256.err
$B --steps 5 --events 20000000 --nodes 1000000 --features 128 > $O/config3_per_gpu.json 2> $O/config3.err
timeout 600 python tools/probes/hub_streams.py > $O/hub_streams.txt 2> $O/hub_streams.err
tail -c 200 $O/*.err
for f in $O/*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('ms_per_step','value','api_path_ms_per_step','projected_ms_per_step','max_rank_compute_ms','peak_hbm_gib') if k in d})
"; done
cat $O/multi_order.txt $O/config2_scale_free.txt $O/multi_order_shapes.txt $O/events_1e8_k5.txt
