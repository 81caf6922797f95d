#!/bin/bash
# Run ON the GPU box: every file of profiles/r05_shapes/ — bench lines of the non-headline BASELINE shapes (each with >= 3 warm-up steps), the
# emulation at 2 / 4 / 8 ranks, the API flow and the hub streams.  -> gpurun_out/shapes/   (copy to profiles/r05_shapes/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/shapes; mkdir -p $O
cd $R
B="timeout 600 python bench.py --no-cpu-baseline --warmup 3"
$B --emulate-ranks 8 --steps 5 > $O/emulate8.json 2> $O/emulate.err
$B --emulate-ranks 4 --steps 5 > $O/emulate4.json 2>> $O/emulate.err
$B --emulate-ranks 2 --steps 5 > $O/emulate2.json 2>> $O/emulate.err
$B --steps 10 --mode streams > $O/streams_mode.json 2> $O/streams.err
$B --steps 10 --events 2000000 --nodes 100000 --span 1000000 --delta 100000 > $O/config1.json 2> $O/config1.err
$B --steps 5 --features 128 > $O/f128.json 2> $O/f128.err
$B --steps 5 --features 256 > $O/f256.json 2> $O/f256.err
$B --steps 5 --events 20000000 --nodes 1000000 --features 128 > $O/config3_per_gpu.json 2> $O/config3.err
# configs[4]: one GPU's eighth of the 10^8-event stream at its 256-dim width (the whole stream's DBGNN does not fit one GPU)
$B --steps 5 --events 12500000 --nodes 625000 --span 12500000 --delta 1250000 --features 256 > $O/config4_per_gpu_share.json 2> $O/config4.err
# ten times the headline stream on ONE GPU
timeout 900 python bench.py --no-cpu-baseline --warmup 2 --steps 3 --events 100000000 --nodes 5000000 --span 100000000 --delta 10000000 > $O/events_1e8.json 2> $O/events_1e8.err
timeout 600 python tools/probes/api_flow.py > $O/api_flow.txt 2> $O/api_flow.err
timeout 600 python tools/probes/hub_streams.py > $O/hub_streams.txt 2> $O/hub_streams.err
timeout 300 python tools/probes/long_run.py > $O/long_run.txt 2> $O/long_run.err
timeout 300 python tools/probes/tile_reuse.py > $O/tile_reuse.txt 2> $O/tile_reuse.err
timeout 300 python tools/probes/fwd_stage.py > $O/fwd_stage.txt 2> $O/fwd_stage.err
timeout 900 python tools/probes/config2_scale_free.py > $O/config2_scale_free.txt 2> $O/config2.err
timeout 900 python tools/probes/multi_order.py > $O/multi_order.txt 2> $O/multi_order.err
tail -c 200 $O/*.err
for f in $O/*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('ms_per_step','value','api_path_ms_per_step','projected_ms_per_step','max_rank_compute_ms','peak_hbm_gib') if k in d})
"; done
cat $O/api_flow.txt $O/hub_streams.txt
