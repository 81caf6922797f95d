#!/bin/bash
# Run ON the GPU box: the 8-rank projection (bench.py --emulate-ranks 8) at the sizes the north star's 8-GPU configs name, each next to the
# single-GPU step of the same stream measured on the same box.  -> gpurun_out/proj/*.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/proj; mkdir -p $O
cd $R
B="timeout 900 python bench.py --no-cpu-baseline --warmup 8 --steps ${STEPS:-10}"
# (8 warm-up steps, for both runs so that their losses compare: 8 ranks share ONE caching allocator, which needs more than 3 steps to stop calling hipMalloc at the larger sizes —
#  the report counts the device allocations inside the timed steps: emulation_allocator.device_mallocs_in_timed_steps must be 0)
run() { name=$1; shift; $B "$@" > $O/${name}_1gpu.json 2> $O/${name}.err; $B --emulate-ranks 8 "$@" > $O/${name}_emulate8.json 2>> $O/${name}.err; }
run headline
$B --emulate-ranks 8 --emulate-clock drain > $O/headline_emulate8_drain_clock.json 2>> $O/headline.err
run config3 --events 20000000 --nodes 1000000 --features 128
run f256 --features 256
run events5e7 --events 50000000 --nodes 2500000 --span 50000000 --delta 5000000
tail -c 300 $O/*.err
for f in $O/*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('ms_per_step','projected_ms_per_step','projected_ms_per_step_aggregate_model','projected_ms_per_step_no_overlap','max_rank_compute_ms','per_rank_host_ms','amdahl_terms_ms','peak_hbm_gib') if k in d})
"; done
