#!/bin/bash
# Run ON the GPU box: 8-rank projection with the owned-row backward (default) against the round-2 halo-row backward, three shapes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/owned_ab; mkdir -p $O; cd $R
B="timeout 900 python bench.py --no-cpu-baseline --warmup 3 --steps 5 --emulate-ranks 8"
run() { name=$1; shift; for rep in 1 2; do $B "$@" > $O/${name}_owned_$rep.json 2>> $O/err.txt; $B --halo-row-backward "$@" > $O/${name}_halo_$rep.json 2>> $O/err.txt; done; }
run headline
run config3 --events 20000000 --nodes 1000000 --features 128
run f256 --features 256
for f in $O/*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f'.split('/')[-1], round(d['max_rank_compute_ms'],2), round(d['projected_ms_per_step'],2), round(d['loss'],6))
"; done
