#!/bin/bash
# Run ON the GPU box: 8-rank projection, first-order shard with the dense halo against the discovered halo (two alternations, --trace).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fo_halo_ab; mkdir -p $O; cd $R
B="timeout 900 python bench.py --no-cpu-baseline --warmup 3 --steps 5 --emulate-ranks 8 --trace"
for rep in 1 2; do for mode in dense discovered; do $B --fo-halo $mode "$@" > $O/${mode}_$rep.json 2>> $O/err.txt; done; done
for f in $O/*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); p=d['phase_ms_rank1']
print('$f'.split('/')[-1], round(d['max_rank_compute_ms'],2), round(d['projected_ms_per_step'],2), {k.split(':')[1].strip()[:28]: round(v,2) for k,v in p.items() if 'first-order' in k})
"; done
