#!/bin/bash
# Run ON the GPU box: same-box A/B of the 8-rank projection between _ab_old/ and this tree.
R=$GRAFT_REPO_ROOT
for i in 1 2; do for side in old new; do
  D=$R; [ $side = old ] && D=$R/_ab_old
  (cd $D && python bench.py --no-cpu-baseline --emulate-ranks ${1:-8} --steps 5 --warmup 3 2>/dev/null) | python -c "
import json,sys;d=json.loads(sys.stdin.readlines()[-1]);print('$side', round(d['max_rank_compute_ms'],2), round(d['projected_ms_per_step'],2), round(d['amdahl_terms_ms']['of_which_graph_build'],2))"
done; done
