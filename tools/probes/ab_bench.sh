#!/bin/bash
# Run ON the GPU box: same-box A/B of bench.py between the tree under _ab_old/ (git archive of an earlier commit, built here) and this tree.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O
for i in 1 2 3; do
  for side in old new; do
    D=$R; [ $side = old ] && D=$R/_ab_old
    (cd $D && python bench.py --no-cpu-baseline --steps 20 "$@" 2>/dev/null) > $O/${side}_$i.json
    python -c "
import json;d=json.load(open('$O/${side}_$i.json'));print('$side', $i, round(d['ms_per_step'],3), round(d['lift_ms'],3), round(d['dbgnn_step_ms'],3))"
  done
done
