#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timeline; mkdir -p $O
rm -rf /tmp/p_tl; rocprofv3 --kernel-trace -d /tmp/p_tl -o x -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $O/line.json 2> $O/err.txt
python $R/tools/rocprof_timeline.py $(find /tmp/p_tl -name "*.db" | head -1) 2 > $O/timeline.txt
tail -2 $O/timeline.txt
