#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel statistics of bench.py with the given arguments -> gpurun_out/stats_<tag>.txt.  usage: stats_any.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift
mkdir -p $R/gpurun_out
rm -rf /tmp/p_any; rocprofv3 --kernel-trace --stats -d /tmp/p_any -o x -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > /tmp/any_line.json 2> /tmp/any_err.txt
python $R/tools/rocprof_summary.py $(find /tmp/p_any -name "*.db" | head -1) --top 40 > $R/gpurun_out/stats_$TAG.txt
head -16 $R/gpurun_out/stats_$TAG.txt | cut -c1-150
