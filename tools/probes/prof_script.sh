#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel statistics of any python script -> gpurun_out/stats_<tag>.txt.  usage: prof_script.sh <tag> <script> [args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; SCRIPT=$R/$2; shift; shift
mkdir -p $R/gpurun_out
rm -rf /tmp/p_s; rocprofv3 --kernel-trace --stats -d /tmp/p_s -o x -- python $SCRIPT "$@" > $R/gpurun_out/out_$TAG.txt 2> $R/gpurun_out/err_$TAG.txt
python $R/tools/rocprof_summary.py $(find /tmp/p_s -name "*.db" | head -1) --top 45 > $R/gpurun_out/stats_$TAG.txt
head -40 $R/gpurun_out/stats_$TAG.txt | cut -c1-160
