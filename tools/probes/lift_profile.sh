#!/bin/bash
# Run ON the GPU box: per-kernel rocprofv3 statistics of the lift micro-benchmark (tools/bench_kernels.py --ops lift).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OPS=${1:-lift}
rm -rf /tmp/prof_lift
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_lift -o lift -- python $R/tools/bench_kernels.py --ops $OPS --iters 5 > /tmp/log_lift.txt 2>&1
grep -v amdgpu.ids /tmp/log_lift.txt | tail -${2:-12}
python $R/tools/rocprof_summary.py $(find /tmp/prof_lift -name "*.db" | head -1) --top ${3:-24} 2>&1 | cut -c1-150
