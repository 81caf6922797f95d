#!/bin/bash
# Run ON the GPU box: kernel table + per-phase trace of the 8-rank emulation on a 5e7-event stream (per-rank share kernel-bound, not launch-bound).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/emu_big; mkdir -p $O
ARGS="--events 50000000 --nodes 2500000 --span 50000000 --delta 5000000 --no-cpu-baseline"
python $R/bench.py $ARGS --emulate-ranks 8 --steps 4 --warmup 2 --trace > $O/trace.json 2> $O/err.txt
rm -rf /tmp/p_emu; rocprofv3 --kernel-trace -d /tmp/p_emu -o x -- python $R/bench.py $ARGS --emulate-ranks 8 --steps 4 --warmup 1 > /dev/null 2>> $O/err.txt
python $R/tools/rocprof_summary.py $(find /tmp/p_emu -name "*.db" | head -1) --top 70 > $O/kernels_emulate8.txt
rm -rf /tmp/p_one; rocprofv3 --kernel-trace -d /tmp/p_one -o x -- python $R/bench.py $ARGS --steps 4 --warmup 1 > /dev/null 2>> $O/err.txt
python $R/tools/rocprof_summary.py $(find /tmp/p_one -name "*.db" | head -1) --top 50 > $O/kernels_1gpu.txt
python -c "
import json; d=json.loads(open('$O/trace.json').read().strip().splitlines()[-1]); print(d['max_rank_compute_ms'], d['amdahl_terms_ms']); print(json.dumps(d['phase_ms_rank1'], indent=0))"
