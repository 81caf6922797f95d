#!/bin/bash
# Run ON the GPU box: kernel trace of the 8-rank emulation -> kernels and GPU-busy time per rank and step (how much of a rank's turn is
# host / launch overhead rather than kernels).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/emu; mkdir -p $O
RANKS=${1:-8}
rm -rf /tmp/p_emu; rocprofv3 --kernel-trace -d /tmp/p_emu -o x -- python $R/bench.py --emulate-ranks $RANKS --steps 10 --warmup 2 --no-cpu-baseline > $O/emu$RANKS.json 2> $O/emu$RANKS.err
python $R/tools/rocprof_summary.py $(find /tmp/p_emu -name "*.db" | head -1) --top 60 > $O/emu${RANKS}_kernels.txt
tail -3 $O/emu${RANKS}_kernels.txt
python -c "
import json; d=json.load(open('$O/emu$RANKS.json')); print(d['max_rank_compute_ms'], d['amdahl_terms_ms'])"
