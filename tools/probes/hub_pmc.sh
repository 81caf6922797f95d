#!/bin/bash
# Run ON the GPU box: kernel statistics + HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) of the fused builder on the two hub streams
# (tools/probes/hub_one.py zipf | contact) -> gpurun_out/hub_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hub_pmc.txt
{
for S in zipf contact; do
  echo "######## $S: rocprofv3 --kernel-trace --stats (3 builder calls + setup)"
  rm -rf /tmp/p_h; rocprofv3 --kernel-trace --stats -d /tmp/p_h -o x -- python $R/tools/probes/hub_one.py $S > /dev/null 2>&1
  python $R/tools/rocprof_summary.py $(find /tmp/p_h -name "*.db" | head -1) --top 16 | cut -c1-150
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "######## $S: rocprofv3 --pmc $C --kernel-trace (KiB; gfx950: double FETCH_SIZE for bytes)"
    rm -rf /tmp/p_h; rocprofv3 --pmc $C --kernel-trace -d /tmp/p_h -o x -- python $R/tools/probes/hub_one.py $S > /dev/null 2>&1
    python $R/tools/rocprof_pmc.py $(find /tmp/p_h -name "*.db" | head -1) --top 8 | cut -c1-170
  done
done
} > $O 2>&1
head -80 $O
