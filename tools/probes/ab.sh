#!/bin/bash
# Run ON the GPU box: same-box A/B of the built library against tools/probes/_bin/libold.so.  usage: bash tools/probes/ab.sh "<bench_kernels args>" [grep pattern] [rounds]
cp pathpyg_amd/lib/libpathpyg_amd.so /tmp/new.so
for i in $(seq 1 ${3:-2}); do for L in new old; do
  if [ $L = old ]; then cp tools/probes/_bin/libold.so pathpyg_amd/lib/libpathpyg_amd.so; else cp /tmp/new.so pathpyg_amd/lib/libpathpyg_amd.so; fi
  echo "== LIB $L"
  [ -n "$1" ] && timeout 300 python tools/bench_kernels.py $1 2>&1 | grep -i "${2:-median}"
  timeout 120 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('step', round(d['ms_per_step'],3), 'dbgnn', round(d['dbgnn_step_ms'],3), 'lift', round(d['lift_ms'],3)); [print('   ', k['kernel'][:58], round(k['avg_launch_ms'],3)) for k in d['kernel_rooflines'][:4]]"
done; done
cp /tmp/new.so pathpyg_amd/lib/libpathpyg_amd.so
