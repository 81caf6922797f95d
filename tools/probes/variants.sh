#!/bin/bash
# Run ON the GPU box: time library variants tools/probes/_bin/lib_<tag>.so with bench_kernels.  usage: variants.sh "<tags>" "<bench_kernels args>" [grep]
cp pathpyg_amd/lib/libpathpyg_amd.so /tmp/keep.so
for rep in 1 2; do for T in $1; do
  cp tools/probes/_bin/lib_$T.so pathpyg_amd/lib/libpathpyg_amd.so
  echo "== $T"; timeout 300 python tools/bench_kernels.py $2 2>&1 | grep -i "${3:-median}"
done; done
cp /tmp/keep.so pathpyg_amd/lib/libpathpyg_amd.so
