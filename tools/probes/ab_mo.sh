#!/bin/bash
# Run ON the GPU box: same-box A/B of the multi-order builder, built library against tools/probes/_bin/libold.so.   usage: bash tools/probes/ab_mo.sh [rounds]
cp pathpyg_amd/lib/libpathpyg_amd.so /tmp/new.so
for i in $(seq 1 ${1:-2}); do for L in new old; do
  if [ $L = old ]; then cp tools/probes/_bin/libold.so pathpyg_amd/lib/libpathpyg_amd.so; else cp /tmp/new.so pathpyg_amd/lib/libpathpyg_amd.so; fi
  echo "== LIB $L"
  timeout 120 python tools/probes/multi_order_k.py 5 4 2>&1 | tail -2 | cut -c1-40
  timeout 120 python tools/probes/multi_order_k.py 3 4 2>&1 | tail -1 | cut -c1-40
  timeout 200 python tools/probes/config2_k3.py 1500000 3 2>&1 | tail -1 | cut -c1-40
done; done
cp /tmp/new.so pathpyg_amd/lib/libpathpyg_amd.so
