#!/bin/bash
# Run ON the GPU box: cProfile of bench.py (world 1) on a stream 1/8 of the headline — the host-bound regime of a per-rank step.
# usage: host_profile.sh [extra bench args]  -> gpurun_out/host_profile.txt
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
ARGS="--events 1250000 --nodes 62500 --span 1250000 --delta 125000 --steps 60 --warmup 5 --no-cpu-baseline"
python $R/bench.py $ARGS "$@" > $R/gpurun_out/host_line.json 2>/dev/null
python -m cProfile -o /tmp/host.prof $R/bench.py $ARGS "$@" > /dev/null 2>&1
python - <<PY > $R/gpurun_out/host_profile.txt
import pstats, json
line = json.loads(open("$R/gpurun_out/host_line.json").read().strip().splitlines()[-1])
print("ms_per_step (unprofiled):", line["ms_per_step"])
p = pstats.Stats("/tmp/host.prof")
pat = r"pathpyg_amd|bench\.py|torch\._C|built-in method torch|of 'torch|_ctypes|CFuncPtr|optim|autograd"
p.sort_stats("tottime").print_stats(pat, 70)
p.sort_stats("cumtime").print_stats(r"pathpyg_amd|bench\.py|optim|autograd", 70)
p.print_callers("_cuda_getDeviceCount")
p.print_callers("device_count")
p.print_callers("is_available")
PY
head -90 $R/gpurun_out/host_profile.txt | cut -c1-170
