#!/bin/bash
# Run HERE after `gpurun -- bash tools/collect_profiles.sh` (+ tools/bench_kernels.py): copies the summaries from gpurun_out/ into
# profiles/rNN_* with the header lines that say which command and commit they come from.   usage: bash tools/install_profiles.sh r01
R=${1:-r01}; C=$(git rev-parse --short HEAD); O=gpurun_out/profiles; CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
{ echo "# rocprofv3 --kernel-trace --stats -- $CMD   (MI355X, round ${R#r}, commit $C)"; echo "# summarised on the GPU box by tools/rocprof_summary.py (tools/collect_profiles.sh); totals cover setup + 7 steps + 3 untimed k=3 lifts"; cat $O/bench_kernel_stats.txt; } > profiles/${R}_bench_kernel_stats.txt
{ echo "# rocprofv3 --pmc FETCH_SIZE --kernel-trace -- $CMD   (separate pass; KiB; commit $C)"; echo "# gfx950: FETCH_SIZE tallies 64 B per 128-B request -> double it for byte counts (MI355X_MICROARCH.md, HBM)"; cat $O/pmc_FETCH_SIZE.txt; } > profiles/${R}_pmc_fetch_size.txt
{ echo "# rocprofv3 --pmc WRITE_SIZE --kernel-trace -- $CMD   (separate pass; KiB; commit $C)"; cat $O/pmc_WRITE_SIZE.txt; } > profiles/${R}_pmc_write_size.txt
cp $O/pmc_traffic.json profiles/${R}_pmc_traffic.json; cp $O/bench_line.json profiles/${R}_bench_line.json
{ echo "# python tools/bench_kernels.py --ops lift,agg,plan,spmm,dense,gcn  (per-op medians over 10 runs, HIP events; headline workload m=10^7, N=5*10^5, F=64; MI355X, commit $C)"; cat gpurun_out/bench_kernels.txt; } > profiles/${R}_per_op_timings.txt
