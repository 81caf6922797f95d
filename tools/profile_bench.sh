cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/p; rocprofv3 --kernel-trace --stats -d /tmp/p -o x -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /tmp/log.txt 2>&1
tail -1 /tmp/log.txt | cut -c1-200
DB=$(find /tmp/p -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB --top 45 > $R/gpurun_out/kernel_stats.txt 2>&1
head -50 $R/gpurun_out/kernel_stats.txt | cut -c1-150
