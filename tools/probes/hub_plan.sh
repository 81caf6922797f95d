cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/p; rocprofv3 --kernel-trace --stats -d /tmp/p -o x -- python $R/tools/probes/hub_plan.py > /tmp/log.txt 2>&1; tail -5 /tmp/log.txt
python $R/tools/rocprof_summary.py $(find /tmp/p -name "*.db" | head -1) --top 14 | cut -c1-150
