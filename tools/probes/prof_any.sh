# usage (on the GPU box): bash tools/probes/prof_any.sh <script.py> [top]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/p; rocprofv3 --kernel-trace --stats -d /tmp/p -o x -- python $R/$1 > /tmp/log.txt 2>&1; tail -2 /tmp/log.txt
python $R/tools/rocprof_summary.py $(find /tmp/p -name "*.db" | head -1) --top ${2:-12} | cut -c1-150
