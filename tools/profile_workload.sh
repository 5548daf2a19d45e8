#!/bin/bash
# rocprofv3 per-kernel durations of one bench.py workload (GPU box):  tools/profile_workload.sh <tag> <bench.py arguments...>
#   -> gpurun_out/<tag>_kernel_stats.csv.  Every step runs under its own timeout; nothing reads stdin.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; D=/tmp/prof_$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $D
timeout 300 rocprofv3 --kernel-trace --stats -d $D --output-format csv -- python $R/bench.py "$@" > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err < /dev/null
f=$(find $D -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ] && [ -f "$f" ]; then cp "$f" $O/${TAG}_kernel_stats.csv; head -12 "$f" | cut -c1-220; else echo "no kernel_stats.csv under $D"; tail -5 $O/${TAG}_bench.err; fi
