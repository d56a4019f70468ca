#!/bin/bash
# usage: tools/gpu_prof.sh <case> [iters]   -> rocprofv3 kernel stats csv under gpurun_out/prof_<case>/
set -u
CASE=$1; IT=${2:-30}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$CASE -o $CASE -- python $ROOT/tools/profile_case.py $CASE $IT > $ROOT/gpurun_out/prof_$CASE.log 2>&1 < /dev/null
F=$(find $ROOT/gpurun_out/prof_$CASE -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then python $ROOT/tools/kstats.py "$F"; else echo "no stats csv"; tail -5 $ROOT/gpurun_out/prof_$CASE.log; fi
