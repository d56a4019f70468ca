#!/bin/bash
# call ah: launch-by-launch profile of the eager step after the sparse RPN loss path + one lean bench line with the host-issue time
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04g
export MDT_MIOPEN_SKIP_NAIVE=1
LEAN="--no-secondary --no-roofline --no-eager-leg --no-graph-leg --no-exec-leg --no-graph-preflight"
BENCH_ARGS="$LEAN --graph 0" OUT_NAME=r04g/steady_eager_sparse_rpn GLUE_OUT=r04g/glue_breakdown_eager_sparse_rpn.txt bash tools/prof_step.sh 5 400 | head -12
python bench.py --steps 15 --warmup 4 $LEAN --no-cpu-baseline --no-h2d-leg --no-rccl-selftest 2>/dev/null | grep '^{"metric"' | cut -c1-700
