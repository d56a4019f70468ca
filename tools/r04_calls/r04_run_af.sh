#!/bin/bash
# call af: per-call breakdown of the eager step's non-convolution kernels and of the convolution kernels by grid (tools/glue_breakdown.py)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04g
export MDT_MIOPEN_SKIP_NAIVE=1
LEAN="--no-secondary --no-roofline --no-eager-leg --no-graph-leg --no-exec-leg --no-graph-preflight"
BENCH_ARGS="$LEAN --graph 0" OUT_NAME=r04g/steady_eager GLUE_OUT=r04g/glue_breakdown_eager.txt bash tools/prof_step.sh 5 400 | head -12
sed -n '/last step in time order/,$p' gpurun_out/r04g/glue_breakdown_eager.txt | cut -c1-150
