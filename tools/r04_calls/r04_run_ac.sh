#!/bin/bash
# final-tree steady-state profile of the eager headline step (merged RPN heads, residual tap, conv3 epilogue, batched matching)
mkdir -p gpurun_out/r04p
export MDT_MIOPEN_SKIP_NAIVE=1
LEAN="--no-secondary --no-roofline --no-eager-leg --no-graph-leg --no-exec-leg --no-graph-preflight"
BENCH_ARGS="$LEAN --graph 0" OUT_NAME=r04p/r04_bench_train_step_steady_state_kernels_eager_final_tree bash tools/prof_step.sh 5 400 | head -12
sed -n 20,60p gpurun_out/r04p/r04_bench_train_step_steady_state_kernels_eager_final_tree.csv | cut -c1-150
