#!/bin/bash
export MIOPEN_LOG_LEVEL=0 MDT_MIOPEN_SKIP_NAIVE=1
mkdir -p gpurun_out/r04p
timeout 900 python -m pytest tests/test_epilogue_gpu.py -q -x -k "conv3x3x3" 2>&1 | grep -v "MIOpen(HIP)" | tail -3 | cut -c1-300
LEAN="--steps 15 --warmup 4 --no-secondary --no-roofline --no-eager-leg --no-graph-leg --no-exec-leg --no-graph-preflight --no-cpu-baseline --no-h2d-leg --no-rccl-selftest"
for rep in 1 2; do for f in 0 1; do
  timeout 300 python bench.py $LEAN --conv3-small-epilogue $f 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('A/B conv3_small epilogue (compile-time switch)=$f', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r04p/r04_ab_conv3_epilogue.txt
done; done
BENCH_ARGS="--no-secondary --no-roofline --no-eager-leg --no-graph-leg --no-exec-leg --no-graph-preflight --graph 0" OUT_NAME=r04p/r04_bench_train_step_steady_state_kernels_eager_final_tree bash tools/prof_step.sh 5 400 | head -3 | cut -c1-200
grep "conv3x3x3_small_kernel\|conv1x1_dgrad" gpurun_out/r04p/r04_bench_train_step_steady_state_kernels_eager_final_tree.csv | cut -c1-150
