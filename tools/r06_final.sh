#!/bin/bash
# round-6 closing evidence run on the GPU box -> gpurun_out/r06/ (copied to profiles/r06/ by the builder): suite, driver-style bench line,
# steady-state profile + launch listing, launch sites per source line
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r06
export MDT_MIOPEN_SKIP_NAIVE=1
( time python -m pytest tests -m gpu -q --tb=short ) 2>&1 | tail -14 > gpurun_out/r06/r06_gpu_test_suite_tail.txt
tail -6 gpurun_out/r06/r06_gpu_test_suite_tail.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/r06_bench_line_1gpu_final.json 2> gpurun_out/r06/r06_bench_final.err
python tools/fill_r06_numbers.py gpurun_out/r06/r06_bench_line_1gpu_final.json --print
LEAN="--no-secondary --no-roofline --no-graph-leg --no-exec-leg"
BENCH_ARGS="$LEAN" OUT_NAME=r06/r06_bench_train_step_steady_state_kernels_eager_final GLUE_OUT=r06/r06_step_launch_by_launch_final.txt bash tools/prof_step.sh 5 400 | head -9
python tools/launch_sites.py 3 > gpurun_out/r06/r06_launch_sites_after.txt 2> /dev/null
head -3 gpurun_out/r06/r06_step_launch_by_launch_final.txt
