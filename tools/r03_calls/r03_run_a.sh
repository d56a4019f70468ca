#!/bin/bash
# full gpu suite + the bench line + the steady-state profile of the step
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gpu_suite.log 2>&1
tail -6 gpurun_out/r03_gpu_suite.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r03_bench_line_a.json 2> gpurun_out/r03_bench_a.err
tail -c 6000 gpurun_out/r03_bench_line_a.json; tail -3 gpurun_out/r03_bench_a.err
OUT_NAME=r03_step_steady_state_a bash tools/prof_step.sh 5 400 | head -45
