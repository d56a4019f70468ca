#!/bin/bash
# round-3 closing evidence (lean: the RoIAlign kernel-stats / PMC passes of tools/r03_profile_all.sh are unchanged since run 2):
# full gpu suite, bench line (incl. secondary configs, roofline variants, RCCL self-test, host-batch leg), steady-state step profile
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r03_gpu_suite_final.log 2>&1
tail -4 gpurun_out/r03_gpu_suite_final.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_line_final.json 2> gpurun_out/r03_bench_final.err
wc -l gpurun_out/r03_bench_line_final.json; tail -c 6000 gpurun_out/r03_bench_line_final.json; tail -2 gpurun_out/r03_bench_final.err
BENCH_ARGS="--no-secondary --no-roofline" OUT_NAME=r03_step_steady_state_final bash tools/prof_step.sh 5 300 | head -14
