#!/bin/bash
# refresh the in-tree MIOpen cache with the find results of configs 2 and 5, profile the Retina U-Net step, try the bench's secondary leg
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export MDT_MIOPEN_SKIP_NAIVE=1
MDT_MIOPEN_CACHE_INPLACE=1 timeout 500 python bench.py --model retina_unet --steps 3 --warmup 2 --no-cpu-baseline --no-h2d-leg --no-rccl-selftest --no-secondary --no-roofline 2>gpurun_out/r03_retina_find.err | tail -1 | cut -c1-400
MDT_MIOPEN_CACHE_INPLACE=1 timeout 300 python tools/bench_inference.py --amp bf16 --test-aug 0 --repeats 1 2>/dev/null | tail -1 | cut -c1-300
MDT_MIOPEN_CACHE_INPLACE=1 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-h2d-leg --no-rccl-selftest --no-secondary --no-roofline 2>/dev/null | tail -1 | cut -c1-200
rm -rf gpurun_out/miopen_cache_new; cp -r medicaldetectiontoolkit_amd/miopen_cache gpurun_out/miopen_cache_new; du -sh gpurun_out/miopen_cache_new
BENCH_ARGS="--model retina_unet" OUT_NAME=r03_retina_unet_step_kernels bash tools/prof_step.sh 3 500 | head -60
