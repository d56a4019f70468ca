#!/bin/bash
# round 5: the Retina U-Net's C1 layer in space-to-depth form -- MIOpen find for the two new problems (without the naive solvers: seconds per
# trial on 1.2-TFLOP problems), the probe, the config-2 bench; the find-db comes back under gpurun_out/r05/miopen_cache/db
mkdir -p gpurun_out/r05
export MDT_MIOPEN_SKIP_NAIVE=1
rm -rf gpurun_out/r05/miopen_cache; cp -r medicaldetectiontoolkit_amd/miopen_cache gpurun_out/r05/miopen_cache
export MDT_MIOPEN_CACHE=$PWD/gpurun_out/r05/miopen_cache
timeout 300 python tools/c1_probe.py --iters 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/c1_probe_own.jsonl | tail -14
timeout 330 python bench.py --model retina_unet --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --no-rccl-selftest --no-h2d-leg --no-graph-leg --no-eager-leg --no-dense-rpn-leg --no-exec-leg 2>gpurun_out/r05/retina_bench_s2d.err | tail -1 | tee gpurun_out/r05/retina_bench_s2d.json | cut -c1-500
tail -3 gpurun_out/r05/retina_bench_s2d.err | cut -c1-300
rm -rf gpurun_out/r05/miopen_cache/kernels
