#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export MDT_MIOPEN_SKIP_NAIVE=1
timeout 300 python -m pytest tests/test_epilogue_gpu.py -x -q -m gpu -k "upsample" > gpurun_out/r03_tests_g.log 2>&1
rc=$?; tail -4 gpurun_out/r03_tests_g.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "TESTS FAILED"; exit 1; fi
timeout 120 python -m pytest tests/test_step_parity_gpu.py tests/test_models_gpu.py -x -q -m gpu 2>&1 | tail -2
for flag in 0 1; do
  MDT_MIOPEN_CACHE_INPLACE=1 timeout 400 python bench.py --model retina_unet --steps 3 --warmup 2 --no-cpu-baseline --no-h2d-leg --no-rccl-selftest --no-secondary --no-roofline --upsample-cl $flag 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('upsample_cl=$flag', d['value'], d['ms_per_step'])"
done
rm -rf gpurun_out/miopen_cache_new; cp -r medicaldetectiontoolkit_amd/miopen_cache gpurun_out/miopen_cache_new
BENCH_ARGS="--model retina_unet" OUT_NAME=r03_retina_unet_step_kernels bash tools/prof_step.sh 3 500 | sed -n 2,8p
grep -E "upsample|manual_unroll" gpurun_out/r03_retina_unet_step_kernels.csv | cut -c1-160
