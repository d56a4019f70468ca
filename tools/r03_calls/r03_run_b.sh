#!/bin/bash
# step parity + predictor parity tests, then config 5 (4 records) and the Retina U-Net step profile
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_step_parity_gpu.py tests/test_predictor_parity_gpu.py tests/test_models_gpu.py -q -m gpu > gpurun_out/r03_tests_b.log 2>&1
tail -40 gpurun_out/r03_tests_b.log | cut -c1-400
for amp in bf16 none; do for aug in 0 1; do
  timeout 400 python tools/bench_inference.py --amp $amp --test-aug $aug > gpurun_out/r03_config5_${amp}_aug${aug}.json 2> gpurun_out/r03_config5_${amp}_aug${aug}.err
  tail -1 gpurun_out/r03_config5_${amp}_aug${aug}.json; tail -2 gpurun_out/r03_config5_${amp}_aug${aug}.err | cut -c1-300
done; done
