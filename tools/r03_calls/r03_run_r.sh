#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_flat_adam_gpu.py tests/test_models_gpu.py tests/test_distributed_gpu.py -q -m gpu > gpurun_out/r03_tests_r.log 2>&1
tail -5 gpurun_out/r03_tests_r.log | cut -c1-300
grep -n "Error\|assert \|FAILED" gpurun_out/r03_tests_r.log | head -30 | cut -c1-300
