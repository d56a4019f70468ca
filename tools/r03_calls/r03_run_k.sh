#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_epilogue_gpu.py -x -q -m gpu -k "conv3x3x3 or conv1x1 or stem" > gpurun_out/r03_tests_k.log 2>&1
rc=$?; tail -6 gpurun_out/r03_tests_k.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "TESTS FAILED"; tail -40 gpurun_out/r03_tests_k.log | cut -c1-300; exit 1; fi
timeout 120 python tools/conv3_probe.py 2>&1 | grep "^{" | tee gpurun_out/r03_conv3_probe.jsonl
