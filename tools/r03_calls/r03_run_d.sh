#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_epilogue_gpu.py -x -q -m gpu -k wgrad > gpurun_out/r03_tests_d.log 2>&1
rc=$?; tail -4 gpurun_out/r03_tests_d.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "TESTS FAILED"; exit 1; fi
for wgs in 2 4 8; do
  echo "== MDT_WGRAD_WGS_PER_CU=$wgs"
  MDT_WGRAD_WGS_PER_CU=$wgs timeout 200 python tools/wgrad_probe.py 2>&1 | grep "^{" | cut -c1-200
done
