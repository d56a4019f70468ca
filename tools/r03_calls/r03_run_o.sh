#!/bin/bash
# round 3, call O: fused stem bias+ReLU test, host-side issue time (resident vs numpy batches), op profile with the small-op table
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_epilogue_gpu.py tests/test_backbone_parity_gpu.py -x -q -m gpu -k "stem or backbone" > gpurun_out/r03_tests_o.log 2>&1
rc=$?; tail -4 gpurun_out/r03_tests_o.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "TESTS FAILED"; tail -60 gpurun_out/r03_tests_o.log | cut -c1-400; fi
timeout 240 python tools/host_issue_probe.py > gpurun_out/r03_host_issue_probe.json 2> gpurun_out/r03_host_issue_probe.err
cat gpurun_out/r03_host_issue_probe.json; tail -3 gpurun_out/r03_host_issue_probe.err
timeout 240 python tools/op_profile.py 2 > gpurun_out/r03_op_profile_b.txt 2>&1
tail -125 gpurun_out/r03_op_profile_b.txt | cut -c1-200
