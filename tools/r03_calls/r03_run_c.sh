#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_epilogue_gpu.py -x -q -m gpu > gpurun_out/r03_tests_c.log 2>&1
rc=$?; tail -8 gpurun_out/r03_tests_c.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "TESTS FAILED"; exit 1; fi
timeout 200 python tools/wgrad_probe.py 2>&1 | grep "^{" | tee gpurun_out/r03_wgrad_probe.jsonl
for flag in 0 1 0 1; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-rccl-selftest --no-secondary --no-roofline --wgrad-1x1 $flag 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('wgrad_1x1=$flag', d['value'], d['ms_per_step'])"
done
