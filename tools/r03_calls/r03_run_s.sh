#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 120 python -m pytest tests/test_flat_adam_gpu.py -q -m gpu > gpurun_out/r03_tests_s.log 2>&1
tail -3 gpurun_out/r03_tests_s.log | cut -c1-300
grep -n "Error\|FAILED" gpurun_out/r03_tests_s.log | head -20 | cut -c1-300
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-secondary --no-roofline 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('flat adam, gathered grads', d['value'], d['ms_per_step'], 'host issue', d['host_issue_ms_per_step'], d['distributed'].get('rccl_world1_selftest'))" | tee gpurun_out/r03_flat_adam_gather.txt
