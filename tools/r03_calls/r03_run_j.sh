#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_epilogue_gpu.py tests/test_glue_parity_gpu.py tests/test_step_parity_gpu.py tests/test_golden_gpu.py -x -q -m gpu > gpurun_out/r03_tests_j.log 2>&1
rc=$?; tail -4 gpurun_out/r03_tests_j.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "TESTS FAILED"; tail -30 gpurun_out/r03_tests_j.log | cut -c1-300; exit 1; fi
timeout 200 python tools/wgrad_probe.py 2>&1 | grep "^{" | tee gpurun_out/r03_wgrad_probe_db.jsonl | cut -c1-200
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-rccl-selftest --no-secondary --no-roofline 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('bench', d['value'], d['ms_per_step'])"
