#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_models_gpu.py tests/test_step_parity_gpu.py tests/test_glue_parity_gpu.py -x -q -m gpu > gpurun_out/r03_tests_h.log 2>&1
rc=$?; tail -4 gpurun_out/r03_tests_h.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "TESTS FAILED"; tail -30 gpurun_out/r03_tests_h.log | cut -c1-300; exit 1; fi
for flag in 0 1 0 1; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-rccl-selftest --no-secondary --no-roofline --head-as-linear $flag 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('head_as_linear=$flag', d['value'], d['ms_per_step'])"
done
