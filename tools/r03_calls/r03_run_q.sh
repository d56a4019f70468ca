#!/bin/bash
# round 3, call Q: FlatAdam + raw stream getter + numpy staging: tests, A/B in the step, host-batch rate
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_flat_adam_gpu.py tests/test_models_gpu.py tests/test_distributed_gpu.py tests/test_step_parity_gpu.py -x -q -m gpu > gpurun_out/r03_tests_q.log 2>&1
rc=$?; tail -4 gpurun_out/r03_tests_q.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "TESTS FAILED"; tail -60 gpurun_out/r03_tests_q.log | cut -c1-400; fi
for flag in 0 1; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-rccl-selftest --no-secondary --no-roofline --flat-adam $flag 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('A/B flat_adam=$flag', d['value'], d['ms_per_step'], 'host issue', d['host_issue_ms_per_step'])"
done 2>&1 | tee gpurun_out/r03_ab_flat_adam.txt
timeout 200 python bench.py --steps 8 --warmup 3 --host-batches --no-cpu-baseline --no-rccl-selftest --no-secondary --no-roofline 2>gpurun_out/r03_host_batches_c.err | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('host-batches (numpy staging, pinned ring)', d['value'], d['ms_per_step'], 'host issue', d['host_issue_ms_per_step'])" | tee gpurun_out/r03_host_batches_c.txt
tail -3 gpurun_out/r03_host_batches_c.err
