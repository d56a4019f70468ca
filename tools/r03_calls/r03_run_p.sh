#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 200 python bench.py --steps 6 --warmup 3 --host-batches --no-cpu-baseline --no-rccl-selftest --no-secondary --no-roofline 2>gpurun_out/r03_host_batches_b.err | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('host-batches (pinned ring)', d['value'], d['ms_per_step'], 'host issue', d['host_issue_ms_per_step'])" | tee gpurun_out/r03_host_batches_b.txt
tail -3 gpurun_out/r03_host_batches_b.err
timeout 200 python -m pytest tests/test_step_parity_gpu.py tests/test_models_gpu.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-200
timeout 240 python tools/host_profile.py > gpurun_out/r03_host_profile.txt 2>&1
cut -c1-170 gpurun_out/r03_host_profile.txt | tail -130
