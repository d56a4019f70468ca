#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_flat_adam_gpu.py -q -x -m gpu 2>&1 | tail -2 | cut -c1-200
timeout 100 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-h2d-leg --no-secondary --no-roofline --no-rccl-selftest 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('final tree', d['value'], d['ms_per_step'], 'host issue', d['host_issue_ms_per_step'])"
