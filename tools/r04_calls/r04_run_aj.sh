#!/bin/bash
# call aj: stride tap (stage outputs sub-sampled once, gradients of the three consumers meet in one node) -- backbone / step parity, same-box A/B
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04g
export MDT_MIOPEN_SKIP_NAIVE=1
python -m pytest tests/test_backbone_parity_gpu.py tests/test_step_parity_gpu.py tests/test_epilogue_gpu.py tests/test_graph_step_gpu.py -x -q 2>&1 | grep -v "MIOpen(HIP)" | tail -4 | cut -c1-250
LEAN="--steps 15 --warmup 4 --no-secondary --no-roofline --no-eager-leg --no-graph-leg --no-exec-leg --no-graph-preflight --no-cpu-baseline --no-h2d-leg --no-rccl-selftest"
one() { timeout 300 python bench.py $LEAN "$@" 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('A/B $*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  one --stride-tap 0 | tee -a gpurun_out/r04g/stride_tap_ab.txt
  one --stride-tap 1 | tee -a gpurun_out/r04g/stride_tap_ab.txt
done
