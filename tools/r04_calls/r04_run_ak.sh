#!/bin/bash
# call ak: pyramid tap (FPN output gradient assembled channels-last in one node, sampled RPN rows as a sparse gradient) + level-vectorised
# index arithmetic of rpn_at_anchors -- equality / parity tests, same-box A/B.  The tap was NEUTRAL (32.80 vs 32.80 ms: the mixed-layout add it
# removes already did the layout change) and was taken out again (bench.py --pyramid-tap no longer exists); the index arithmetic stayed
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04g
export MDT_MIOPEN_SKIP_NAIVE=1
python -m pytest tests/test_models_gpu.py tests/test_step_parity_gpu.py tests/test_graph_step_gpu.py -x -q 2>&1 | grep -v "MIOpen(HIP)" | tail -4 | cut -c1-250
LEAN="--steps 15 --warmup 4 --no-secondary --no-roofline --no-eager-leg --no-graph-leg --no-exec-leg --no-graph-preflight --no-cpu-baseline --no-h2d-leg --no-rccl-selftest"
one() { timeout 300 python bench.py $LEAN "$@" 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('A/B $*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  one --pyramid-tap 0 | tee -a gpurun_out/r04g/pyramid_tap_ab.txt
  one --pyramid-tap 1 | tee -a gpurun_out/r04g/pyramid_tap_ab.txt
done
one --pyramid-tap 1 --graph 1 | tee -a gpurun_out/r04g/pyramid_tap_ab.txt
