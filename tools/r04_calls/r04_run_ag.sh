#!/bin/bash
# call ag: RPN losses through the sampled anchors only (models/mrcnn.rpn_at_anchors) -- equality with the dense-graph step, parity with the
# reference step goldens, graphed step, same-box A/B
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04g
export MDT_MIOPEN_SKIP_NAIVE=1
python -m pytest tests/test_models_gpu.py tests/test_step_parity_gpu.py tests/test_graph_step_gpu.py tests/test_glue_parity_gpu.py -x -q 2>&1 | grep -v "MIOpen(HIP)" | tail -6 | cut -c1-250
LEAN="--steps 15 --warmup 4 --no-secondary --no-roofline --no-eager-leg --no-graph-leg --no-exec-leg --no-graph-preflight --no-cpu-baseline --no-h2d-leg --no-rccl-selftest"
one() { timeout 300 python bench.py $LEAN "$@" 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('A/B $*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  one --sparse-rpn-loss 0 | tee -a gpurun_out/r04g/sparse_rpn_ab.txt
  one --sparse-rpn-loss 1 | tee -a gpurun_out/r04g/sparse_rpn_ab.txt
done
one --sparse-rpn-loss 1 --graph 1 | tee -a gpurun_out/r04g/sparse_rpn_ab.txt
