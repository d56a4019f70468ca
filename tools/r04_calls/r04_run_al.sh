#!/bin/bash
# call al: channels-last nearest up-sampling in the FPN top-down path -- backbone / step parity, same-box A/B, launch-by-launch profile
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04g
export MDT_MIOPEN_SKIP_NAIVE=1
python -m pytest tests/test_backbone_parity_gpu.py tests/test_step_parity_gpu.py tests/test_graph_step_gpu.py -x -q 2>&1 | grep -v "MIOpen(HIP)" | tail -4 | cut -c1-250
LEAN="--steps 15 --warmup 4 --no-secondary --no-roofline --no-eager-leg --no-graph-leg --no-exec-leg --no-graph-preflight --no-cpu-baseline --no-h2d-leg --no-rccl-selftest"
one() { timeout 300 python bench.py $LEAN "$@" 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('A/B $*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  one --upsample-nearest-cl 0 | tee -a gpurun_out/r04g/upsample_nearest_cl_ab.txt
  one --upsample-nearest-cl 1 | tee -a gpurun_out/r04g/upsample_nearest_cl_ab.txt
done
LEAN2="--no-secondary --no-roofline --no-eager-leg --no-graph-leg --no-exec-leg --no-graph-preflight"
BENCH_ARGS="$LEAN2 --graph 0" OUT_NAME=r04g/steady_eager_after_taps GLUE_OUT=r04g/glue_breakdown_eager_after_taps.txt bash tools/prof_step.sh 5 400 | head -10
sed -n '/every OTHER launch/,$p' gpurun_out/r04g/glue_breakdown_eager_after_taps.txt | cut -c1-150
