#!/bin/bash
# round-3 first GPU call: new tests + where the RoIAlign-backward time goes (stage stamps, rocprof per level)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_distributed_gpu.py tests/test_hip_gpu.py tests/test_pyramid_roialign_gpu.py -x -q -m gpu > gpurun_out/r03_tests1.log 2>&1
tail -5 gpurun_out/r03_tests1.log
timeout 120 python tools/bwd_stage_probe.py > gpurun_out/r03_stage_probe.jsonl 2>&1
cat gpurun_out/r03_stage_probe.jsonl
for spec in "bwd_fast P2 trainlike" "bwd_fast P2 random" "bwd_fast P3 trainlike" "bwd_fast P5 trainlike" "pyramid_bwd P2 trainlike" "fwd P2 trainlike"; do
  set -- $spec
  export MDT_LEVEL=$2 MDT_ROIS=$3
  echo "== $spec"
  bash tools/gpu_prof.sh $1 40 | head -4
  mv gpurun_out/prof_$1 gpurun_out/prof_$1_$2_$3 2>/dev/null
done
