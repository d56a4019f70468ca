#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_hip_gpu.py tests/test_pyramid_roialign_gpu.py -x -q -m gpu -k "roialign or pyramid" > gpurun_out/r03_tests3.log 2>&1
rc=$?
tail -12 gpurun_out/r03_tests3.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc -- skipping the timing runs"; exit 1; fi
timeout 60 python tools/bwd3_probe.py 2>&1 | tee gpurun_out/r03_bwd3_probe_b.jsonl | cut -c1-600
export MDT_BWD_KERNEL=v3
for spec in "bwd_fast P2 trainlike" "bwd_fast P2 random" "bwd_fast P3 trainlike" "bwd_fast P5 trainlike" "pyramid_bwd P2 trainlike"; do
  set -- $spec
  export MDT_LEVEL=$2 MDT_ROIS=$3
  echo "== v3b $spec"
  bash tools/gpu_prof.sh $1 40 | head -1
  rm -rf gpurun_out/prof_$1_$2_$3_v3b; mv gpurun_out/prof_$1 gpurun_out/prof_$1_$2_$3_v3b 2>/dev/null
done
