#!/bin/bash
# gather-form backward: parity tests, then rocprof kernel time per case (and the round-2 kernel on the same box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_gpu.py tests/test_pyramid_roialign_gpu.py -x -q -m gpu -k "roialign or pyramid" > gpurun_out/r03_tests2.log 2>&1
tail -15 gpurun_out/r03_tests2.log
for kern in v3 r2; do
  export MDT_BWD_KERNEL=$kern
  for spec in "bwd_fast P2 trainlike" "bwd_fast P2 random" "bwd_fast P3 trainlike" "bwd_fast P5 trainlike" "pyramid_bwd P2 trainlike"; do
    set -- $spec
    export MDT_LEVEL=$2 MDT_ROIS=$3
    echo "== $kern $spec"
    bash tools/gpu_prof.sh $1 40 | head -2
    rm -rf gpurun_out/prof_$1_$2_$3_$kern; mv gpurun_out/prof_$1 gpurun_out/prof_$1_$2_$3_$kern 2>/dev/null
  done
done
