#!/bin/bash
# zero role in isolation: prologue cost and sensitivity to the number of zero workgroups (rocprof kernel time)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export MDT_LEVEL=P2 MDT_ROIS=trainlike MDT_BWD_TUNE=1
for spec in "1 0" "9 0" "1 256" "1 512" "1 768" "1 1024" "9 512" "2 0" "0 0" "0 160" "0 192" "0 256"; do
  set -- $spec
  export MDT_BWD3_DBG=$1 MDT_BWD3_ZERO_WGS=$2
  echo "== dbg=$1 zero_wgs=$2"
  bash tools/gpu_prof.sh bwd_fast 30 | head -1
done
