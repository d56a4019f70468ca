#!/bin/bash
mkdir -p gpurun_out/r04
export MDT_MIOPEN_SKIP_NAIVE=1
run() { timeout 200 env $ENVV python tools/graph_small_bisect.py "$@" > gpurun_out/r04/sb.log 2>&1; echo "[$ENVV] $* rc=$?"; grep -v "Warning\|amdgpu.ids\|^  \|Extension modules\|^$\|dumped core" gpurun_out/r04/sb.log | tail -2 | cut -c1-200; }
ENVV="" run rpnloss_only
ENVV="" run rpnloss_only
ENVV="AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3" run rpnloss_only
ENVV="DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" run rpnloss_only
ENVV="HIP_LAUNCH_BLOCKING=1" run rpnloss_only
ENVV="DEBUG_HIP_FORCE_GRAPH_QUEUES=1" run rpnloss_only
ENVV="GPU_MAX_HW_QUEUES=1" run rpnloss_only
