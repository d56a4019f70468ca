#!/bin/bash
mkdir -p gpurun_out/r04
export MDT_MIOPEN_SKIP_NAIVE=1
run() { timeout 200 python tools/graph_small_bisect.py "$@" > gpurun_out/r04/sb.log 2>&1; echo "$* rc=$?"; grep -v "Warning\|amdgpu.ids\|^  \|Extension modules\|^$\|dumped core" gpurun_out/r04/sb.log | tail -3 | cut -c1-200; }
run fpn_fwd
run fpn_fwdbwd
run fpn_rpn_fwdbwd
run forward_nograd
run step_nobwd
run step_full
run step_full fe0
run step_full asfwd0
run step_full stem0
run step_full pool0
run step_full nobench
run step_full nocl
