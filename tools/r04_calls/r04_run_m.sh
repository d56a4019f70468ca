#!/bin/bash
mkdir -p gpurun_out/r04
export MDT_MIOPEN_SKIP_NAIVE=1
run() { timeout 200 python tools/graph_small_bisect.py "$@" > gpurun_out/r04/sb.log 2>&1; echo "$* rc=$?"; grep -v "Warning\|amdgpu.ids\|^  \|Extension modules\|^$\|dumped core" gpurun_out/r04/sb.log | tail -3 | cut -c1-200; }
run step_nobwd seed
run step_nobwd alloc
run targets_only
run heads_only
run rpnloss_only
run targets_only seed
run heads_only seed
