#!/bin/bash
# round 4, call e: micro-bisect of the capture failure in compute_rpn_losses; each piece again in its own process if the first run dies
mkdir -p gpurun_out/r04
timeout 300 python -X faulthandler tools/graph_micro_bisect.py > gpurun_out/r04/micro_bisect.log 2> gpurun_out/r04/micro_bisect.err
echo "rc=$?"; cat gpurun_out/r04/micro_bisect.log; grep -v "Warning\|amdgpu.ids" gpurun_out/r04/micro_bisect.err | head -12 | cut -c1-300
for p in p_rand_topk p_gather_ce p_softmax_topk30 p_arange_cmp p_zeros_long_ce p_index_f64 p_delta_targets p_smooth_l1 p_sum_mean; do
  timeout 120 python tools/graph_micro_bisect.py $p > gpurun_out/r04/micro_$p.log 2>&1; echo "$p rc=$?"
done
