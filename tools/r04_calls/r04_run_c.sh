#!/bin/bash
# round 4, call c: whole-capture of the FPN / FPN+RPN segment (replay cost with MIOpen + own kernels as nodes); whole-step capture with faulthandler
mkdir -p gpurun_out/r04
for what in fpn fpn_rpn; do
  timeout 400 python -X faulthandler tools/graph_segment_whole_probe.py 10 $what 2> gpurun_out/r04/graph_segment_whole_$what.err | tee -a gpurun_out/r04/graph_segment_whole.jsonl
  grep -v Warning gpurun_out/r04/graph_segment_whole_$what.err | tail -15 | cut -c1-300
done
timeout 600 python -X faulthandler tools/graph_step_probe.py 10 > gpurun_out/r04/graph_step_probe2.json 2> gpurun_out/r04/graph_step_probe2.err
echo "step probe rc=$?"; cat gpurun_out/r04/graph_step_probe2.json; grep -v Warning gpurun_out/r04/graph_step_probe2.err | tail -70 | cut -c1-300
