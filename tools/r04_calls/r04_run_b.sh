#!/bin/bash
# round 4, call b: cost of a hipGraph replay per kernel node under the runtime's graph flags; whole-step capture with faulthandler
mkdir -p gpurun_out/r04
for envs in "" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 HIP_FORCE_DEV_KERNARG=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1"; do
  env $envs timeout 200 python tools/graph_launch_probe.py 1000 2>/dev/null | tee -a gpurun_out/r04/graph_launch_probe.jsonl
done
timeout 600 python -X faulthandler tools/graph_step_probe.py 10 > gpurun_out/r04/graph_step_probe2.json 2> gpurun_out/r04/graph_step_probe2.err
echo "step probe rc=$?"; cat gpurun_out/r04/graph_step_probe2.json; grep -v Warning gpurun_out/r04/graph_step_probe2.err | tail -60 | cut -c1-300
