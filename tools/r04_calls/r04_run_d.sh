#!/bin/bash
# round 4, call d: bisect the capture_end crash of the whole-step capture
mkdir -p gpurun_out/r04
for mode in proposals forward forward_masks targets heads match losses full_nobwd full; do
  timeout 300 python -X faulthandler tools/graph_bisect_probe.py $mode 2> gpurun_out/r04/bisect_$mode.err | tee -a gpurun_out/r04/graph_bisect.jsonl
  echo "mode $mode rc=${PIPESTATUS[0]}" | tee -a gpurun_out/r04/graph_bisect.jsonl
done
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 300 python -X faulthandler tools/graph_bisect_probe.py full 2> gpurun_out/r04/bisect_full_nopc.err | tee -a gpurun_out/r04/graph_bisect.jsonl
echo "mode full (packet capture off) rc=${PIPESTATUS[0]}" | tee -a gpurun_out/r04/graph_bisect.jsonl
