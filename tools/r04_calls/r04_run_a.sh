#!/bin/bash
# round 4, call a: whole-step hipGraph capture probe + the segment probe of round 3 (make_graphed_callables)
mkdir -p gpurun_out/r04
timeout 600 python tools/graph_step_probe.py 10 > gpurun_out/r04/graph_step_probe.json 2> gpurun_out/r04/graph_step_probe.err
echo "step probe rc=$?"; cat gpurun_out/r04/graph_step_probe.json; tail -5 gpurun_out/r04/graph_step_probe.err
timeout 400 python tools/graph_probe.py 10 > gpurun_out/r04/graph_segment_probe.json 2> gpurun_out/r04/graph_segment_probe.err
echo "segment probe rc=$?"; cat gpurun_out/r04/graph_segment_probe.json; tail -5 gpurun_out/r04/graph_segment_probe.err
