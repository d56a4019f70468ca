#!/bin/bash
# as the driver runs it at round end: the GPU suite with -x, smoke(), the bench line
mkdir -p gpurun_out/r04f
SECONDS=0
python -m pytest tests/ -x -q -m gpu > gpurun_out/r04f/driver_style_suite.log 2>&1; echo "suite rc=$? wall=${SECONDS}s"; grep -v "MIOpen(HIP)" gpurun_out/r04f/driver_style_suite.log | tail -3 | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids" | tail -1
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04f/driver_style_bench.json 2> gpurun_out/r04f/driver_style_bench.err; echo "bench rc=$? wall=${SECONDS}s"
python -c "
import json; d = json.load(open('gpurun_out/r04f/driver_style_bench.json')); print({k: d[k] for k in ('value','ms_per_step','steps','warmup','n_gpus')}, d['graph'], d['graphed_step']['value'], d['exec_equivalent']['value'], d['roofline']['frac'], {k: v.get('wall_s') for k, v in d['secondary'].items()})"
