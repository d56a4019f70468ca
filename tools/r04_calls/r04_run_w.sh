#!/bin/bash
mkdir -p gpurun_out/r04f
export MIOPEN_LOG_LEVEL=0
timeout 1200 python -m pytest tests/test_epilogue_gpu.py -q -x --durations=8 2>&1 | grep -v "MIOpen(HIP)" | tail -16 | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids" | tail -2
/usr/bin/time -v timeout 1500 python bench.py > gpurun_out/r04f/bench_default.json 2> gpurun_out/r04f/bench_default.err
echo "bench default rc=$?"; grep "Elapsed (wall" gpurun_out/r04f/bench_default.err; python -c "
import json; d = json.load(open('gpurun_out/r04f/bench_default.json')); print({k: d[k] for k in ('value','ms_per_step','steps','warmup')}, d['graph'], {k: (v.get('value', v.get('s_per_patient')), v.get('wall_s')) for k, v in d['secondary'].items()}, d['exec_equivalent']['value'], d['roofline']['frac'])"
