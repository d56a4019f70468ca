#!/bin/bash
# round 4, call i: graphed-step tests with full log, then the rest of the new tests (no -x), bench with host-work accounting
mkdir -p gpurun_out/r04
export MDT_MIOPEN_SKIP_NAIVE=1
rm -rf gpurun_out/r04/miopen_cache; cp -r medicaldetectiontoolkit_amd/miopen_cache gpurun_out/r04/miopen_cache
export MDT_MIOPEN_CACHE=$PWD/gpurun_out/r04/miopen_cache
timeout 900 python -X faulthandler -m pytest tests/test_graph_step_gpu.py -q -x > gpurun_out/r04/i_graph_tests.log 2>&1
echo "graph tests rc=$?"; grep -n "Memory access\|Error\|error\|passed\|failed\|HSA\|fault" gpurun_out/r04/i_graph_tests.log | head -20 | cut -c1-300
timeout 1800 python -m pytest tests/test_step_parity_gpu.py tests/test_golden_gpu.py tests/test_glue_parity_gpu.py "tests/test_hip_gpu.py::test_roialign_forward_uint8_input_bit_equal_to_fp32_and_oracle" "tests/test_models_gpu.py" tests/test_flat_adam_gpu.py tests/test_distributed_gpu.py -q 2>&1 | tail -40 | cut -c1-500 | tee gpurun_out/r04/i_tests.log
unset MDT_MIOPEN_CACHE
timeout 900 python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline > gpurun_out/r04/bench_i.json 2> gpurun_out/r04/bench_i.err
echo "bench rc=$?"; python - <<'PY'
import json
d = json.load(open("gpurun_out/r04/bench_i.json"))
print({k: d[k] for k in ("value", "ms_per_step", "host_issue_ms_per_step")})
for k in ("graph", "eager_step", "exec_equivalent", "h2d_inclusive"):
    print(k, json.dumps(d.get(k))[:900])
PY
grep -v "Warning\|amdgpu.ids\|^  " gpurun_out/r04/bench_i.err | tail -8 | cut -c1-300
rm -rf gpurun_out/r04/miopen_cache/kernels
