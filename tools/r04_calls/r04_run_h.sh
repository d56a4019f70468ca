#!/bin/bash
# round 4, call h: new tests (continued) + graphed bench line with host-phase timing of the exec leg
mkdir -p gpurun_out/r04
export MDT_MIOPEN_SKIP_NAIVE=1
rm -rf gpurun_out/r04/miopen_cache; cp -r medicaldetectiontoolkit_amd/miopen_cache gpurun_out/r04/miopen_cache
export MDT_MIOPEN_CACHE=$PWD/gpurun_out/r04/miopen_cache
timeout 1800 python -m pytest tests/test_graph_step_gpu.py tests/test_step_parity_gpu.py tests/test_golden_gpu.py tests/test_glue_parity_gpu.py "tests/test_hip_gpu.py::test_roialign_forward_uint8_input_bit_equal_to_fp32_and_oracle" "tests/test_models_gpu.py" tests/test_flat_adam_gpu.py tests/test_distributed_gpu.py -q 2>&1 | tail -40 | cut -c1-500 | tee gpurun_out/r04/h_tests.log
unset MDT_MIOPEN_CACHE
timeout 900 python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline > gpurun_out/r04/bench_h.json 2> gpurun_out/r04/bench_h.err
echo "bench rc=$?"; python - <<'PY'
import json
d = json.load(open("gpurun_out/r04/bench_h.json"))
print({k: d[k] for k in ("value", "ms_per_step", "host_issue_ms_per_step")})
for k in ("graph", "eager_step", "exec_equivalent", "h2d_inclusive"):
    print(k, json.dumps(d.get(k))[:900])
PY
grep -v "Warning\|amdgpu.ids\|^  " gpurun_out/r04/bench_h.err | tail -8 | cut -c1-300
rm -rf gpurun_out/r04/miopen_cache/kernels
