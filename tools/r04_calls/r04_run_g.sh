#!/bin/bash
# round 4, call g: the new tests (batched matching, u8 RoIAlign, large / bench step parity with dispatch, graphed step, bf16 pin) + first graphed bench line
mkdir -p gpurun_out/r04
export MDT_MIOPEN_SKIP_NAIVE=1
rm -rf gpurun_out/r04/miopen_cache; cp -r medicaldetectiontoolkit_amd/miopen_cache gpurun_out/r04/miopen_cache
export MDT_MIOPEN_CACHE=$PWD/gpurun_out/r04/miopen_cache
timeout 1500 python -m pytest tests/test_graph_step_gpu.py tests/test_step_parity_gpu.py tests/test_golden_gpu.py tests/test_glue_parity_gpu.py "tests/test_hip_gpu.py::test_roialign_forward_uint8_input_bit_equal_to_fp32_and_oracle" "tests/test_models_gpu.py" tests/test_flat_adam_gpu.py -x -q 2>&1 | tail -25 | cut -c1-400 | tee gpurun_out/r04/g_tests.log
unset MDT_MIOPEN_CACHE
timeout 900 python bench.py --steps 20 --warmup 3 --no-secondary > gpurun_out/r04/bench_g.json 2> gpurun_out/r04/bench_g.err
echo "bench rc=$?"; cat gpurun_out/r04/bench_g.json | cut -c1-3000; grep -v "Warning\|amdgpu.ids\|^  " gpurun_out/r04/bench_g.err | tail -15 | cut -c1-300
rm -rf gpurun_out/r04/miopen_cache/kernels
