#!/bin/bash
mkdir -p gpurun_out/r04
export MDT_MIOPEN_SKIP_NAIVE=1
timeout 900 python -m pytest tests/test_hip_gpu.py tests/test_pyramid_roialign_gpu.py tests/test_golden_gpu.py -q -x 2>&1 | tail -4 | cut -c1-300
timeout 600 python -m pytest "tests/test_models_gpu.py::test_bf16_patch_tiled_inference_tracks_fp32" -q -x -s 2>&1 | grep "bf16 vs\|passed\|failed\|Error" | cut -c1-600
python tools/fwd_bench.py 2>/dev/null | tee gpurun_out/r04/fwd_bench_wave2.jsonl | cut -c1-240
MDT_FWD_CPW=1 python tools/fwd_bench.py 2>/dev/null | grep "N600\|N4096" | cut -c1-240
MDT_FWD_CPW=8 python tools/fwd_bench.py 2>/dev/null | grep "N600\|N4096" | cut -c1-240
