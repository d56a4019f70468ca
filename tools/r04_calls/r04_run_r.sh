#!/bin/bash
mkdir -p gpurun_out/r04
export MDT_MIOPEN_SKIP_NAIVE=1 MIOPEN_LOG_LEVEL=0
timeout 900 python -m pytest tests/test_hip_gpu.py tests/test_pyramid_roialign_gpu.py tests/test_glue_parity_gpu.py "tests/test_models_gpu.py" -q -x > gpurun_out/r04/r_tests.log 2>&1; echo "tests rc=$?"; grep -v "MIOpen" gpurun_out/r04/r_tests.log | tail -6 | cut -c1-300
python tools/fwd_bench.py 2>/dev/null | tee gpurun_out/r04/fwd_bench_wave3.jsonl | cut -c1-200
MDT_FWD_CPW=8 python tools/fwd_bench.py 2>/dev/null | grep "N600\|N4096" | cut -c1-200
MDT_FWD_KERNEL=direct python tools/fwd_bench.py 2>/dev/null | tee gpurun_out/r04/fwd_bench_direct3.jsonl | cut -c1-200
