#!/bin/bash
export MIOPEN_LOG_LEVEL=0 MDT_MIOPEN_SKIP_NAIVE=1
mkdir -p gpurun_out/r04p
timeout 900 python -m pytest tests/test_epilogue_gpu.py -q -x -k "dgrad_add or residual_tap" 2>&1 | grep -v "MIOpen(HIP)" | tail -6 | cut -c1-300
python tools/res_tap_probe.py 2>/dev/null | tee gpurun_out/r04p/r04_res_tap_probe.jsonl
MDT_DGRAD_ADD=valu python tools/res_tap_probe.py 2>/dev/null | sed 's/^/valu form: /' | tee -a gpurun_out/r04p/r04_res_tap_probe.jsonl
timeout 900 python -m pytest tests/test_step_parity_gpu.py tests/test_backbone_parity_gpu.py -q -x 2>&1 | grep -v "MIOpen(HIP)" | tail -3 | cut -c1-300
LEAN="--steps 15 --warmup 4 --no-secondary --no-roofline --no-eager-leg --no-graph-leg --no-exec-leg --no-graph-preflight --no-cpu-baseline --no-h2d-leg --no-rccl-selftest"
for rep in 1 2; do for f in 0 1; do
  timeout 300 python bench.py $LEAN --res-tap $f 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('A/B res_tap(mfma)=$f', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r04p/r04_ab_res_tap.txt
done; done
