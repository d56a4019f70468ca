#!/bin/bash
# round 3, call N: stem forward kernel + LDS-image stem weight gradient (tests, timing, A/B in the step), host-batch leg check,
# and the MIOpen pose probe for the layers that lead the step now
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_epilogue_gpu.py -x -q -m gpu -k "stem" > gpurun_out/r03_tests_n.log 2>&1
rc=$?; tail -4 gpurun_out/r03_tests_n.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "TESTS FAILED"; tail -60 gpurun_out/r03_tests_n.log | cut -c1-400; fi
timeout 200 python tools/conv_pose_probe.py stem h2d > gpurun_out/r03_conv_pose_probe.jsonl 2> gpurun_out/r03_conv_pose_probe.err
cat gpurun_out/r03_conv_pose_probe.jsonl
for v in v2 v1; do
MDT_STEM_WGRAD=$v timeout 120 python - <<'PY'
import torch, json, os
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
dev = torch.device("cuda:0")
x = torch.randn(8, 1, 128, 128, 128, device=dev)
w = torch.randn(18, 1, 7, 7, 7, device=dev)
gy = torch.randn(8, 18, 64, 64, 128, device=dev).contiguous(memory_format=torch.channels_last_3d)
xp = torch.nn.functional.pad(x.reshape(8, 128, 128, 128), (3,) * 6).contiguous()
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
print(json.dumps({"stem_wgrad_7x7x7_8x128^3": os.environ.get("MDT_STEM_WGRAD"), "us_incl_pad": round(t(lambda: fe.stem_weight_grad(gy, x, w, (2, 2, 1))), 1),
                  "us_padded_copy_given": round(t(lambda: fe.stem_weight_grad(gy, x, w, (2, 2, 1), xp=xp)), 1)}))
PY
done 2>&1 | tee gpurun_out/r03_stem_wgrad_forms.jsonl
if [ $rc -ne 0 ]; then exit 1; fi
for flag in 0 1; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-rccl-selftest --no-secondary --no-roofline --stem-fwd $flag 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('A/B stem_fwd=$flag', d['value'], d['ms_per_step'])"
done 2>&1 | tee gpurun_out/r03_ab_stem_fwd.txt
timeout 200 python bench.py --steps 5 --warmup 2 --host-batches --no-cpu-baseline --no-rccl-selftest --no-secondary --no-roofline 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('host-batches', d['value'], d['ms_per_step'])" | tee gpurun_out/r03_host_batches.txt
timeout 400 python tools/conv_pose_probe.py rpn p2 c3 >> gpurun_out/r03_conv_pose_probe.jsonl 2>> gpurun_out/r03_conv_pose_probe.err
tail -20 gpurun_out/r03_conv_pose_probe.jsonl
