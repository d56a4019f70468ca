#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_epilogue_gpu.py -x -q -m gpu -k "stem_wgrad" > gpurun_out/r03_tests_m.log 2>&1
rc=$?; tail -4 gpurun_out/r03_tests_m.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "TESTS FAILED"; tail -40 gpurun_out/r03_tests_m.log | cut -c1-300; exit 1; fi
python - <<'PY'
import torch, json
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
x = torch.randn(8, 1, 128, 128, 128, device=dev)
w = torch.randn(18, 1, 7, 7, 7, device=dev)
gy = torch.randn(8, 18, 64, 64, 128, device=dev).contiguous(memory_format=torch.channels_last_3d)
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
new = t(lambda: fe.stem_weight_grad(gy, x, w, (2, 2, 1)))
ref = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2, 1], [3, 3, 3], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False]))
print(json.dumps({"stem_wgrad_7x7x7_8x128^3": {"mdt_us_incl_pad": round(new, 1), "miopen_us": round(ref, 1)}}))
PY
for flag in 0 1; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-rccl-selftest --no-secondary --no-roofline --stem-wgrad $flag 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('stem_wgrad=$flag', d['value'], d['ms_per_step'])"
done
