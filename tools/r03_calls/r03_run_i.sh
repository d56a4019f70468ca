#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_epilogue_gpu.py tests/test_backbone_parity_gpu.py -x -q -m gpu > gpurun_out/r03_tests_i.log 2>&1
rc=$?; tail -4 gpurun_out/r03_tests_i.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "TESTS FAILED"; tail -30 gpurun_out/r03_tests_i.log | cut -c1-300; exit 1; fi
python - <<'PY'
import torch, time
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
dev = torch.device("cuda:0")
x = torch.randn(8, 18, 64, 64, 128, device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
pool = fe.MaxPool3dStem(kernel_size=3, stride=(2, 2, 1), padding=1)
y = pool(x)
g = torch.randn_like(y)
for _ in range(3):
    x.grad = None; y.backward(g, retain_graph=True)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    x.grad = None; y.backward(g, retain_graph=True)
b.record(); torch.cuda.synchronize()
print("maxpool k3s221 channels-last backward, 8x18x64x64x128: %.1f us per call (round 2: 600 us)" % (a.elapsed_time(b) * 1e3 / 20))
PY
