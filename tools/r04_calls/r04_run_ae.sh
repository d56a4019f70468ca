#!/bin/bash
# call ae: two-stage top-k of the RPN losses -- unit test, glue/step parity, op timing, same-box A/B.  NEGATIVE (144/167 us vs torch's 124 us):
# the patch (utils/model_utils.topk_rows, bench.py --topk-two-stage) was reverted; the script is kept as the record of what ran
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04g
export MDT_MIOPEN_SKIP_NAIVE=1
python -m pytest tests/test_golden_gpu.py tests/test_step_parity_gpu.py tests/test_graph_step_gpu.py -x -q 2>&1 | grep -v "MIOpen(HIP)" | tail -4 | cut -c1-220
python - <<'PY' 2>&1 | tee gpurun_out/r04g/topk_timing.txt
import torch, time
from medicaldetectiontoolkit_amd.utils import model_utils as mu
dev = torch.device("cuda:0")
x = torch.rand((8, 449280), device=dev)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for k in (3, 30):
    print("k", k, "torch.topk %.1f us" % t(lambda: torch.topk(x, k, dim=1)), "two-stage %.1f us" % t(lambda: mu.topk_rows(x, k)))
    for chunk in (1024, 2048, 8192):
        print("   chunk", chunk, "%.1f us" % t(lambda: mu.topk_rows(x, k, chunk=chunk)))
PY
LEAN="--steps 15 --warmup 4 --no-secondary --no-roofline --no-eager-leg --no-graph-leg --no-exec-leg --no-graph-preflight --no-cpu-baseline --no-h2d-leg --no-rccl-selftest"
one() { timeout 300 python bench.py $LEAN "$@" 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('A/B $*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  one --topk-two-stage 0 | tee -a gpurun_out/r04g/topk_ab.txt
  one --topk-two-stage 1 | tee -a gpurun_out/r04g/topk_ab.txt
done
