mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_epilogue_gpu.py -q -x 2>&1 | tail -5
Q="--no-cpu-baseline --no-secondary --no-roofline --no-rccl-selftest --no-h2d-leg --no-graph-leg --no-eager-leg --no-dense-rpn-leg --no-exec-leg --no-graph-preflight"
for rep in 1 2; do
for f in "" "--bias-bwd-no-copy 0" "--bias-grad-transpose 0" "--lateral-upsample-fused 0" "--bias-bwd-no-copy 0 --bias-grad-transpose 0 --lateral-upsample-fused 0"; do
  echo "== $f"; python bench.py --steps 20 --warmup 5 $Q $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
