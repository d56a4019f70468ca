# same-box A/B of round 6's epilogue switches: headline step only, two passes (usage: bash tools/r06_epilogue_ab.sh "<flag set 1>" "<flag set 2>" ...)
Q="--no-cpu-baseline --no-secondary --no-roofline --no-rccl-selftest --no-h2d-leg --no-graph-leg --no-eager-leg --no-dense-rpn-leg --no-exec-leg --no-graph-preflight"
for rep in 1 2; do
for f in "$@"; do
  echo "== [$f]"; python bench.py --steps 20 --warmup 5 $Q $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
