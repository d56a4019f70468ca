#!/bin/bash
# round-4 same-box A/B pairs (lean bench runs, 15 timed steps each, two alternations): merged RPN heads on / off; graphed step on / off
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04p
export MDT_MIOPEN_SKIP_NAIVE=1
LEAN="--steps 15 --warmup 4 --no-secondary --no-roofline --no-eager-leg --no-graph-leg --no-exec-leg --no-graph-preflight --no-cpu-baseline --no-h2d-leg --no-rccl-selftest"
one() { timeout 300 python bench.py $LEAN "$@" 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('A/B $*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  one --merge-rpn-heads 0 | tee -a gpurun_out/r04p/r04_ab_same_box.txt
  one --merge-rpn-heads 1 | tee -a gpurun_out/r04p/r04_ab_same_box.txt
done
for rep in 1 2; do
  one --graph 0 | tee -a gpurun_out/r04p/r04_ab_same_box.txt
  one --graph 1 | tee -a gpurun_out/r04p/r04_ab_same_box.txt
done
