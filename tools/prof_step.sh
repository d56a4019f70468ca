#!/bin/bash
# rocprofv3 kernel trace of a short bench.py run + steady-state breakdown -> gpurun_out/${OUT_NAME:-r03_step_steady_state}.csv
# env: BENCH_ARGS (e.g. "--model retina_unet"), OUT_NAME, GLUE_OUT (launch-by-launch listing), TOP_OUT (tools/top_launches.py: the longest launches of one step)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
STEPS=${1:-5}
BENCH_ARGS=${BENCH_ARGS:-}
mkdir -p $ROOT/gpurun_out $ROOT/gpurun_out/$(dirname ${OUT_NAME:-x}) $ROOT/gpurun_out/$(dirname ${GLUE_OUT:-x})
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_step
timeout ${2:-420} rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_step -o step -- python $ROOT/bench.py --steps $STEPS --warmup 3 --settle 0 --no-cpu-baseline --no-h2d-leg --no-rccl-selftest --no-dense-rpn-leg $BENCH_ARGS > $ROOT/gpurun_out/prof_step.log 2>&1 < /dev/null
LINE=$(grep '^{"metric"' $ROOT/gpurun_out/prof_step.log | tail -1)
echo "$LINE" | cut -c1-300
MS=$(echo "$LINE" | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
T=$(find $ROOT/gpurun_out/prof_step -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/steady_state.py "$T" $STEPS $MS $ROOT/gpurun_out/${OUT_NAME:-r03_step_steady_state}.csv
if [ -n "$GLUE_OUT" ]; then python $ROOT/tools/glue_breakdown.py "$T" 3 > $ROOT/gpurun_out/$GLUE_OUT; fi
if [ -n "$TOP_OUT" ]; then python $ROOT/tools/top_launches.py "$T" 70 > $ROOT/gpurun_out/$TOP_OUT; fi
rm -rf $ROOT/gpurun_out/prof_step
