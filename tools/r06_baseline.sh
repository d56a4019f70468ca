#!/bin/bash
# round-6 opening run on the GPU box: suite, driver-style bench line (exec-form headline), launch-by-launch profile of the step "before"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
T=${1:-a}
mkdir -p gpurun_out/r06
export MDT_MIOPEN_SKIP_NAIVE=1
python -m pytest tests -m gpu -q --tb=short -x ${PYTEST_EXTRA:-} 2>&1 | tail -25 > gpurun_out/r06/r06_gpu_test_suite_tail_$T.txt
tail -4 gpurun_out/r06/r06_gpu_test_suite_tail_$T.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/r06_bench_line_1gpu_$T.json 2> gpurun_out/r06/r06_bench_$T.err
python - $T <<'PY'
import json, sys
d = json.load(open("gpurun_out/r06/r06_bench_line_1gpu_%s.json" % sys.argv[1]))
e = d.get("exec_equivalent") or {}
print("value", d["value"], "ms", d["ms_per_step"], "host", d["host_issue_ms_per_step"], "| no_readout", (d.get("no_readout_step") or {}).get("value"),
      (d.get("no_readout_step") or {}).get("host_issue_ms_per_step"), "| exec_eq(host batches)", e.get("value"), e.get("steps"), (e.get("synchronous_readout_form") or {}).get("value"),
      "| heads_full", (d.get("heads_full_step") or {}).get("value"), "| roofline", (d.get("roofline") or {}).get("frac"), "| cpu", (d.get("cpu_baseline") or {}).get("value"),
      "| graphed", (d.get("graphed_step") or {}).get("value"), "| dense", (d.get("dense_rpn_graph_step") or {}).get("value"))
print({k: v.get("value", v.get("patches_per_s", v.get("failed"))) for k, v in (d.get("secondary") or {}).items() if isinstance(v, dict)})
print(d["readout_consumed"])
PY
LEAN="--no-secondary --no-roofline --no-graph-leg --no-exec-leg"
BENCH_ARGS="$LEAN" OUT_NAME=r06/r06_bench_train_step_steady_state_kernels_eager_$T GLUE_OUT=r06/r06_step_launch_by_launch_$T.txt bash tools/prof_step.sh 5 400 | head -14
