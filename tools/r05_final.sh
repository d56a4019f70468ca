#!/bin/bash
# round-5 closing evidence run on the GPU box -> gpurun_out/r05/ (copied to profiles/r05/ by the builder)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r05
export MDT_MIOPEN_SKIP_NAIVE=1
python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -12 > gpurun_out/r05/r05_gpu_test_suite_tail.txt
cat gpurun_out/r05/r05_gpu_test_suite_tail.txt | tail -4
python bench.py > gpurun_out/r05/r05_bench_line_1gpu_driver_style.json 2> gpurun_out/r05/r05_bench_driver_style.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/r05_bench_line_1gpu_driver_style.json"))
e = d["exec_equivalent"]
print("value", d["value"], "ms", d["ms_per_step"], "host", d["host_issue_ms_per_step"], "| exec_eq", e.get("value"), (e.get("synchronous_readout_form") or {}).get("value"),
      "| heads_full", (d.get("heads_full_step") or {}).get("value"), "| roofline", d["roofline"]["frac"], d["roofline"]["traffic"], "| cpu", (d.get("cpu_baseline") or {}).get("value"),
      (d.get("cpu_baseline") or {}).get("kind"), "| graphed", (d.get("graphed_step") or {}).get("value"), "| dense", (d.get("dense_rpn_graph_step") or {}).get("value"))
print({k: v.get("value") for k, v in (d.get("secondary") or {}).items() if isinstance(v, dict)})
PY
LEAN="--no-secondary --no-roofline --no-graph-leg --no-exec-leg"
BENCH_ARGS="$LEAN" OUT_NAME=r05/r05_bench_train_step_steady_state_kernels_eager GLUE_OUT=r05/r05_step_launch_by_launch.txt bash tools/prof_step.sh 5 400 | head -14
