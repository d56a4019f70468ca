#!/bin/bash
# round-4 closing run: full gpu suite (find results collected for the in-tree MIOpen db), the bench line incl. secondary configs
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04f
export MDT_MIOPEN_SKIP_NAIVE=1
rm -rf gpurun_out/r04f/miopen_cache; cp -r medicaldetectiontoolkit_amd/miopen_cache gpurun_out/r04f/miopen_cache
export MDT_MIOPEN_CACHE=$PWD/gpurun_out/r04f/miopen_cache
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r04f/r04_gpu_suite_final.log 2>&1
echo "suite rc=$?"; grep -v "MIOpen(HIP)" gpurun_out/r04f/r04_gpu_suite_final.log | tail -4 | cut -c1-300
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r04f/r04_bench_line_1gpu_final.json 2> gpurun_out/r04f/r04_bench_final.err
echo "bench rc=$?"; wc -c gpurun_out/r04f/r04_bench_line_1gpu_final.json; grep -v "Warning\|amdgpu.ids\|^  \|MIOpen(HIP)" gpurun_out/r04f/r04_bench_final.err | tail -4 | cut -c1-300
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04f/r04_bench_line_1gpu_final.json"))
print({k: d[k] for k in ("value", "ms_per_step", "host_issue_ms_per_step")})
for k in ("graph", "eager_step", "exec_equivalent", "h2d_inclusive", "cpu_baseline", "secondary"):
    print(k, json.dumps(d.get(k))[:1200])
r = d["roofline"]; print("roofline head", r["frac"], r["avg_us"])
for k, v in r["variants"].items(): print("  %-66s %.3f %6.1f us rois %s" % (k, v["frac"], v["avg_us"], v.get("rois")))
print(json.dumps(d["distributed"])[:900])
PY
rm -rf gpurun_out/r04f/miopen_cache/kernels
