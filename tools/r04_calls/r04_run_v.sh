#!/bin/bash
mkdir -p gpurun_out/r04f
export MDT_MIOPEN_SKIP_NAIVE=1
timeout 300 python tools/graph_after_eager_probe.py small 2>&1 | grep -v "Warning\|amdgpu.ids\|MIOpen(HIP)\|^  " | tail -3
timeout 300 python tools/graph_after_eager_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids\|MIOpen(HIP)\|^  " | tail -3
export MDT_MIOPEN_CACHE=$PWD/gpurun_out/r04f/miopen_cache
rm -rf gpurun_out/r04f/miopen_cache; cp -r medicaldetectiontoolkit_amd/miopen_cache gpurun_out/r04f/miopen_cache
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r04f/r04_bench_line_1gpu_final.json 2> gpurun_out/r04f/r04_bench_final.err
echo "bench rc=$?"; wc -c gpurun_out/r04f/r04_bench_line_1gpu_final.json; grep -v "Warning\|amdgpu.ids\|^  \|MIOpen(HIP)" gpurun_out/r04f/r04_bench_final.err | tail -4 | cut -c1-300
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04f/r04_bench_line_1gpu_final.json"))
print({k: d[k] for k in ("value", "ms_per_step", "host_issue_ms_per_step")})
for k in ("graph", "eager_step", "graphed_step", "exec_equivalent", "h2d_inclusive", "cpu_baseline", "secondary"):
    print(k, json.dumps(d.get(k))[:1400])
r = d["roofline"]; print("roofline head", r["frac"], r["avg_us"])
for k, v in r["variants"].items(): print("  %-66s %.3f %6.1f us rois %s" % (k, v["frac"], v["avg_us"], v.get("rois")))
print(json.dumps(d["distributed"])[:900])
PY
timeout 600 python bench.py --steps 10 --warmup 3 --graph 1 --no-secondary --no-cpu-baseline > gpurun_out/r04f/r04_bench_line_graph_headline.json 2> gpurun_out/r04f/r04_bench_graph.err
echo "bench --graph 1 rc=$?"; python -c "
import json; d = json.load(open('gpurun_out/r04f/r04_bench_line_graph_headline.json')); print({k: d[k] for k in ('value','ms_per_step')}, json.dumps(d['graph'])[:400], json.dumps(d['eager_step'])[:300], json.dumps(d['exec_equivalent'])[:300])"
rm -rf gpurun_out/r04f/miopen_cache/kernels
