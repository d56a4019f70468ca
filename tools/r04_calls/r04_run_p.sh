#!/bin/bash
# round 4, call p: whole GPU suite (packet capture off, wave-staged forward kernel), bench line, forward kernel A/B, zero-role geometry A/B
mkdir -p gpurun_out/r04
export MDT_MIOPEN_SKIP_NAIVE=1
rm -rf gpurun_out/r04/miopen_cache; cp -r medicaldetectiontoolkit_amd/miopen_cache gpurun_out/r04/miopen_cache
export MDT_MIOPEN_CACHE=$PWD/gpurun_out/r04/miopen_cache
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r04/p_tests.log 2>&1
echo "suite rc=$?"; tail -15 gpurun_out/r04/p_tests.log | cut -c1-300
unset MDT_MIOPEN_CACHE
timeout 900 python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline > gpurun_out/r04/bench_p.json 2> gpurun_out/r04/bench_p.err
echo "bench rc=$?"; python - <<'PY'
import json
d = json.load(open("gpurun_out/r04/bench_p.json"))
print({k: d[k] for k in ("value", "ms_per_step", "host_issue_ms_per_step")})
for k in ("graph", "eager_step", "exec_equivalent", "h2d_inclusive"):
    print(k, json.dumps(d.get(k))[:700])
r = d["roofline"]; print("roofline head", r["frac"], r["avg_us"], r["kernel"][:120])
for k, v in r["variants"].items(): print("  %-66s %.3f %6.1f us rois %s" % (k, v["frac"], v["avg_us"], v.get("rois")))
PY
grep -v "Warning\|amdgpu.ids\|^  " gpurun_out/r04/bench_p.err | tail -5 | cut -c1-300
python tools/fwd_bench.py 2>/dev/null | tee gpurun_out/r04/fwd_bench_wave.jsonl | cut -c1-260
MDT_FWD_KERNEL=direct python tools/fwd_bench.py 2>/dev/null | tee gpurun_out/r04/fwd_bench_direct.jsonl | cut -c1-260
for cr in 0 8 32 128; do MDT_BWD_TUNE=1 MDT_BWD3_ZERO_CHUNK_ROWS=$cr python tools/bwd_cold_probe.py 2>/dev/null | tee -a gpurun_out/r04/bwd_cold_probe.jsonl | cut -c1-900; done
rm -rf gpurun_out/r04/miopen_cache/kernels
