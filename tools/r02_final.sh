#!/bin/bash
# round-2 closing evidence run on the GPU box (one gpurun call): GPU test suite, bench line, rocprofv3 kernel stats of the
# headline kernel, steady-state step breakdown, micro-benchmarks.  MIOpen find results are exported to gpurun_out/miopen_final.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/miopen_final && cp -r medicaldetectiontoolkit_amd/miopen_cache gpurun_out/miopen_final
export MDT_MIOPEN_CACHE=$ROOT/gpurun_out/miopen_final MDT_MIOPEN_SKIP_NAIVE=1
timeout 420 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r02_gputest_tail.txt
cat gpurun_out/r02_gputest_tail.txt
timeout -s INT -k 20 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench.err
cut -c1-400 gpurun_out/r02_bench_line.json
MDT_ROIS=trainlike bash tools/gpu_prof.sh bwd_fast 60 > gpurun_out/r02_prof_trainlike.txt 2>&1
F=$(find gpurun_out/prof_bwd_fast -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" gpurun_out/r02_final_roialign3d_bwd_P2_N48_trainlike_kernel_stats.csv
rm -rf gpurun_out/prof_bwd_fast
cat gpurun_out/r02_prof_trainlike.txt | tail -5
timeout 300 bash tools/prof_step.sh 5 280 2>&1 | head -30
timeout 200 python tools/microbench.py --iters 30 > gpurun_out/r02_microbench_final.jsonl 2> /dev/null
grep -E "pyramid|trainlike_P|fast_P2_N48" gpurun_out/r02_microbench_final.jsonl | cut -c1-200
rm -rf gpurun_out/miopen_final/kernels
