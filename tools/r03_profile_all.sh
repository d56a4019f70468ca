#!/bin/bash
# round-3 evidence run on the GPU box: full gpu suite, bench line (incl. secondary configs), steady-state step profile,
# rocprofv3 kernel stats and PMC passes of the RoIAlign-3D backward
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r03_gpu_suite.log 2>&1
tail -4 gpurun_out/r03_gpu_suite.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench.err
tail -c 7000 gpurun_out/r03_bench_line.json; tail -2 gpurun_out/r03_bench.err
for flag in 0 1; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-rccl-selftest --no-secondary --no-roofline --conv3-small $flag 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); print('A/B conv3_small=$flag', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r03_ab_conv3_small.txt
done
OUT_NAME=r03_step_steady_state bash tools/prof_step.sh 5 400 | head -12
for R in trainlike random; do
  T=$([ $R = random ] && echo survey_random || echo trainlike)
  MDT_ROIS=$R bash tools/gpu_prof.sh bwd_fast 60 > gpurun_out/r03_prof_$R.txt 2>&1
  F=$(find gpurun_out/prof_bwd_fast -name "*kernel_stats.csv" | head -1); cp "$F" gpurun_out/r03_roialign3d_bwd_P2_N48_${T}_kernel_stats.csv
  rm -rf gpurun_out/prof_bwd_fast
  for C in WRITE_SIZE FETCH_SIZE; do
    MDT_ROIS=$R bash tools/gpu_pmc.sh $C 5 > gpurun_out/r03_pmc_${T}_$C.txt 2>&1
    F=$(find gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1); cp "$F" gpurun_out/r03_pmc_${T}_${C}_counter_collection.csv
    rm -rf gpurun_out/pmc_$C
  done
done
MDT_ROIS=trainlike bash tools/gpu_prof.sh pyramid_bwd 60 > gpurun_out/r03_prof_pyramid.txt 2>&1
F=$(find gpurun_out/prof_pyramid_bwd -name "*kernel_stats.csv" | head -1); cp "$F" gpurun_out/r03_roialign3d_bwd_all_levels_N48_kernel_stats.csv; rm -rf gpurun_out/prof_pyramid_bwd
cat gpurun_out/r03_prof_*.txt gpurun_out/r03_pmc_*.txt
python tools/make_traffic_json.py gpurun_out gpurun_out/r03_traffic.json | tail -25
