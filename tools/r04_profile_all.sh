#!/bin/bash
# round-4 evidence run on the GPU box (profiles/r04_*): steady-state step profiles (graphed and eager), rocprofv3 kernel stats of the RoIAlign-3D
# backward cache-warm AND with rotating outputs (cache-cold), PMC traffic in both states, kernel stats of the RoIAlign forward kernels
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04p
export MDT_MIOPEN_SKIP_NAIVE=1
LEAN="--no-secondary --no-roofline --no-eager-leg --no-graph-leg --no-exec-leg --no-graph-preflight"
BENCH_ARGS="$LEAN --graph 1" OUT_NAME=r04p/r04_bench_train_step_steady_state_kernels_graphed bash tools/prof_step.sh 5 400 | head -14
BENCH_ARGS="$LEAN --graph 0" OUT_NAME=r04p/r04_bench_train_step_steady_state_kernels_eager bash tools/prof_step.sh 5 400 | head -14
for R in random trainlike; do
  T=$([ $R = random ] && echo survey_random || echo trainlike)
  for ROTN in 0 4; do
    S=$([ $ROTN = 0 ] && echo cache_warm || echo rotating_outputs_cache_cold)
    MDT_ROTATE=$ROTN MDT_ROIS=$R bash tools/gpu_prof.sh bwd_fast 60 > gpurun_out/r04p/prof_${T}_$S.txt 2>&1
    F=$(find gpurun_out/prof_bwd_fast -name "*kernel_stats.csv" | head -1); cp "$F" gpurun_out/r04p/r04_roialign3d_bwd_P2_N48_${T}_${S}_kernel_stats.csv
    rm -rf gpurun_out/prof_bwd_fast
    cat gpurun_out/r04p/prof_${T}_$S.txt | head -6
  done
done
for ROTN in 0 4; do
  S=$([ $ROTN = 0 ] && echo cache_warm || echo rotating_outputs_cache_cold)
  for C in WRITE_SIZE FETCH_SIZE; do
    MDT_ROTATE=$ROTN MDT_ROIS=random bash tools/gpu_pmc.sh $C 8 > gpurun_out/r04p/r04_pmc_survey_random_${S}_$C.txt 2>&1
    F=$(find gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1); cp "$F" gpurun_out/r04p/r04_pmc_survey_random_${S}_${C}_counter_collection.csv
    rm -rf gpurun_out/pmc_$C
    cat gpurun_out/r04p/r04_pmc_survey_random_${S}_$C.txt
  done
done
for CASE in "240 14,14,5" "600 7,7,3"; do
  set -- $CASE
  for K in wave direct; do
    MDT_FWD_KERNEL=$K MDT_N=$1 MDT_CROP=$2 bash tools/gpu_prof.sh fwd 60 > gpurun_out/r04p/prof_fwd_$1_$K.txt 2>&1
    F=$(find gpurun_out/prof_fwd -name "*kernel_stats.csv" | head -1); cp "$F" gpurun_out/r04p/r04_roialign3d_fwd_P2_N$1_$(echo $2 | tr , x)_${K}_kernel_stats.csv
    rm -rf gpurun_out/prof_fwd
    head -4 gpurun_out/r04p/prof_fwd_$1_$K.txt
  done
done
ls gpurun_out/r04p
