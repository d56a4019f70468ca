#!/bin/bash
# round-2 evidence run on the GPU box: bench line, rocprofv3 kernel stats and PMC passes of the RoIAlign-3D backward
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
python bench.py --steps 10 --warmup 3 > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench.err
tail -c 3000 gpurun_out/r02_bench_line.json
for R in trainlike random; do
  MDT_ROIS=$R bash tools/gpu_prof.sh bwd_fast 60 > gpurun_out/r02_prof_$R.txt 2>&1
  F=$(find gpurun_out/prof_bwd_fast -name "*kernel_stats.csv" | head -1); cp "$F" gpurun_out/r02_roialign3d_bwd_P2_N48_${R}_kernel_stats.csv
  rm -rf gpurun_out/prof_bwd_fast
  for C in WRITE_SIZE FETCH_SIZE; do
    MDT_ROIS=$R bash tools/gpu_pmc.sh $C 5 > gpurun_out/r02_pmc_${R}_$C.txt 2>&1
    F=$(find gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1); cp "$F" gpurun_out/r02_pmc_${R}_${C}_counter_collection.csv
    rm -rf gpurun_out/pmc_$C
  done
done
cat gpurun_out/r02_prof_*.txt gpurun_out/r02_pmc_*.txt
