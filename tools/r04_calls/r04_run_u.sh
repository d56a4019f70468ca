#!/bin/bash
mkdir -p gpurun_out/r04p
for CAP in 81920 53248 40960 32768; do for ROTN in 0 4; do for R in trainlike random; do
  MDT_BWD_TUNE=1 MDT_BWD3_LDS_CAP=$CAP MDT_ROTATE=$ROTN MDT_ROIS=$R bash tools/gpu_prof.sh bwd_fast 60 2>&1 | grep "crop_bwd\|territory" | sed "s/^/lds_cap=$CAP rotate=$ROTN $R /" | tee -a gpurun_out/r04p/r04_bwd_lds_cap_occupancy.txt
  rm -rf gpurun_out/prof_bwd_fast
done; done; done
for Z in 224 448 736; do for ROTN in 0 4; do
  MDT_BWD_TUNE=1 MDT_BWD3_ZERO_WGS=$Z MDT_ROTATE=$ROTN MDT_ROIS=trainlike bash tools/gpu_prof.sh bwd_fast 60 2>&1 | grep "crop_bwd" | sed "s/^/zero_wgs=$Z rotate=$ROTN trainlike /" | tee -a gpurun_out/r04p/r04_bwd_lds_cap_occupancy.txt
  rm -rf gpurun_out/prof_bwd_fast
done; done
