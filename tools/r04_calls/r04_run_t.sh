#!/bin/bash
mkdir -p gpurun_out/r04p
for CR in 0 8 16 32 64; do for ROTN in 0 4; do
  MDT_BWD_TUNE=1 MDT_BWD3_ZERO_CHUNK_ROWS=$CR MDT_ROTATE=$ROTN MDT_ROIS=random bash tools/gpu_prof.sh bwd_fast 60 2>&1 | grep crop_bwd | sed "s/^/chunk_rows=$CR rotate=$ROTN survey /" | tee -a gpurun_out/r04p/r04_bwd_zero_window.txt
  rm -rf gpurun_out/prof_bwd_fast
done; done
for CR in 0 16 32; do for ROTN in 0 4; do
  MDT_BWD_TUNE=1 MDT_BWD3_ZERO_CHUNK_ROWS=$CR MDT_ROTATE=$ROTN MDT_ROIS=trainlike bash tools/gpu_prof.sh bwd_fast 60 2>&1 | grep crop_bwd | sed "s/^/chunk_rows=$CR rotate=$ROTN trainlike /" | tee -a gpurun_out/r04p/r04_bwd_zero_window.txt
  rm -rf gpurun_out/prof_bwd_fast
done; done
