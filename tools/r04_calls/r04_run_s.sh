#!/bin/bash
# which role of the RoIAlign-backward kernel loses the time when the output map is cache-cold? role switches (mdt_debug_bwd3) x rotation
mkdir -p gpurun_out/r04p
for DBG in 0 1 2 8; do for ROTN in 0 4; do
  MDT_BWD3_DBG=$DBG MDT_ROTATE=$ROTN MDT_ROIS=random bash tools/gpu_prof.sh bwd_fast 60 2>&1 | grep crop_bwd | sed "s/^/dbg=$DBG rotate=$ROTN /" | tee -a gpurun_out/r04p/r04_bwd_roles_cold_warm.txt
  rm -rf gpurun_out/prof_bwd_fast
done; done
