#!/bin/bash
mkdir -p gpurun_out/r04f gpurun_out/r04p
export MIOPEN_LOG_LEVEL=0
for DBG in 0 16; do for ROTN in 0 4; do for R in random trainlike; do
  MDT_BWD3_DBG=$DBG MDT_ROTATE=$ROTN MDT_ROIS=$R bash tools/gpu_prof.sh bwd_fast 60 2>&1 | grep crop_bwd | sed "s/^/zero_role_nontemporal=$((DBG/16)) rotate=$ROTN $R /" | tee -a gpurun_out/r04p/r04_bwd_zero_nontemporal.txt
  rm -rf gpurun_out/prof_bwd_fast
done; done; done
SECONDS=0
timeout 1500 python bench.py > gpurun_out/r04f/bench_default.json 2> gpurun_out/r04f/bench_default.err
echo "bench default rc=$? wall=${SECONDS}s"; python -c "
import json; d = json.load(open('gpurun_out/r04f/bench_default.json')); print({k: d[k] for k in ('value','ms_per_step','steps','warmup')}, d['graph'], {k: (v.get('value', v.get('s_per_patient')), v.get('wall_s')) for k, v in d['secondary'].items()}, d['exec_equivalent']['value'], d['roofline']['frac'])"
