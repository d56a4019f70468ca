#!/bin/bash
# call am: where the Retina U-Net step (BASELINE config 2: 388 ms) spends its time -- steady-state categories + launch-by-launch list, for the next round
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04g
export MDT_MIOPEN_SKIP_NAIVE=1
BENCH_ARGS="--model retina_unet --no-secondary --no-roofline" OUT_NAME=r04g/steady_retina_unet GLUE_OUT=r04g/launch_by_launch_retina_unet.txt bash tools/prof_step.sh 3 300 | head -12
awk '/last step in time order: convolution/{f=1} f && $2 >= 1500' gpurun_out/r04g/launch_by_launch_retina_unet.txt | cut -c1-130 | head -70
