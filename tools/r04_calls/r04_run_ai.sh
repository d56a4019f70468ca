#!/bin/bash
# call ai: host-side cProfile of the eager step
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04g
export MDT_MIOPEN_SKIP_NAIVE=1
timeout 400 python tools/host_profile.py 10 2>&1 | grep -v "MIOpen(HIP)" > gpurun_out/r04g/host_profile_eager.txt
timeout 400 python tools/host_profile.py 10 64,64,32 2>&1 | grep -v "MIOpen(HIP)" > gpurun_out/r04g/host_profile_eager_small_patch.txt
head -125 gpurun_out/r04g/host_profile_eager_small_patch.txt
