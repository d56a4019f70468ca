#!/bin/bash
# Refresh the committed MIOpen find-db / kernel cache (medicaldetectiontoolkit_amd/miopen_cache) on the GPU box: run the given command with the
# cache written IN PLACE, then pack the cache into gpurun_out/ (the only directory that travels back); unpack it here with
#   tar xzf gpurun_out/miopen_cache.tgz -C medicaldetectiontoolkit_amd
# usage (GPU box): bash tools/refresh_miopen_cache.sh "python -m pytest tests/test_step_parity_gpu.py -q -m gpu"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
export MDT_MIOPEN_CACHE_INPLACE=1 MDT_MIOPEN_SKIP_NAIVE=1
bash -c "$1" 2>&1 | tail -15
tar czf gpurun_out/miopen_cache.tgz -C medicaldetectiontoolkit_amd miopen_cache
ls -la gpurun_out/miopen_cache.tgz
