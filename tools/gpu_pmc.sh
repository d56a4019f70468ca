#!/bin/bash
# PMC pass (counters only, no tracing -- separate pass per counter as the guide prescribes)
# usage: tools/gpu_pmc.sh <COUNTER> [iters]
set -u
CTR=$1; IT=${2:-5}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc $CTR --output-format csv -d $ROOT/gpurun_out/pmc_$CTR -o pmc -- python $ROOT/tools/profile_case.py pmc_bwd $IT > $ROOT/gpurun_out/pmc_$CTR.log 2>&1 < /dev/null
F=$(find $ROOT/gpurun_out/pmc_$CTR -name "*counter_collection.csv" | head -1)
if [ -n "$F" ]; then python - "$F" "$CTR" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ctr = sys.argv[2]
agg = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") == ctr:
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "crop_bwd" in k or "FillFunctor" in k or "fill" in k.lower():
        print("%s  n=%d  mean=%.1f  %s" % (ctr, len(v), sum(v) / len(v), k[:90]))
PY
else echo "no counter csv"; tail -3 $ROOT/gpurun_out/pmc_$CTR.log; fi
