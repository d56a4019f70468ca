#!/bin/bash
# round 5: rocprofv3 kernel stats + PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs) of the RoIAlign-3D forward (channel-quad kernel) at the
# SURVEY 8(d) inference call sizes on P2 -> gpurun_out/r05/
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CASE in "240 14,14,5" "600 7,7,3"; do
  set -- $CASE
  TAG=N$1_$(echo $2 | tr , x)
  rm -rf $ROOT/gpurun_out/prof_fwd
  MDT_N=$1 MDT_CROP=$2 bash $ROOT/tools/gpu_prof.sh fwd 60 > $OUT/prof_fwd_$TAG.txt 2>&1
  F=$(find $ROOT/gpurun_out/prof_fwd -name "*kernel_stats.csv" | head -1); cp "$F" $OUT/r05_roialign3d_fwd_P2_${TAG}_cq_kernel_stats.csv
  grep -i "crop_fwd" $OUT/r05_roialign3d_fwd_P2_${TAG}_cq_kernel_stats.csv | cut -c1-160
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $ROOT/gpurun_out/pmc_fwd
    MDT_N=$1 MDT_CROP=$2 timeout 120 rocprofv3 --pmc $C --output-format csv -d $ROOT/gpurun_out/pmc_fwd -o pmc -- python $ROOT/tools/profile_case.py fwd 8 > $OUT/pmc_fwd.log 2>&1 < /dev/null
    F=$(find $ROOT/gpurun_out/pmc_fwd -name "*counter_collection.csv" | head -1)
    python - "$F" $C $TAG <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float)
names = {}
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") == sys.argv[2] and "crop_fwd" in r["Kernel_Name"]:
        agg[r.get("Dispatch_Id")] += float(r["Counter_Value"])
v = list(agg.values())
print("%s %s per launch: n=%d mean=%.1f KB (FETCH_SIZE: double it for wide coalesced reads on gfx950)" % (sys.argv[3], sys.argv[2], len(v), sum(v) / max(1, len(v))))
PY
  done
done 2>&1 | tee $OUT/r05_roialign3d_fwd_pmc_summary.txt
rm -rf $ROOT/gpurun_out/prof_fwd $ROOT/gpurun_out/pmc_fwd
