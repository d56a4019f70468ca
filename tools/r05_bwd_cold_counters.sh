#!/bin/bash
# round 5: why does the RoIAlign-3D backward's zero stream take 2x as long when its 151 MB output map is not the one it wrote
# last (4 rotating outputs) while a plain fill loses 3 us?  L2 <-> fabric counters of both kernels in both states, one counter per pass.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\(STALL\|WRREQ\|WRITEBACK\|EVICT\|PROBE\|TOO_MANY\)[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $OUT/r05_tcc_counters_available.txt
cat $OUT/r05_tcc_counters_available.txt; echo
for C in ${COUNTERS:-TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WR_UNCACHED_32B_sum}; do
  for ROTN in 0 4; do
    rm -rf $ROOT/gpurun_out/pmc_cold
    MDT_ROTATE=$ROTN MDT_ROIS=random timeout 100 rocprofv3 --pmc $C --output-format csv -d $ROOT/gpurun_out/pmc_cold -o pmc -- python $ROOT/tools/profile_case.py pmc_bwd 8 > $OUT/pmc_cold.log 2>&1 < /dev/null
    F=$(find $ROOT/gpurun_out/pmc_cold -name "*counter_collection.csv" | head -1)
    if [ -z "$F" ]; then echo "$C rotate=$ROTN: no csv ($(tail -1 $OUT/pmc_cold.log | cut -c1-120))"; continue; fi
    python - "$F" $C $ROTN <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") == sys.argv[2]:
        k = r.get("Dispatch_Id")
        agg.setdefault(k, [r["Kernel_Name"], 0.0])[1] += float(r["Counter_Value"])
op = [v for n, v in agg.values() if "crop_bwd" in n]
fill = [v for n, v in agg.values() if "FillFunctor" in n]
m = lambda v: (sum(v) / len(v)) if v else float("nan")
print("%-36s outputs %s  roialign_bwd %14.1f   plain_fill %14.1f   (per launch, n = %d / %d)" % (sys.argv[2], "ROTATED(4x151MB)" if sys.argv[3] != "0" else "same buffer     ", m(op), m(fill), len(op), len(fill)))
PY
  done
done 2>&1 | tee $OUT/r05_bwd_cold_state_tcc_counters.txt
rm -rf $ROOT/gpurun_out/pmc_cold
