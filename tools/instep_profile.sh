#!/bin/bash
# (round 5 form: tools/r05_instep_profile.sh)  ROUND=r06 bash tools/instep_profile.sh [steps]: rocprofv3 kernel trace + stats and the two PMC passes of training steps with FULL RoI heads (tools/instep_heads_full.py)
# -> gpurun_out/r05/ (copied to profiles/r05/ by the builder).  Counters in their own runs, never with a trace (gpurun refuses the mix).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
STEPS=${1:-6}
ROUND=${ROUND:-r06}
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MDT_MIOPEN_SKIP_NAIVE=1
rm -rf $ROOT/gpurun_out/instep_kt $ROOT/gpurun_out/instep_w $ROOT/gpurun_out/instep_f
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/instep_kt -o instep -- python $ROOT/tools/instep_heads_full.py $STEPS > $OUT/instep_kt.log 2>&1 < /dev/null
grep '^{"steps"' $OUT/instep_kt.log | tail -1 > $OUT/instep_meta.json
cp $(find $ROOT/gpurun_out/instep_kt -name "*kernel_stats.csv" | head -1) $OUT/${ROUND}_instep_heads_full_training_steps_kernel_stats.csv
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/instep_w -o pmc -- python $ROOT/tools/instep_heads_full.py $STEPS > $OUT/instep_w.log 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/instep_f -o pmc -- python $ROOT/tools/instep_heads_full.py $STEPS > $OUT/instep_f.log 2>&1 < /dev/null
T=$(find $ROOT/gpurun_out/instep_kt -name "*kernel_trace.csv" | head -1)
W=$(find $ROOT/gpurun_out/instep_w -name "*counter_collection.csv" | head -1)
F=$(find $ROOT/gpurun_out/instep_f -name "*counter_collection.csv" | head -1)
python $ROOT/tools/instep_extract.py "$T" "$W" "$F" $OUT $STEPS $OUT/instep_meta.json $ROUND
python - "$W" "$F" $OUT $ROUND <<'PY'
import csv, sys
# keep only the RoIAlign / fill rows of the counter CSVs (the full files are tens of MB)
for path, tag in ((sys.argv[1], "WRITE_SIZE"), (sys.argv[2], "FETCH_SIZE")):
    rows = list(csv.DictReader(open(path)))
    keep = [r for r in rows if "crop_bwd_gather" in r["Kernel_Name"] or "FillFunctor" in r["Kernel_Name"]]
    with open("%s/%s_instep_heads_full_pmc_%s_counter_collection_roialign_and_fill_rows.csv" % (sys.argv[3], sys.argv[4], tag), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)
PY
rm -rf $ROOT/gpurun_out/instep_kt $ROOT/gpurun_out/instep_w $ROOT/gpurun_out/instep_f
ls -la $OUT
