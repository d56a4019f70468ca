"""Reads the rocprofv3 outputs of tools/instep_heads_full.py (tools/r05_instep_profile.sh) and writes
  <out>/r05_instep_heads_full_roialign_bwd_launches.csv   one row per crop_bwd_gather_kernel launch of the profiled steps
  <out>/traffic_instep.json                               PMC bytes per launch of the MASK HEAD's pyramid backward (bench.py reads it)
usage: instep_extract.py <kernel_trace.csv> <WRITE_SIZE counter csv> <FETCH_SIZE counter csv> <out dir> <steps> <tool stdout json>"""
import collections
import csv
import json
import os
import sys

trace, wcsv, fcsv, out, steps, meta = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6]
ROUND = sys.argv[7] if len(sys.argv) > 7 else "r05"
os.makedirs(out, exist_ok=True)
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
bwd = [r for r in rows if "crop_bwd_gather_kernel" in r["Kernel_Name"]]
bwd = bwd[-2 * steps:]                       # the profiled steps: two launches each (mask head first, classifier second)
with open(os.path.join(out, "%s_instep_heads_full_roialign_bwd_launches.csv" % ROUND), "w") as f:
    f.write("# crop_bwd_gather_kernel launches of the last %d training steps of tools/instep_heads_full.py (RoI heads full), rocprofv3 --kernel-trace\n" % steps)
    f.write("step,which,duration_us,grid,workgroup,lds_bytes,vgprs\n")
    for i, r in enumerate(bwd):
        f.write("%d,%s,%.2f,%d,%d,%s,%s\n" % (i // 2, "mask_head_pool_14x14x5" if i % 2 == 0 else "classifier_pool_7x7x3",
                                           (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                           int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]),
                                           int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]),
                                           r.get("LDS_Block_Size", ""), r.get("VGPR_Count", "")))
mask_us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in bwd[0::2]]
cls_us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in bwd[1::2]]


def pmc(path, ctr):
    rs = [r for r in csv.DictReader(open(path)) if r.get("Counter_Name") == ctr]
    key = "Dispatch_Id" if rs and "Dispatch_Id" in rs[0] else None
    if key:
        rs.sort(key=lambda r: int(r[key]))
    agg = collections.OrderedDict()
    for r in rs:                              # one row per (dispatch, dimension instance): sum the instances of a dispatch
        k = r.get("Dispatch_Id", id(r))
        agg.setdefault(k, [r["Kernel_Name"], 0.0])[1] += float(r["Counter_Value"])
    op = [v for n, v in agg.values() if "crop_bwd_gather_kernel" in n][-2 * steps:]
    fill = [v for n, v in agg.values() if "FillFunctor" in n and abs(v) > 0]
    return op[0::2], op[1::2], fill


wm, wc, wfill = pmc(wcsv, "WRITE_SIZE")
fm, fc, _ = pmc(fcsv, "FETCH_SIZE")
# calibration: the largest fills of the pass are the 150 994 944-byte zero_() calls of the tool (one per step)
big = sorted(wfill)[-steps:] if wfill else []
cal = (147456.0 / (sum(big) / len(big))) if big else 1.0
m = json.loads(open(meta).read().strip().splitlines()[-1])
n_valid = sum(m["valid_rois"]) / max(1, len(m["valid_rois"]))
maps = 4 * 8 * 36 * (32 * 32 * 128 + 16 * 16 * 64 + 8 * 8 * 32 + 4 * 4 * 16)
alg = maps + 4 * n_valid * 36 * 980 + 36 * 48
res = {"_how": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE in SEPARATE passes over tools/instep_heads_full.py (training steps with full RoI heads); per launch of the "
               "mask head's mdt_pyramid_roi_align_backward (first crop_bwd_gather_kernel of each step), KB, mean over the profiled steps; hbm_bytes = (WRITE_SIZE * cal "
               "+ 2 * FETCH_SIZE) * 1024: FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), cal = 147456 KB / WRITE_SIZE of the 150 994 944-byte torch zero_() "
               "dispatched before every step of the same pass",
       "in_training_step_heads_full": {
           "write_kb": round(sum(wm) / len(wm), 1) if wm else None, "fetch_kb": round(sum(fm) / len(fm), 1) if fm else None,
           "fill_write_kb_calibration": round(sum(big) / len(big), 1) if big else None,
           "hbm_bytes": int(round(((sum(wm) / len(wm)) * cal + 2 * (sum(fm) / len(fm))) * 1024)) if (wm and fm) else None,
           "algorithmic_bytes": int(alg), "valid_rois": round(n_valid, 2), "launches": len(wm),
           "rocprofv3_kernel_us_mean": round(sum(mask_us) / len(mask_us), 2) if mask_us else None,
           "rocprofv3_kernel_us_all": [round(v, 2) for v in mask_us]},
       "classifier_head_launch": {"write_kb": round(sum(wc) / len(wc), 1) if wc else None, "fetch_kb": round(sum(fc) / len(fc), 1) if fc else None,
                                  "rocprofv3_kernel_us_mean": round(sum(cls_us) / len(cls_us), 2) if cls_us else None}}
json.dump(res, open(os.path.join(out, "traffic_instep.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
