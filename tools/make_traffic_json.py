"""profiles/r03_pmc/traffic.json from the rocprofv3 --pmc counter csvs of tools/r03_profile_all.sh.

usage: python tools/make_traffic_json.py <dir with r03_pmc_{survey_random,trainlike}_{WRITE_SIZE,FETCH_SIZE}_counter_collection.csv> <out.json>
Per MI355X_MICROARCH.md (HBM section): WRITE_SIZE and FETCH_SIZE come from separate passes; FETCH_SIZE of wide coalesced reads
is doubled on gfx950; WRITE_SIZE is calibrated on the plain 150 994 944-byte torch fill dispatched in the same pass."""
import collections
import csv
import json
import os
import sys


def means(path, ctr):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == ctr:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    op = [v for k, vs in agg.items() if "crop_bwd" in k for v in vs]
    fill = [v for k, vs in agg.items() if ("FillFunctor" in k or "fill" in k.lower()) and "crop" not in k for v in vs]
    return (sum(op) / len(op) if op else None), (sum(fill) / len(fill) if fill else None), len(op)


def main():
    d, out = sys.argv[1], sys.argv[2]
    alg = 4 * 8 * 36 * 32 * 32 * 128 + 4 * 48 * 36 * 980 + 28 * 48
    res = {"_how": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE in separate passes (tools/gpu_pmc.sh via tools/r03_profile_all.sh), KB per launch "
                   "(mean over the launches of the pass); hbm_bytes = (WRITE_SIZE * cal + 2 * FETCH_SIZE) * 1024: FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (gfx950), cal = 147456 KB / WRITE_SIZE of the 150 994 944-byte torch fill of the same pass",
           "shape": "grads_image 8x36x32x32x128 fp32 (P2), pool 14x14x5, 48 RoIs, kernel crop_bwd_gather_kernel (one launch)"}
    for key, tag in (("survey_random_48_rois", "survey_random"), ("trainlike_48_rois", "trainlike")):
        w, wf, n = means(os.path.join(d, "r03_pmc_%s_WRITE_SIZE_counter_collection.csv" % tag), "WRITE_SIZE")
        f, _, _ = means(os.path.join(d, "r03_pmc_%s_FETCH_SIZE_counter_collection.csv" % tag), "FETCH_SIZE")
        if w is None or f is None:
            continue
        cal = (147456.0 / wf) if wf else 1.0
        res[key] = {"write_kb": round(w, 1), "fetch_kb": round(f, 1), "fill_write_kb_calibration": round(wf, 1) if wf else None,
                    "hbm_bytes": int(round((w * cal + 2 * f) * 1024)), "algorithmic_bytes": alg, "launches": n}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
