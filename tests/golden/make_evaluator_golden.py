"""tests/golden/evaluator.pkl: synthetic results lists + the dataframes / APs the REFERENCE's evaluator.py produces for
them (imported from /root/reference).  Run once in the build container: timeout 300 python tests/golden/make_evaluator_golden.py"""
import logging, os, pickle, sys, types, warnings
warnings.filterwarnings("ignore")
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
import evaluator as ref_eval
sys.path.remove("/root/reference")
log = logging.getLogger("g"); log.addHandler(logging.NullHandler())
rng = np.random.default_rng(0)


def make_results(n_pat, dim, p_empty=0.2):
    out = []
    for p in range(n_pat):
        boxes = []
        n_gt = 0 if rng.random() < p_empty else int(rng.integers(1, 4))
        gts = []
        for _ in range(n_gt):
            c = rng.uniform(20, 100, size=dim); s = rng.uniform(6, 20, size=dim)
            co = [c[0] - s[0], c[1] - s[1], c[0] + s[0], c[1] + s[1]] + ([c[2] - s[2], c[2] + s[2]] if dim == 3 else [])
            gts.append(np.array(co)); boxes.append({"box_coords": np.array(co), "box_label": int(rng.integers(1, 3)), "box_type": "gt"})
        for _ in range(int(rng.integers(0, 9))):
            if gts and rng.random() < 0.6:
                co = gts[int(rng.integers(0, len(gts)))] + rng.normal(0, 2.5, size=2 * dim)
            else:
                c = rng.uniform(20, 100, size=dim); s = rng.uniform(6, 20, size=dim)
                co = np.array([c[0] - s[0], c[1] - s[1], c[0] + s[0], c[1] + s[1]] + ([c[2] - s[2], c[2] + s[2]] if dim == 3 else []))
            boxes.append({"box_coords": co, "box_score": float(rng.uniform(0.05, 1.0)), "box_pred_class_id": int(rng.integers(1, 3)), "box_type": "det"})
        out.append([[boxes], "pid_%d" % p])
    return out


gold = {}
for name, dim, ious in (("3d", 3, [0.1]), ("2d", 2, [0.1, 0.5])):
    cf = types.SimpleNamespace(ap_match_ious=ious, class_dict={1: "benign", 2: "malignant"}, fold=0, min_det_thresh=0.1)
    results = make_results(40, dim)
    ev = ref_eval.Evaluator(cf, log, mode="test")
    ev.evaluate_predictions(results)
    df = ev.test_df
    aps = {}
    for cl in (1, 2):
        cdf = df[df.pred_class == cl]
        aps[cl] = (float(ref_eval.get_roi_ap_from_df((cdf, 0.1, False))), float(ref_eval.get_roi_ap_from_df((cdf, 0.1, True))))
    gold[name] = {"results": results, "ious": ious, "df": df.to_dict(orient="list"), "aps": aps}
    print(name, len(df), aps)
pickle.dump(gold, open(os.path.join(HERE, "evaluator.pkl"), "wb"))
