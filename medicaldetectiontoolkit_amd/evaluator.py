"""Box <-> ground-truth matching and COCO-style ROI AP: the metric core of the reference's evaluator.py
(evaluate_predictions :39-192, get_roi_ap_from_df :361-399, compute_roi_ap :402-436), so parity can be shown at
the metric level.  Offline host code (numpy / pandas), not on the throughput path; plots, patient-level AUC and the
results.txt bookkeeping of the reference are out of scope.
"""
import numpy as np
import pandas as pd


def compute_overlaps(boxes1, boxes2):
    """utils/model_utils.py:83-110: IoU matrix [n1, n2], continuous coordinates (no +1), float64."""
    b1 = np.asarray(boxes1, dtype=np.float64)[:, None, :]
    b2 = np.asarray(boxes2, dtype=np.float64)[None, :, :]
    dim = b1.shape[-1] // 2
    y1 = np.maximum(b2[..., 0], b1[..., 0])
    y2 = np.minimum(b2[..., 2], b1[..., 2])
    x1 = np.maximum(b2[..., 1], b1[..., 1])
    x2 = np.minimum(b2[..., 3], b1[..., 3])
    inter = np.maximum(x2 - x1, 0) * np.maximum(y2 - y1, 0)
    v1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    v2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    if dim == 3:
        z1 = np.maximum(b2[..., 4], b1[..., 4])
        z2 = np.minimum(b2[..., 5], b1[..., 5])
        inter = inter * np.maximum(z2 - z1, 0)
        v1 = v1 * (b1[..., 5] - b1[..., 4])
        v2 = v2 * (b2[..., 5] - b2[..., 4])
    return inter / (v2 + v1 - inter)


def _match_element(boxes, cl, match_iou):
    """One batch element, one class: rows (score, label, det_type) in the reference's append order (:84-166)."""
    tar = np.array([b["box_coords"] for b in boxes if b["box_type"] == "gt" and b["box_label"] == cl])
    cand = np.array([b["box_coords"] for b in boxes if b["box_type"] == "det" and b["box_pred_class_id"] == cl])
    scores = np.array([b["box_score"] for b in boxes if b["box_type"] == "det" and b["box_pred_class_id"] == cl])
    rows = []
    has_c, has_t = 0 not in cand.shape, 0 not in tar.shape
    if has_c and has_t:
        ov = compute_overlaps(cand, tar)
        mx = ov.max(1)
        match = np.nonzero(mx > match_iou)[0]
        non_match = np.nonzero(mx <= match_iou)[0]
        match_gt = ov[match].argmax(1) if match.size else np.array([], dtype=np.int64)
        non_match_gt = np.array([g for g in range(tar.shape[0]) if g not in match_gt])
        # several detections on one GT: only the best scoring counts as true positive (:105-121)
        fp_double = []
        for g in np.unique(match_gt):
            c = match[match_gt == g]
            if c.size > 1:
                best = c[np.argmax(scores[c])]
                fp_double += [i for i in c if i != best]
        fp_double = [i for i in match if i in fp_double]             # reference order: order of match_cand_ixs
        match = np.array([i for i in match if i not in fp_double], dtype=np.int64)
        rows += [(scores[i], 0, "det_fp") for i in fp_double]
        rows += [(scores[i], 1, "det_tp") for i in match]
        rows += [(scores[i], 0, "det_fp") for i in non_match]
        rows += [(0, 1, "det_fn")] * int(non_match_gt.shape[0])
    elif has_c:
        rows += [(s, 0, "det_fp") for s in scores]
    elif has_t:
        rows += [(0, 1, "det_fn")] * int(tar.shape[0])
    return rows


def evaluate_predictions(results_list, cf, mode="test"):
    """results_list: 'test' / 'val_patient' form [[results_0, pid_0], ...] with results = list over batch elements of
    box-dict lists; 'train' / 'val_sampling' form [[[results per element], [pids]], ...].  Returns the reference's
    internal dataframe (pred_score, class_label, pred_class, pid, det_type, fold, match_iou)."""
    if mode in ("train", "val_sampling"):
        elements = [[b] for item in results_list for b in item[0]]
        pids = [pid for item in results_list for pid in item[1]]
    else:
        elements = [item[0] for item in results_list]
        pids = [item[1] for item in results_list]
    cols = {"pred_score": [], "class_label": [], "pred_class": [], "pid": [], "det_type": [], "match_iou": []}
    for match_iou in cf.ap_match_ious:
        for cl in list(cf.class_dict.keys()):
            for pix, pid in enumerate(pids):
                n_before = len(cols["pid"])
                for b_boxes in elements[pix]:
                    for s, label, typ in _match_element(b_boxes, cl, match_iou):
                        cols["pred_score"].append(s); cols["class_label"].append(label); cols["pred_class"].append(cl)
                        cols["pid"].append(pid); cols["det_type"].append(typ); cols["match_iou"].append(match_iou)
                if len(cols["pid"]) == n_before:      # empty patient: dummy true negative so it stays in the stats (:167-175)
                    cols["pred_score"].append(0); cols["class_label"].append(0); cols["pred_class"].append(cl)
                    cols["pid"].append(pid); cols["det_type"].append("patient_tn"); cols["match_iou"].append(match_iou)
    df = pd.DataFrame()
    df["pred_score"] = cols["pred_score"]
    df["class_label"] = cols["class_label"]
    df["pred_class"] = cols["pred_class"]
    df["pid"] = cols["pid"]
    df["det_type"] = cols["det_type"]
    df["fold"] = getattr(cf, "fold", 0)
    df["match_iou"] = cols["match_iou"]
    return df


def compute_roi_ap(df, all_p):
    """evaluator.py:402-436 (101-point interpolated AP, pycocotools style), incl. the truncated tail when recall never
    reaches a threshold."""
    tp = df.class_label.values
    fp = (tp == 0) * 1
    R = np.linspace(.0, 1, 101, endpoint=True)
    tp_sum, fp_sum = np.cumsum(tp), np.cumsum(fp)
    nd = len(tp)
    rc = tp_sum / all_p
    pr = (tp_sum / (fp_sum + tp_sum)).tolist()
    q = np.zeros((len(R),)).tolist()
    for i in range(nd - 1, 0, -1):
        if pr[i] > pr[i - 1]:
            pr[i - 1] = pr[i]
    inds = np.searchsorted(rc, R, side="left")
    for ri, pi in enumerate(inds):
        if pi >= nd:
            break
        q[ri] = pr[pi]
    return np.mean(q)


def get_roi_ap_from_df(df, det_thresh, per_patient_ap=False):
    """evaluator.py:361-399."""
    aps = []
    for match_iou in df.match_iou.unique():
        iou_df = df[df.match_iou == match_iou]
        groups = [iou_df[iou_df.pid == pid] for pid in df.pid.unique()] if per_patient_ap else [iou_df]
        for g in groups:
            all_p = len(g[g.class_label == 1])
            d = g[(g.det_type == "det_fp") | (g.det_type == "det_tp")].sort_values("pred_score", ascending=False)
            d = d[d.pred_score > det_thresh]
            if per_patient_ap:
                if len(d) == 0 and all_p == 0:
                    continue
                aps.append(0 if all_p == 0 else compute_roi_ap(d, all_p))
            elif all_p > 0:
                aps.append(compute_roi_ap(d, all_p))
    return np.mean(aps)
