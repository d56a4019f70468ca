"""Writes tests/golden/predictor_reference.npz: the reference's OWN prediction pipeline
(/root/reference/predictor.py: Predictor.data_aug_forward :279, spatial_tiling_forward :370, get_mirrored_patch_crops
:777, apply_wbc_to_patient :514, weighted_box_clustering :597; utils/dataloader_utils.get_patch_crop_coords :140)
driven with the canned per-patch detections of tests/golden/predictor_inputs.py on a non-square volume with 4
mirrored passes, plus weighted_box_clustering at the full config-5 size (n = 45 000).
Run once in the build container:  timeout 900 python tests/golden/make_predictor_golden.py"""
import hashlib
import logging
import os
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
import predictor as ref_predictor            # noqa: E402  (the reference's)
import utils.dataloader_utils as ref_du      # noqa: E402
sys.path.remove(REF)
from tests.golden import predictor_inputs as pi   # noqa: E402
from tests.golden import wbc_inputs               # noqa: E402


def box_table(boxes):
    """box dict list -> sorted float table (coords6, score, class, pass, patch, centre factor, n_overlaps)"""
    rows = []
    for b in boxes:
        _, aug, pix = b["patch_id"].split("_")
        rows.append(list(np.asarray(b["box_coords"], dtype=np.float64)) + [b["box_score"], b["box_pred_class_id"], float(aug), float(pix),
                                                                           b["box_patch_center_factor"], b["box_n_overlaps"]])
    a = np.asarray(rows, dtype=np.float64)
    return a[np.lexsort((a[:, 6], a[:, 9], a[:, 8]))]


def main():
    out = {}
    cf = pi.make_cf()
    vol = pi.make_volume()
    coords = ref_du.get_patch_crop_coords(vol[0], cf.patch_size)
    out["patch_crop_coords"] = coords
    patches = np.array([vol[:, c[0]:c[1], c[2]:c[3], c[4]:c[5]] for c in coords])
    P = object.__new__(ref_predictor.Predictor)
    P.cf, P.mode, P.rank_ix, P.patched_patient = cf, "test", "0", True
    P.logger = logging.getLogger("pred_golden")
    net = pi.CannedNet()
    P.batch_tiling_forward = lambda batch: net.test_forward(batch)
    batch = {"data": patches, "patch_crop_coords": coords, "original_img_shape": (1,) + vol.shape}
    res = P.data_aug_forward(batch)
    raw = res["boxes"][0]
    out["raw_table"] = box_table(raw)
    wbc = ref_predictor.apply_wbc_to_patient([[raw], "pid", cf.class_dict, cf.wcs_iou, 4])[0][0]
    t = np.array([list(np.asarray(b["box_coords"], dtype=np.float64)) + [b["box_score"], b["box_pred_class_id"]] for b in wbc])
    out["wbc_table"] = t[np.lexsort((t[:, 6], t[:, 7]))]

    # tiler fuzz: hash of the concatenated grids over seeded shapes / patch sizes
    h = hashlib.sha256()
    for shape, ps in wbc_inputs.tiler_cases():
        h.update(np.ascontiguousarray(ref_du.get_patch_crop_coords(np.zeros(shape, np.uint8), ps).astype(np.int64)).tobytes())
    out["tiler_fuzz_sha256"] = np.frombuffer(h.digest(), dtype=np.uint8)

    # weighted box clustering at the config-5 size (75 patches x 30 dets x 4 TTA x 5 epochs)
    dets, pid_int = wbc_inputs.wbc_case(45000, 20, seed=5)
    pid_str = np.array(["%d_%d_%d" % (p // 300, (p // 75) % 4, p % 75) for p in pid_int])
    ks, kc = ref_predictor.weighted_box_clustering(dets.copy(), pid_str, 1e-5, 20)
    out["wbc45000_scores"] = np.array(ks)
    out["wbc45000_coords"] = np.array(kc).reshape(len(ks), 6)

    np.savez_compressed(os.path.join(HERE, "predictor_reference.npz"), **out)
    for k, v in out.items():
        print(k, np.asarray(v).shape)


if __name__ == "__main__":
    main()
