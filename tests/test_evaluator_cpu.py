"""CPU: the metric core (box<->GT matching, 101-point ROI AP) against dataframes / APs produced by the reference's
evaluator.py (tests/golden/make_evaluator_golden.py)."""
import os
import pickle
import types

import numpy as np
import pytest

from medicaldetectiontoolkit_amd import evaluator

GOLD = pickle.load(open(os.path.join(os.path.dirname(__file__), "golden", "evaluator.pkl"), "rb"))


@pytest.mark.parametrize("name", ["3d", "2d"])
def test_matching_dataframe_and_ap_equal_reference(name):
    g = GOLD[name]
    cf = types.SimpleNamespace(ap_match_ious=g["ious"], class_dict={1: "benign", 2: "malignant"}, fold=0)
    df = evaluator.evaluate_predictions(g["results"], cf, mode="test")
    want = g["df"]
    assert list(df.columns) == list(want.keys())
    for col in want:
        if col == "pred_score":
            assert np.allclose(np.array(df[col], dtype=np.float64), np.array(want[col], dtype=np.float64), rtol=0, atol=0), col
        else:
            assert list(df[col]) == list(want[col]), col
    for cl, (ap, ap_pp) in g["aps"].items():
        cdf = df[df.pred_class == cl]
        assert abs(evaluator.get_roi_ap_from_df(cdf, 0.1, False) - ap) < 1e-12
        assert abs(evaluator.get_roi_ap_from_df(cdf, 0.1, True) - ap_pp) < 1e-12


def test_compute_overlaps_matches_definition():
    a = np.array([[0, 0, 10, 10, 0, 10], [5, 5, 15, 15, 5, 15]], dtype=np.float64)
    ov = evaluator.compute_overlaps(a, a)
    assert np.allclose(np.diag(ov), 1.0) and abs(ov[0, 1] - 125.0 / (2000 - 125)) < 1e-12
