"""Shared synthetic-input builders for the tests.  The generators live beside the synthetic batches in
medicaldetectiontoolkit_amd/utils/synthetic_data.py (the measurement path -- bench.py, tools/ -- must not depend on the test package);
the tests keep importing them from here."""
from medicaldetectiontoolkit_amd.utils.synthetic_data import nms_boxes, random_boxes_2d, random_boxes_3d, trainlike_rois_3d  # noqa: F401
