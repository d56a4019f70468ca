"""Patch tiling of a patient volume -- the one hot-path function of the reference's utils/dataloader_utils.py
(get_patch_crop_coords, :140-180; consumed by Predictor.spatial_tiling_forward, predictor.py:398-440)."""
import numpy as np


def _axis_intervals(extent, patch, min_overlap):
    """[lo, hi) float bounds of the patches along one axis.  One patch spans the axis when it fits; otherwise the two
    outer patches touch the borders and the rest are spread evenly, one extra patch being added when neighbours
    would overlap by less than `min_overlap` (:151-165).  Centres are rounded with numpy's round-half-to-even."""
    count = -(-int(extent) // int(patch))
    if count == 1:
        return np.zeros(1), np.full(1, float(extent))
    stride = (extent - patch) / (count - 1)
    if patch - stride < min_overlap:
        count += 1
        stride = (extent - patch) / (count - 1)
    centres = np.round(patch / 2 + stride * np.arange(count))
    return centres - patch / 2, centres + patch / 2


def get_patch_crop_coords(img, patch_size, min_overlap=30):
    """img: anything with a spatial .shape (y, x[, z]).  Returns int [n_patches, 2*dim] rows (y0, y1, x0, x1[, z0, z1]),
    y slowest and z fastest; with patch_size[2] == 1 every z slice is its own patch (2D models on 3D volumes)."""
    shape = tuple(int(s) for s in img.shape)
    bounds = []
    for axis, extent in enumerate(shape):
        if axis == 2 and patch_size[2] == 1:
            lo = np.arange(extent, dtype=np.float64)
            bounds.append((lo, lo + 1))
        else:
            bounds.append(_axis_intervals(extent, patch_size[axis], min_overlap))
    pick = np.indices([len(lo) for lo, _ in bounds]).reshape(len(bounds), -1)      # C order: last axis fastest
    cols = []
    for axis, (lo, hi) in enumerate(bounds):
        cols += [lo[pick[axis]], hi[pick[axis]]]
    return np.stack(cols, axis=1).astype(int)
