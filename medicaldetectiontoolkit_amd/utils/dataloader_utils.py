"""Mirror of the one hot-path function of the reference's utils/dataloader_utils.py."""
import numpy as np


def get_patch_crop_coords(img, patch_size, min_overlap=30):
    """Patch tiling grid (utils/dataloader_utils.py:140-180).
    img: array (or anything with .shape) of spatial shape (y, x, (z)); returns int [n_patches, 2*dim] rows
    (y0, y1, x0, x1, (z0, z1))."""
    shape = tuple(img.shape)
    per_axis = []
    for d in range(len(shape)):
        n_patches = int(np.ceil(shape[d] / patch_size[d]))
        if n_patches == 1:
            per_axis.append([(0, shape[d])])
            continue
        center_dists = (shape[d] - patch_size[d]) / (n_patches - 1)
        if (patch_size[d] - center_dists) < min_overlap:
            n_patches += 1
            center_dists = (shape[d] - patch_size[d]) / (n_patches - 1)
        centers = np.round([(patch_size[d] / 2 + (center_dists * ii)) for ii in range(n_patches)])
        per_axis.append([(c - patch_size[d] / 2, c + patch_size[d] / 2) for c in centers])
    grid = []
    for ymin, ymax in per_axis[0]:
        for xmin, xmax in per_axis[1]:
            if len(per_axis) == 3 and patch_size[2] > 1:
                for zmin, zmax in per_axis[2]:
                    grid.append([ymin, ymax, xmin, xmax, zmin, zmax])
            elif len(per_axis) == 3 and patch_size[2] == 1:
                for zmin in range(shape[2]):
                    grid.append([ymin, ymax, xmin, xmax, zmin, zmin + 1])
            else:
                grid.append([ymin, ymax, xmin, xmax])
    return np.array(grid).astype(int)
