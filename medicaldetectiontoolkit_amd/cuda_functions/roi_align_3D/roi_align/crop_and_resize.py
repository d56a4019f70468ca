"""Drop-in for the reference's cuda_functions/roi_align_3D/roi_align/crop_and_resize.py
(CropAndResizeFunction :10-51, CropAndResize :54-69), running the gfx950 HIP kernels.

Semantics = the reference's CUDA kernel (the behaviour the models were trained with):
half-pixel sampling with clamping, one sample per bin, extrapolation_value accepted
and ignored, rows with box_ind outside [0, B) left at zero.
"""
import torch.nn as nn

from ..._roi_align_impl import CropAndResizeFunctionBase


class CropAndResizeFunction(CropAndResizeFunctionBase):
    _dim = 3

    def __init__(self, crop_height, crop_width, crop_zdepth, extrapolation_value=0):
        super(CropAndResizeFunction, self).__init__(crop_height, crop_width, crop_zdepth, extrapolation_value)


class CropAndResize(nn.Module):
    def __init__(self, crop_height, crop_width, crop_zdepth, extrapolation_value=0):
        super(CropAndResize, self).__init__()
        self.fn = CropAndResizeFunction(crop_height, crop_width, crop_zdepth, extrapolation_value)

    def forward(self, image, boxes, box_ind):
        return self.fn(image, boxes, box_ind)
