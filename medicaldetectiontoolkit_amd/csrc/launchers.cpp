// launchers.cpp -- the reference's raw-pointer launcher names over libmdt_hip.so (include/mdt_launchers.h).
// Compiled twice: -DMDT_LAUNCHERS_DIM=3 -> libmdt_launchers_3d.so, -DMDT_LAUNCHERS_DIM=2 -> libmdt_launchers_2d.so.
// Host code only: every kernel lives in libmdt_hip.so.
#include <stdio.h>
#include <stdlib.h>
#include "mdt_launchers.h"
#include "mdt_hip.h"

namespace {
void die_on(int rc, const char *what)
{
    if (rc == MDT_OK) return;
    // the reference launchers: fprintf(stderr, ...); exit(-1);  (crop_and_resize_kernel.cu:326-331)
    fprintf(stderr, "%s failed : %s\n", what, mdt_error_string(rc));
    exit(-1);
}
}  // namespace

extern "C" {

void _nms(int boxes_num, float *boxes_dev, unsigned long long *mask_dev, float nms_overlap_thresh)
{
#if MDT_LAUNCHERS_DIM == 3
    die_on(mdt_nms_mask_full_3d(boxes_dev, boxes_num, nms_overlap_thresh, MDT_NMS_RULE_GT, mask_dev, nullptr), "_nms");
#else
    die_on(mdt_nms_mask_full_2d(boxes_dev, boxes_num, nms_overlap_thresh, MDT_NMS_RULE_GT, mask_dev, nullptr), "_nms");
#endif
}

#if MDT_LAUNCHERS_DIM == 3
void CropAndResizeLaucher(const float *image_ptr, const float *boxes_ptr, const int *box_ind_ptr, int num_boxes, int batch,
                          int image_height, int image_width, int image_zdepth, int crop_height, int crop_width, int crop_zdepth,
                          int depth, float extrapolation_value, float *crops_ptr, hipStream_t stream)
{
    die_on(mdt_crop_and_resize_3d_forward(image_ptr, boxes_ptr, box_ind_ptr, num_boxes, batch, image_height, image_width,
                                          image_zdepth, crop_height, crop_width, crop_zdepth, depth, extrapolation_value,
                                          crops_ptr, (void *)stream), "CropAndResizeKernel");
}

void CropAndResizeBackpropImageLaucher(const float *grads_ptr, const float *boxes_ptr, const int *box_ind_ptr, int num_boxes,
                                       int batch, int image_height, int image_width, int image_zdepth, int crop_height,
                                       int crop_width, int crop_zdepth, int depth, float *grads_image_ptr, hipStream_t stream)
{
    // no workspace argument in the reference prototype: the single-launch kernel needs none (num_boxes <= 128); beyond it the
    // two-kernel form would want one, so the any-shape exact-order kernel takes over
    int rc = mdt_crop_and_resize_3d_backward(grads_ptr, boxes_ptr, box_ind_ptr, num_boxes, batch, image_height, image_width,
                                             image_zdepth, crop_height, crop_width, crop_zdepth, depth, grads_image_ptr,
                                             nullptr, 0, (void *)stream);
    if (rc == MDT_ERR_WORKSPACE_TOO_SMALL)
        rc = mdt_crop_and_resize_3d_backward_ordered(grads_ptr, boxes_ptr, box_ind_ptr, num_boxes, batch, image_height,
                                                     image_width, image_zdepth, crop_height, crop_width, crop_zdepth, depth,
                                                     grads_image_ptr, (void *)stream);
    die_on(rc, "CropAndResizeBackpropImageKernel");
}
#else
void CropAndResizeLaucher(const float *image_ptr, const float *boxes_ptr, const int *box_ind_ptr, int num_boxes, int batch,
                          int image_height, int image_width, int crop_height, int crop_width, int depth,
                          float extrapolation_value, float *crops_ptr, hipStream_t stream)
{
    die_on(mdt_crop_and_resize_2d_forward(image_ptr, boxes_ptr, box_ind_ptr, num_boxes, batch, image_height, image_width,
                                          crop_height, crop_width, depth, extrapolation_value, crops_ptr, (void *)stream),
           "CropAndResizeKernel");
}

void CropAndResizeBackpropImageLaucher(const float *grads_ptr, const float *boxes_ptr, const int *box_ind_ptr, int num_boxes,
                                       int batch, int image_height, int image_width, int crop_height, int crop_width, int depth,
                                       float *grads_image_ptr, hipStream_t stream)
{
    int rc = mdt_crop_and_resize_2d_backward(grads_ptr, boxes_ptr, box_ind_ptr, num_boxes, batch, image_height, image_width,
                                             crop_height, crop_width, depth, grads_image_ptr, nullptr, 0, (void *)stream);
    if (rc == MDT_ERR_WORKSPACE_TOO_SMALL)
        rc = mdt_crop_and_resize_2d_backward_ordered(grads_ptr, boxes_ptr, box_ind_ptr, num_boxes, batch, image_height,
                                                     image_width, crop_height, crop_width, depth, grads_image_ptr, (void *)stream);
    die_on(rc, "CropAndResizeBackpropImageKernel");
}
#endif

}  // extern "C"
