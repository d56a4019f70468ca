/*
 * mdt_launchers.h -- LINK-COMPATIBLE launcher symbols of the reference's raw-pointer layer (SURVEY.md 8(b), row "the
 * already-raw-pointer layer"), for a maintainer who keeps the reference's own C glue (crop_and_resize_gpu.c, nms_cuda.c) and
 * only swaps the object the glue is linked against.
 *
 * The reference declares the SAME three names once per extension, with different arity for 2D and 3D:
 *   cuda_functions/nms_3D/src/cuda/nms_kernel.h:11-12                 void _nms(int, float*, unsigned long long*, float)
 *   cuda_functions/nms_2D/src/cuda/nms_kernel.h:11-12                 (same prototype, 5-float rows)
 *   cuda_functions/roi_align_3D/roi_align/src/cuda/crop_and_resize_kernel.h:8-18   CropAndResizeLaucher / ...BackpropImageLaucher (3D)
 *   cuda_functions/roi_align_2D/roi_align/src/cuda/crop_and_resize_kernel.h:8-18   the same two names, 2D argument lists
 * and links each into its own extension (nms_3D/build.py:23-34, roi_align_3D/roi_align/build.py:28-40).  So there are two small
 * libraries here, one per dimensionality, each exporting the three names with exactly the reference's prototypes
 * (hipStream_t for cudaStream_t):
 *   medicaldetectiontoolkit_amd/libmdt_launchers_3d.so     (#define MDT_LAUNCHERS_DIM 3 before including this file)
 *   medicaldetectiontoolkit_amd/libmdt_launchers_2d.so     (#define MDT_LAUNCHERS_DIM 2)
 * Both are thin: they call libmdt_hip.so (include/mdt_hip.h), which they load through an $ORIGIN rpath.
 *
 * Behaviour kept from the reference launchers:
 *  - void return; a failed launch prints to stderr and exit(-1)s (crop_and_resize_kernel.cu:326-331, 354-359).  libmdt_hip.so
 *    itself never exits -- only these shims do, because their callers have no other way to learn about a failure.
 *  - _nms has no stream argument: it launches on the null stream like the reference (nms_kernel.cu:88-91); mask layout
 *    [boxes_num, DIVUP(boxes_num, 64)] row-major, every word the reference kernel writes (blocks below the diagonal included).
 *  - CropAndResizeLaucher: extrapolation_value accepted and ignored; rows with box_ind outside [0, batch) are left ZERO --
 *    the reference kernel skips them and relies on its caller's zero-fill (crop_and_resize_gpu.c:26-27); here they are written.
 *  - CropAndResizeBackpropImageLaucher: the reference accumulates with atomics into a grads_image its caller has zero-filled
 *    (crop_and_resize_gpu.c:61); this one WRITES every element of grads_image (no atomics, deterministic), so the caller's
 *    zero-fill is harmless and may be dropped.  A caller that pre-loads grads_image with non-zero values and expects
 *    accumulation is not supported (no such caller exists in the reference).
 */
#ifndef MDT_LAUNCHERS_H
#define MDT_LAUNCHERS_H

#include <hip/hip_runtime_api.h>

#ifndef MDT_LAUNCHERS_DIM
#error "define MDT_LAUNCHERS_DIM to 2 or 3 before including mdt_launchers.h (one library per dimensionality, like the reference)"
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define DIVUP(m,n) ((m) / (n) + ((m) % (n) > 0))

/* boxes_dev [boxes_num, 7] (3D) / [boxes_num, 5] (2D), sorted by descending score; mask_dev [boxes_num, DIVUP(boxes_num, 64)] */
void _nms(int boxes_num, float *boxes_dev, unsigned long long *mask_dev, float nms_overlap_thresh);

#if MDT_LAUNCHERS_DIM == 3
void CropAndResizeLaucher(
    const float *image_ptr, const float *boxes_ptr,
    const int *box_ind_ptr, int num_boxes, int batch, int image_height,
    int image_width, int image_zdepth, int crop_height, int crop_width, int crop_zdepth, int depth,
    float extrapolation_value, float *crops_ptr, hipStream_t stream);

void CropAndResizeBackpropImageLaucher(
    const float *grads_ptr, const float *boxes_ptr,
    const int *box_ind_ptr, int num_boxes, int batch, int image_height,
    int image_width, int image_zdepth, int crop_height, int crop_width, int crop_zdepth, int depth,
    float *grads_image_ptr, hipStream_t stream);
#else
void CropAndResizeLaucher(
    const float *image_ptr, const float *boxes_ptr,
    const int *box_ind_ptr, int num_boxes, int batch, int image_height,
    int image_width, int crop_height, int crop_width, int depth,
    float extrapolation_value, float *crops_ptr, hipStream_t stream);

void CropAndResizeBackpropImageLaucher(
    const float *grads_ptr, const float *boxes_ptr,
    const int *box_ind_ptr, int num_boxes, int batch, int image_height,
    int image_width, int crop_height, int crop_width, int depth,
    float *grads_image_ptr, hipStream_t stream);
#endif

#ifdef __cplusplus
}
#endif
#endif
