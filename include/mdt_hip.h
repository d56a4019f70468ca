/*
 * mdt_hip.h -- C ABI of libmdt_hip.so: the MI355X (gfx950) native hot path of
 * MIC-DKFZ/medicaldetectiontoolkit (2D/3D RoIAlign fwd/bwd, 2D/3D NMS, anchor
 * generation, anchor<->GT matching, box decode/clip, weighted box clustering).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch types.  Every pointer is a
 *    DEVICE pointer unless its name ends in _host.
 *  - `stream` is a hipStream_t passed as void* (0 = the null stream).  All work
 *    is enqueued on it; no call synchronises, allocates or frees.
 *  - The caller owns every buffer, including `workspace` (size from the
 *    matching *_workspace_bytes query; must be 16-byte aligned).
 *  - Return value: MDT_OK (0) or a negative MDT_ERR_* code.  The library never
 *    calls exit() (the reference does: crop_and_resize_kernel.cu:326-331).
 *  - Tensor layouts are the reference's: feature maps (b, c, y, x[, z]) with the
 *    last axis contiguous; boxes (y1, x1, y2, x2[, z1, z2]).
 *
 * Each entry point cites the reference interface it replaces (paths relative
 * to the reference checkout).
 */
#ifndef MDT_HIP_H
#define MDT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDT_OK 0
#define MDT_ERR_INVALID_ARGUMENT (-1)
#define MDT_ERR_WORKSPACE_TOO_SMALL (-2)
#define MDT_ERR_LAUNCH_FAILED (-3)
#define MDT_ERR_UNSUPPORTED (-4)

/* library / build identification; the string names the gfx target */
const char *mdt_version(void);
/* text for an MDT_ERR_* code */
const char *mdt_error_string(int code);

/* ------------------------------------------------------------------------- */
/* RoIAlign ("crop and resize"), one bilinear/trilinear sample per bin        */
/* ------------------------------------------------------------------------- */

/*
 * Replaces CropAndResizeLaucher (3D)
 *   cuda_functions/roi_align_3D/roi_align/src/cuda/crop_and_resize_kernel.h:8-12
 *   (kernel: crop_and_resize_kernel.cu:12-151) together with the output
 *   zero-fill of crop_and_resize_gpu_forward (src/crop_and_resize_gpu.c:26-27):
 *   every element of `crops` is written exactly once (rows whose box_ind is
 *   outside [0,batch) are written as zeros), so the caller need not clear it.
 * image [batch, depth, H, W, D] f32; boxes [num_boxes, 6] f32 normalised
 * (y1,x1,y2,x2,z1,z2); box_ind [num_boxes] i32; crops [num_boxes, depth, ch, cw, cd].
 * extrapolation_value is accepted and ignored, exactly like the reference kernel.
 */
int mdt_crop_and_resize_3d_forward(
    const float *image, const float *boxes, const int *box_ind,
    int num_boxes, int batch, int image_height, int image_width, int image_zdepth,
    int crop_height, int crop_width, int crop_zdepth, int depth,
    float extrapolation_value, float *crops, void *stream);

/* bf16-input forms of the two forward entry points (SURVEY.md 8a precision note (iii), BASELINE config 5: bf16 autocast
 * inference): `image` holds bfloat16 values (raw 16-bit patterns), every load is widened exactly to fp32 and the
 * interpolation and the output stay fp32 -- results equal the fp32 entry point on the widened tensor bit for bit,
 * with half the gathered bytes and without a conversion pass over the feature map.  Forward only. */
int mdt_crop_and_resize_3d_forward_bf16(
    const uint16_t *image, const float *boxes, const int *box_ind,
    int num_boxes, int batch, int image_height, int image_width, int image_zdepth,
    int crop_height, int crop_width, int crop_zdepth, int depth, float *crops, void *stream);
int mdt_crop_and_resize_2d_forward_bf16(
    const uint16_t *image, const float *boxes, const int *box_ind,
    int num_boxes, int batch, int image_height, int image_width,
    int crop_height, int crop_width, int depth, float *crops, void *stream);

/* uint8-input forms (round 4): the GT masks of a batch (Appendix B: 'roi_masks' uint8) are cropped to the mask targets
 * (models/mrcnn.py:551-563: CropAndResizeFunction(mask_shape...)(gt_masks, boxes, box_ids)) straight from their uint8 storage; each
 * byte is widened exactly to fp32, the interpolation is the fp32 one (bit-equal to the fp32 kernel on image.float()). */
int mdt_crop_and_resize_3d_forward_u8(
    const uint8_t *image, const float *boxes, const int *box_ind, int num_boxes, int batch, int H, int W, int D,
    int ch, int cw, int cd, int depth, float *crops, void *stream);
int mdt_crop_and_resize_2d_forward_u8(
    const uint8_t *image, const float *boxes, const int *box_ind, int num_boxes, int batch, int H, int W,
    int ch, int cw, int depth, float *crops, void *stream);

/*
 * Replaces CropAndResizeBackpropImageLaucher (3D)
 *   crop_and_resize_kernel.h:14-18 (kernel: crop_and_resize_kernel.cu:154-304)
 *   together with BOTH zero-fills of grads_image (crop_and_resize.py:40 and
 *   crop_and_resize_gpu.c:61): every byte of grads_image is written exactly once, without atomics.
 * grads [num_boxes, depth, ch, cw, cd]; grads_image [batch, depth, H, W, D].
 *
 * ONE launch, no workspace, no atomics (csrc/roi_align_bwd_v3.hip): workgroups of a "zero" role stream 16-byte zero
 * stores over everything outside the index bounding boxes of the RoIs (a bitmap every workgroup rebuilds from `boxes`); one
 * workgroup per (batch element, channel) of a "scatter" role computes the rest from LDS as wave-uniform separable passes and one
 * ordered sum per voxel over the RoIs covering it.  Deterministic run to run; sums are reassociated relative to the reference's
 * flat 8-corner scatter, so values agree to fp32 rounding (bar: 1e-4).  2D maps are its W = 1 case.
 * Beyond its budgets (more than 128 RoIs, pool extents or maps beyond its LDS plan) the call runs the exact-order kernel
 * (mdt_crop_and_resize_*_backward_ordered below: any shape, bit-exact against the sequential oracle, slower).
 * `workspace` is not used any more (round 5: the two-kernel form that needed one is A/B history in libmdt_hip_ab.so); the argument and
 * the query (which answers a token 256) stay for ABI stability -- NULL / 0 is fine.
 */
size_t mdt_crop_and_resize_backward_workspace_bytes(int dim, int num_boxes, int depth,
                                                   int image_height, int image_width, int image_zdepth,
                                                   int crop_height, int crop_width, int crop_zdepth);
int mdt_crop_and_resize_3d_backward(
    const float *grads, const float *boxes, const int *box_ind,
    int num_boxes, int batch, int image_height, int image_width, int image_zdepth,
    int crop_height, int crop_width, int crop_zdepth, int depth,
    float *grads_image, void *workspace, size_t workspace_bytes, void *stream);


/* All pyramid levels in ONE launch: replaces the per-level loop of mrcnn.py:373-457 (pyramid_roi_align: level rule :403,
 * one CropAndResizeFunction call per level :431-437, torch.cat + sort back :440-455).
 *   images[l]        device pointer of level l's map [batch, depth, H[l], W[l](, D[l])], fp32 (bf16 = 0) or bf16 (bf16 = 1)
 *   boxes [N, 2*dim] normalised (y1, x1, y2, x2[, z1, z2]); batch_ix [N] (outside [0, batch): the row is skipped);
 *   level [N]        index into images of the level the RoI is pooled on (outside [0, n_levels): the row is skipped)
 *   crops [N, depth, ch, cw(, cd)]: row n is RoI n pooled on its level, i.e. already in the input order; skipped rows = 0.
 * H, W, D, images, grads_images are HOST arrays of n_levels entries (D ignored for dim == 2).  n_levels <= 5.
 * forward: bit-exact vs crop_and_resize_kernel.cu:7-118 per RoI.  backward: grads_images[l] (fp32, fully written) gets
 * the scatter of the RoIs with level == l; same numerics contract as mdt_crop_and_resize_*_backward.  Returns
 * MDT_ERR_UNSUPPORTED when a level does not fit the single-launch kernel (contiguous extent not a multiple of 8, map
 * not 16-byte aligned, num_boxes > 128, pool beyond the LDS budget): call mdt_crop_and_resize_*_backward per level then. */
int mdt_pyramid_roi_align_forward(int dim, int n_levels, const void *const *images, int bf16, const int *H, const int *W,
                                  const int *D, const float *boxes, const int *batch_ix, const int *level, int num_boxes,
                                  int batch, int depth, int crop_height, int crop_width, int crop_zdepth, float *crops,
                                  void *stream);
/* The forward on CHANNELS-LAST maps (round 5): images[l] is level l's map stored [batch, H[l], W[l], D[l], depth] (torch channels_last_3d: what the
 * convolution path produces), fp32 (bf16 = 0) or bf16 (bf16 = 1); depth % 4 == 0; 3D only.  A corner voxel is `depth` contiguous values serving every
 * channel: 16-byte loads straight from global memory, no row-major copy of the pyramid per forward.  crops [N, depth, ch, cw, cd] in the reference
 * layout, bit-equal to mdt_pyramid_roi_align_forward on the same maps in row-major storage.  The backward stays in the reference layout. */
int mdt_pyramid_roi_align_forward_cl(int n_levels, const void *const *images, int bf16, const int *H, const int *W, const int *D,
                                     const float *boxes, const int *batch_ix, const int *level, int num_boxes, int batch, int depth,
                                     int ch, int cw, int cd, float *crops, void *stream);
int mdt_pyramid_roi_align_backward(int dim, int n_levels, const float *grads, const float *boxes, const int *batch_ix,
                                   const int *level, int num_boxes, int batch, int depth, const int *H, const int *W,
                                   const int *D, int crop_height, int crop_width, int crop_zdepth,
                                   float *const *grads_images, void *stream);
/* The same ON TOP of what the maps already hold (another RoI head's gradient written by an earlier launch): nothing is zero-filled, the quads the
 * RoIs touch are read-modify-written.  Two heads pooling one pyramid cost one full write of the maps, not two plus a dense add. */
int mdt_pyramid_roi_align_backward_accumulate(int dim, int n_levels, const float *grads, const float *boxes, const int *batch_ix,
                                              const int *level, int num_boxes, int batch, int depth, const int *H, const int *W,
                                              const int *D, int crop_height, int crop_width, int crop_zdepth,
                                              float *const *grads_images, void *stream);

/* Exact-order form: gather kernel that adds, per voxel, the terms in exactly the order a
 * sequential out_idx loop would (corner order of crop_and_resize_kernel.cu:256-301), so the
 * result equals the fp32 CPU oracle bit for bit.  Any shape, no workspace, slower. */
int mdt_crop_and_resize_3d_backward_ordered(
    const float *grads, const float *boxes, const int *box_ind,
    int num_boxes, int batch, int image_height, int image_width, int image_zdepth,
    int crop_height, int crop_width, int crop_zdepth, int depth,
    float *grads_image, void *stream);


/* 2D twins: cuda_functions/roi_align_2D/roi_align/src/cuda/crop_and_resize_kernel.h
 * (kernels crop_and_resize_kernel.cu:11-99, 102-194; glue crop_and_resize_gpu.c:7-67).
 * image [batch, depth, H, W]; boxes [num_boxes, 4]; crops [num_boxes, depth, ch, cw]. */
int mdt_crop_and_resize_2d_forward(
    const float *image, const float *boxes, const int *box_ind,
    int num_boxes, int batch, int image_height, int image_width,
    int crop_height, int crop_width, int depth,
    float extrapolation_value, float *crops, void *stream);

int mdt_crop_and_resize_2d_backward(
    const float *grads, const float *boxes, const int *box_ind,
    int num_boxes, int batch, int image_height, int image_width,
    int crop_height, int crop_width, int depth,
    float *grads_image, void *workspace, size_t workspace_bytes, void *stream);

int mdt_crop_and_resize_2d_backward_ordered(
    const float *grads, const float *boxes, const int *box_ind,
    int num_boxes, int batch, int image_height, int image_width,
    int crop_height, int crop_width, int depth,
    float *grads_image, void *stream);

/* ------------------------------------------------------------------------- */
/* Fused convolution epilogues (FPN / ResNet conv path)                       */
/* ------------------------------------------------------------------------- */

/* Channels-last max pooling of the ResNet stem: MaxPool3d(kernel 3, stride (2, 2, 1), padding 1), models/backbone.py:77-79
 * (torch.nn.functional.max_pool3d semantics: floor mode, first maximum in (y, x, z) scan order, NaN wins).
 *   x      [batch, Y, X, Z, channels] fp32 -- i.e. the channels_last_3d storage of a [batch, channels, Y, X, Z] tensor
 *   y      [batch, OY, OX, Z, channels], OY = (Y - 1) / 2 + 1, OX = (X - 1) / 2 + 1
 *   argmax [same as y] uint8: window tap 0..26 (dy * 9 + dx * 3 + dz) of the maximum, consumed by the backward
 * backward: gx [batch, Y, X, Z, channels] fully written; gather over the windows containing a voxel: no atomics, fixed
 * summation order (torch's kernel uses atomics).  Not part of the reference's native interface: it replaces the torch op
 * around the MIOpen convolutions, like the epilogues below. */
int mdt_maxpool3d_k3s221_cl_forward(const float *x, float *y, unsigned char *argmax, int batch, int Y, int X, int Z, int channels,
                                    void *stream);
int mdt_maxpool3d_k3s221_cl_backward(const float *gy, const unsigned char *argmax, float *gx, int batch, int Y, int X, int Z,
                                     int channels, void *stream);

/* out[ci][co][taps - 1 - t] = w[co][ci][t] for a dense filter w [cout, cin, taps] (channels_last = 0) or its channels-last
 * storage [cout, taps, cin] -> [cin, taps, cout] (channels_last = 1): the filter of the forward convolution that computes
 * a unit-stride convolution's input gradient (utils/fused_epilogue.py). */
int mdt_filter_flip_transpose(const float *w, float *out, int cout, int cin, int taps, int channels_last, void *stream);
/* The same for MANY filters in one launch (a training step flips ~60, once per step since weights change only in the optimizer):
 * records_dev = DEVICE array of n_records records of 40 bytes each, laid out as
 *     { const float *w; float *out; int cout; int cin; int taps; int channels_last; long long first; }
 * with first = number of output elements of all earlier records (ascending, record 0 has first == 0); total_elements = their sum.  The
 * caller owns the table and the buffers (utils/fused_epilogue.py builds it once and re-uses it every step). */
int mdt_filter_flip_transpose_batched(const void *records_dev, int n_records, long long total_elements, void *stream);

/*
 * y = act(x + bias[c] (+ residual)) in one pass; y may alias x.  The convolutions stay on MIOpen (torch); this
 * replaces what torch runs around every one of them in the reference's graph: the broadcast bias add, the residual add
 * of ResBlock.forward (models/backbone.py:203-205) / the top-down add of FPN.forward (:147-153), and the ReLU of
 * NDConvGenerator's Sequential (utils/model_utils.py:770-779).
 * n elements, channel(i) = (i / inner) % channels: inner = 1 for channels_last(_3d) storage, prod(spatial) for NC(D)HW.
 */
int mdt_bias_act_forward(float *y, const float *x, const float *bias, const float *residual /* or NULL */,
                         long long n, int channels, long long inner, int relu, void *stream);

/* gx = gy * (relu ? y > 0 : 1), gbias[c] = sum of gx over channel c -- one pass plus a small second stage, no atomics,
 * fixed summation order (run-to-run deterministic).  y (the forward output) is only read when relu != 0. */
size_t mdt_bias_act_backward_workspace_bytes(long long n, int channels, long long inner);
int mdt_bias_act_backward(float *gx, const float *gy, const float *y, float *gbias,
                          long long n, int channels, long long inner, int relu,
                          void *workspace, size_t workspace_bytes, void *stream);
/* relu == 0: gx may be NULL -- the input gradient IS gy then and nothing is stored (round 6: the bias-only layers' backward was
 * writing a copy of gy, 151 MB on the FPN's P2 map). */
/* The same in ONE launch for channels-last storage (inner == 1): the block that draws the last ticket folds the per-block partials into gbias
 * inside the launch (agent-scope release / acquire around the ticket).  `ticket`: one device int the CALLER owns, 0 on entry, left 0 on exit
 * (launches that share a ticket must be ordered on one stream).  inner != 1 runs the two-launch form.  The summation order differs from
 * mdt_bias_act_backward's (both fixed); no state inside the library. */
int mdt_bias_act_backward_ticket(float *gx, const float *gy, const float *y, float *gbias,
                                 long long n, int channels, long long inner, int relu,
                                 void *workspace, size_t workspace_bytes, int *ticket, void *stream);

/*
 * y = x + bias[c] + coarse[b][y / sy][x / sx][z / sz][c] on channels-last fp32 storage ([batch][Y][X][Z][channels], channels % 4 == 0;
 * coarse: [batch][Y / sy][X / sx][Z / sz][channels]); y may alias x.  The FPN's top-down step (models/backbone.py:147-153:
 * P_conv1(c) + F.interpolate(p, scale_factor=2), mode 'nearest') without materialising the up-sampled map.  2D maps: Z = 1, sz = 1.
 * Same value as mdt_bias_act_forward on the materialised residual (same order of the two additions).
 */
int mdt_bias_act_forward_upsampled_supported(int channels, long long n);
int mdt_bias_act_forward_upsampled(float *y, const float *x, const float *bias, const float *coarse, int batch, int Y, int X, int Z, int channels,
                                   int sy, int sx, int sz, void *stream);

/*
 * Backward of a bias-only epilogue whose output gradient arrives ROW-MAJOR ([batch][channels][inner]) while the layer runs channels-last:
 * gx[b][v][c] = gy[b][c][v] and gbias[c] = sum of gy over channel c in one pass (fixed summation order).  channels <= 48.
 * Replaces tensor.contiguous(memory_format=channels_last_3d) + mdt_bias_act_backward in front of the FPN's P_conv2 layers
 * (models/backbone.py:155-160), whose output gradients come from the RoIAlign backward's row-major maps.
 */
int mdt_bias_grad_to_channels_last_supported(int channels);
size_t mdt_bias_grad_to_channels_last_workspace_bytes(int batch, int channels, long long inner);
int mdt_bias_grad_to_channels_last(float *gx, const float *gy, float *gbias, int batch, int channels, long long inner,
                                   void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- */
/* Non-maximum suppression                                                    */
/* ------------------------------------------------------------------------- */

#define MDT_NMS_RULE_GT 0 /* suppress when IoU >  thresh: GPU rule, nms_kernel.cu:71 */
#define MDT_NMS_RULE_GE 1 /* suppress when IoU >= thresh: CPU rule, nms.c:64          */

/*
 * Pairwise suppression mask.  Replaces _nms
 *   cuda_functions/nms_3D/src/cuda/nms_kernel.h:11-12 (kernel nms_kernel.cu:30-78;
 *   IoU with the +1 pixel convention, nms_kernel.cu:16-28).
 * dets_sorted [n, 7] = (y1,x1,y2,x2,z1,z2,score) (2D: [n,5]) already sorted by
 * descending score; mask [n, ceil(n/64)] u64, bit j of word (i, c) set iff
 * box 64c+j (> i) is suppressed by box i.  Only blocks on or above the diagonal
 * are computed; words below it are written as 0 (the reference fills them but
 * never reads them, nms_cuda.c:54).
 */
int mdt_nms_mask_3d(const float *dets_sorted, int n, float thresh, int rule,
                    unsigned long long *mask, void *stream);
int mdt_nms_mask_2d(const float *dets_sorted, int n, float thresh, int rule,
                    unsigned long long *mask, void *stream);
/* the same mask INCLUDING the blocks below the diagonal, word for word what the reference kernel writes (its early-out
 * `if (row_start > col_start) return;` is commented out, nms_kernel.cu:35): in a lower block, bit j of word (i, c) is set
 * iff IoU(box i, box 64c+j) passes the rule.  This is what the link-compatible `_nms` of include/mdt_launchers.h runs. */
int mdt_nms_mask_full_3d(const float *dets_sorted, int n, float thresh, int rule,
                         unsigned long long *mask, void *stream);
int mdt_nms_mask_full_2d(const float *dets_sorted, int n, float thresh, int rule,
                         unsigned long long *mask, void *stream);

/*
 * Device-resident NMS.  Replaces gpu_nms
 *   cuda_functions/nms_3D/src/nms_cuda.h:1 (nms_cuda.c:17-67: mask launch, D2H
 *   copy of the whole mask, host greedy scan) -- here mask build AND greedy
 *   scan run on the device, nothing is copied to the host.
 * keep [>= min(n, max_keep>0 ? max_keep : n)] i64 receives positions in the
 * sorted list in ascending order; num_out [1] i32 (device) the count.
 * max_keep <= 0: full scan (the reference's behaviour).  max_keep > 0: stop
 * after that many boxes were kept -- identical to truncating the full result
 * (models/mrcnn.py:348 keeps only the first proposal_count).
 * workspace: mdt_nms_workspace_bytes(n) bytes.
 */
size_t mdt_nms_workspace_bytes(int n);
int mdt_nms_3d(const float *dets_sorted, int n, float thresh, int rule, int max_keep,
               long long *keep, int *num_out,
               void *workspace, size_t workspace_bytes, void *stream);
int mdt_nms_2d(const float *dets_sorted, int n, float thresh, int rule, int max_keep,
               long long *keep, int *num_out,
               void *workspace, size_t workspace_bytes, void *stream);

/*
 * Batched form used by the proposal layer (models/mrcnn.py:317-348 loops over
 * batch elements and calls nms once per element): `batch` independent problems
 * of n boxes each, dets_sorted [batch, n, 7|5], keep [batch, keep_stride] i64
 * (rows padded with -1), num_out [batch] i32.
 * workspace: batch * mdt_nms_workspace_bytes(n).
 */
int mdt_nms_3d_batched(const float *dets_sorted, int batch, int n, float thresh, int rule,
                       int max_keep, long long *keep, int keep_stride, int *num_out,
                       void *workspace, size_t workspace_bytes, void *stream);
int mdt_nms_2d_batched(const float *dets_sorted, int batch, int n, float thresh, int rule,
                       int max_keep, long long *keep, int keep_stride, int *num_out,
                       void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- */
/* Box decode + clip (fused)                                                  */
/* ------------------------------------------------------------------------- */
/*
 * Replaces apply_box_deltas_{2D,3D} followed by clip_boxes_{2D,3D}
 *   utils/model_utils.py:318-370, 374-398 as used in models/mrcnn.py:337-344:
 *   out = clip(decode(anchors[order[i]], deltas[order[i]] * std_dev), window),
 *   optionally with the score appended (out_stride 2*dim+1) so the result is
 *   directly the `dets_sorted` input of mdt_nms_*.
 * boxes [A, 2*dim] f32; deltas [A, 2*dim] f32; order [n] i64 or NULL (identity);
 * scores [n] f32 or NULL (already gathered, i.e. scores[i] belongs to order[i]);
 * std_dev_host [2*dim], window_host [2*dim] = (y1,x1,y2,x2[,z1,z2]) HOST floats;
 * out [n, out_stride].
 */
int mdt_decode_clip_boxes(const float *boxes, const float *deltas, const long long *order,
                          const float *scores, int n, int dim,
                          const float *std_dev_host, const float *window_host,
                          float *out, int out_stride, void *stream);

/* ------------------------------------------------------------------------- */
/* Anchors                                                                    */
/* ------------------------------------------------------------------------- */
/*
 * One pyramid level of generate_anchors_3D / generate_anchors
 *   utils/model_utils.py:230-272, 190-226.  Anchor row index =
 *   ((y*X + x)*Z + z)*K + k with K = n_ratios*n_scales, k = ratio_idx*n_scales +
 *   scale_idx; heights = s/sqrt(r), widths = s*sqrt(r), depth = scales_z[k % n_scales]
 *   (faithful to np.tile, :249); centres = arange(0,shape,anchor_stride)*feature_stride.
 * All *_host arrays are HOST doubles.  out [n_anchors, 2*dim] f64 (device), rows
 * (y1,x1,y2,x2[,z1,z2]); out_f32 (optional, may be NULL) receives the same rows
 * rounded to f32 (models/mrcnn.py:846).
 * dim = 2: shape_host = (Y,X), scales_z_host ignored.  dim = 3: shape_host = (Y,X,Z).
 */
int mdt_generate_anchors(int dim, const double *scales_xy_host, const double *scales_z_host,
                         int n_scales, const double *ratios_host, int n_ratios,
                         const int *shape_host, double feature_stride_xy,
                         double feature_stride_z, int anchor_stride,
                         double *out, float *out_f32, void *stream);

/*
 * Anchor <-> ground-truth matching, steps 1-3 of gt_anchor_matching
 *   utils/model_utils.py:505-563 (IoU: compute_overlaps / compute_iou_{2D,3D},
 *   :35-110, float64, no +1 convention).
 * anchors [A, 2*dim] f64; gt_boxes [G, 2*dim] f64; gt_class_ids [G] i32 or NULL
 * (all 1, the RPN case).  Outputs (device):
 *   matches [A] i32     -1 negative (max IoU < neg_thresh), 0 neutral, >0 class id
 *   iou_argmax [A] i32  index of the GT box with max IoU (first on ties, np.argmax)
 *   iou_max [A] f64     may be NULL
 *   gt_best_anchor [G] i32  argmax over anchors per GT (first on ties)
 * The random positive subsampling and the delta targets of the few kept
 * positives (:566-617) stay in the host mirror (they need numpy/torch RNG).
 * workspace: mdt_anchor_match_workspace_bytes(A, G).
 */
size_t mdt_anchor_match_workspace_bytes(int n_anchors, int n_gt);
int mdt_anchor_match(const double *anchors, int n_anchors, int dim,
                     const double *gt_boxes, const int *gt_class_ids, int n_gt,
                     double neg_thresh, double pos_thresh,
                     int *matches, int *iou_argmax, double *iou_max, int *gt_best_anchor,
                     void *workspace, size_t workspace_bytes, void *stream);

/*
 * The same matching for a whole batch in ONE launch pair, with the GT counts read on the DEVICE: replaces the per-element loop of
 * models/mrcnn.py:894 / models/retina_unet.py:408-420 (one gt_anchor_matching call per batch element on the host).
 *   gt_boxes [batch, gmax, 2*dim] f64 (rows >= n_gt[e] are ignored), gt_class_ids [batch, gmax] i32 or NULL,
 *   n_gt_dev [batch] i32 DEVICE array (0 = "gt_boxes is None": every anchor of the element is negative, :524-526),
 *   matches / iou_argmax [batch, A] i32, iou_max [batch, A] f64 or NULL, gt_best_anchor [batch, gmax] i32.
 * No host-side count enters the launch, so a training step that calls it can be captured in a hipGraph.
 * workspace: mdt_anchor_match_batched_workspace_bytes(A, batch, gmax).
 */
size_t mdt_anchor_match_batched_workspace_bytes(int n_anchors, int batch, int gmax);
int mdt_anchor_match_batched(const double *anchors, int n_anchors, int dim, int batch,
                             const double *gt_boxes, const int *gt_class_ids, const int *n_gt_dev, int gmax,
                             double neg_thresh, double pos_thresh,
                             int *matches, int *iou_argmax, double *iou_max, int *gt_best_anchor,
                             void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- */
/* Weighted box clustering                                                    */
/* ------------------------------------------------------------------------- */
/*
 * Replaces weighted_box_clustering  predictor.py:597-706 (float64).
 * dets_sorted [n, 2*dim+3] f64 rows (coords, score, patch_center_factor,
 * n_overlaps) sorted by descending score; patch_ids [n] i32 (the reference uses
 * strings; any injective integer relabelling in [0, n_patch_ids) works).
 * Outputs: out_scores [n] f64, out_coords [n, 2*dim] f64 hold the clusters whose
 * averaged score > 0.01 (:697) in creation order; num_out [1] i32 (device).
 * workspace: mdt_wbc_workspace_bytes(n, n_patch_ids).
 */
size_t mdt_wbc_workspace_bytes(int n, int n_patch_ids);
int mdt_weighted_box_clustering(const double *dets_sorted, const int *patch_ids, int n, int dim,
                                int n_patch_ids, double thresh, double n_ens,
                                double *out_scores, double *out_coords, int *num_out,
                                void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- */
/* 2D -> 3D merge of per-slice detections                                     */
/* ------------------------------------------------------------------------- */
/*
 * Replaces nms_2to3D  predictor.py:710-773 (float64).
 * dets_sorted [n, 6] f64 rows (y1, x1, y2, x2, score, slice_id) sorted by descending score; slice ids are integers in
 * [0, n_slices).  Outputs: keep [n] i64 = positions in the sorted list of the cluster cores, in creation order;
 * keep_z [n, 2] f64 = (z1, z2) = (first connected slice - 1, last connected slice + 1); num_out [1] i32 (device).
 * workspace: mdt_nms_2to3d_workspace_bytes(n).
 */
size_t mdt_nms_2to3d_workspace_bytes(int n);
int mdt_nms_2to3d(const double *dets_sorted, int n, int n_slices, double thresh,
                  long long *keep, double *keep_z, int *num_out,
                  void *workspace, size_t workspace_bytes, void *stream);

/* ---- 1x1(x1) convolution weight gradient (csrc/conv1x1_wgrad.hip) -------------------------------------------------------
 * dW[co][ci] = sum_v grad_out[v][co] * x[v][ci] over the n_voxels rows of channels-last activations (what
 * torch.ops.aten.convolution_backward(..., output_mask = [0, 1, 0]) returns for a 1x1(x1) convolution; the reference gets it
 * from cuDNN through nn.Conv3d, utils/model_utils.py:751).  fp32 MFMA, deterministic (no atomics); workspace = per-workgroup
 * partial sums.  MDT_ERR_UNSUPPORTED never occurs for c_out, c_in <= 4096. */
size_t mdt_conv1x1_wgrad_workspace_bytes(long long n_voxels, int c_out, int c_in);
int mdt_conv1x1_wgrad(const float *grad_out, const float *x, float *grad_weight, long long n_voxels, int c_out, int c_in,
                      void *workspace, size_t workspace_bytes, void *stream);

/* ---- linear x2 up-sampling of (y, x) on channels-last storage (csrc/upsample.hip) -------------------------------------------
 * F.interpolate(x, scale_factor=(2, 2, 1), mode='trilinear', align_corners=False) on a channels_last_3d tensor (the decoder's
 * P2 -> P1 -> P0 path, models/backbone.py:209-217 Interpolate) and scale 2 'bilinear' on channels_last 2D maps: the tensor
 * is [batch, Y, X, inner] with inner = Z * C (3D) or C (2D); out is [batch, 2Y, 2X, inner].  The backward is the exact
 * adjoint in gather form (deterministic). */
int mdt_upsample2x_yx_cl_forward(const float *in, float *out, long long batch, int Y, int X, long long inner, void *stream);
int mdt_upsample2x_yx_cl_backward(const float *grad_out, float *grad_in, long long batch, int Y, int X, long long inner, void *stream);

/* ---- 3x3x3 convolution with few channels (csrc/conv3x3x3_small.hip) ----------------------------------------------------------
 * out[b, y, x, z, co] = sum_{tap, ci} in[b, (y, x, z) + tap - 1, ci] * w[tap][ci][co] (zero padding, stride 1), channels-last fp32,
 * c_in even <= 32, c_out <= 32, Z a multiple of 32: the 18 -> 18 ResBlock.conv2 layers of stage C2 (models/backbone.py:186-190),
 * which cuDNN / MIOpen compute through nn.Conv3d (utils/model_utils.py:751).  w is the filter in [27][c_in][c_out] order
 * (w_torch.permute(2, 3, 4, 1, 0)).  `..._supported` answers 1 when the shape fits; else MDT_ERR_UNSUPPORTED. */
int mdt_conv3x3x3_small_supported(int Y, int X, int Z, int c_in, int c_out);
int mdt_conv3x3x3_small_forward(const float *in, const float *w_tap_ci_co, float *out, int batch, int Y, int X, int Z,
                                int c_in, int c_out, void *stream);
/* the same with the layer's bias add and ReLU (model_utils.py:732-781: conv -> [norm] -> relu; norm = None in the LIDC configs) in the
 * kernel's epilogue: out = act(conv(in) + bias), bias [c_out] or NULL, relu 0 / 1 -- no separate pass over the output */
int mdt_conv3x3x3_small_forward_bias_act(const float *in, const float *w_tap_ci_co, const float *bias, int relu, float *out, int batch, int Y, int X, int Z,
                                         int c_in, int c_out, void *stream);
/* weight gradient of the same layer, dW[tap][ci][co] = sum_v in[v + tap - 1][ci] * grad_out[v][co] (what
 * aten.convolution_backward(..., output_mask = [0, 1, 0]) returns, in [27][c_in][c_out] order); workspace = one partial per
 * workgroup tile; deterministic */
size_t mdt_conv3x3x3_small_wgrad_workspace_bytes(int batch, int Y, int X, int c_in, int c_out);
int mdt_conv3x3x3_small_wgrad(const float *in, const float *grad_out, float *grad_w_tap_ci_co, int batch, int Y, int X, int Z,
                              int c_in, int c_out, void *workspace, size_t workspace_bytes, void *stream);

/* ---- weight gradient of the one-channel stem convolution (csrc/conv_stem_wgrad.hip) ------------------------------------------
 * dW[co][ky, kx, kz] = sum_{b, oy, ox, oz} grad_out[b, oy, ox, oz][co] * x_padded[b, sy * oy + ky, sx * ox + kx, oz + kz]: the
 * backward-weights of C1 = conv(1 -> 18, ks 7, stride (2, 2, 1), pad 3) (models/backbone.py:66-68), which the reference gets from
 * cuDNN.  grad_out channels-last [B, OY, OX, OZ, c_out] (c_out <= 32, OZ % 8 == 0); x_padded = the ONE-channel input zero-padded
 * by k / 2 on every face, [B, YP, XP, ZP]; k odd, k^3 <= 384; z stride 1.  fp32 MFMA, deterministic. */
size_t mdt_conv_stem_wgrad_workspace_bytes(int c_out, int k);
int mdt_conv_stem_wgrad(const float *grad_out, const float *x_padded, float *grad_weight, int batch, int OY, int OX, int OZ,
                        int c_out, int k, int sy, int sx, int YP, int XP, int ZP, void *workspace, size_t workspace_bytes, void *stream);

/* ---- stride-(2, 2, 1) convolutions with MANY input channels (csrc/conv_s221.hip) -----------------------------------------------
 * The Retina U-Net's C1 = conv(18 -> 18, ks 7, stride (2, 2, 1), pad 3) on the full-resolution C0 output (models/backbone.py:84), which
 * the reference gets from cuDNN through nn.Conv3d.  Channels-last fp32 storage everywhere; k odd, pad = k / 2, Y and X even.
 *
 * mdt_s2d221_input: x [B, Y, X, Z, C] -> xs [B, (Y + 2p) / 2, (X + 2p) / 2, Z + 2p, 4C], the 2 x 2 (y, x) phases of the zero-padded input
 *   as channels (c, py, px): the layer becomes a 4C -> c_out, ((k+1)/2, (k+1)/2, k), unit-stride convolution.
 * mdt_s2d221_fold_input_grad: the gradient w.r.t. xs -> the gradient w.r.t. x (the inverse gather, padding rows dropped).
 * mdt_conv_s221_wgrad: grad_weight[co][ky][kx][kz][ci] = sum_{b, oy, ox, z} grad_out[b, oy, ox, z][co] * x[b, 2 oy + ky - p, 2 ox + kx - p,
 *   z + kz - p][ci]  (= aten.convolution_backward(..., output_mask = [0, 1, 0]) in channels_last_3d storage).  fp32 MFMA, deterministic;
 *   supported (`..._supported` answers 1) when k * c_in <= 128, c_out <= 32, Z % 16 == 0; otherwise MDT_ERR_UNSUPPORTED. */
int mdt_s2d221_input(const float *x, float *xs, int batch, int channels, int Y, int X, int Z, int k, void *stream);
int mdt_s2d221_fold_input_grad(const float *grad_xs, float *grad_x, int batch, int channels, int Y, int X, int Z, int k, void *stream);
int mdt_conv_s221_wgrad_supported(int batch, int Y, int X, int Z, int c_in, int c_out, int k);
size_t mdt_conv_s221_wgrad_workspace_bytes(int batch, int Y, int X, int Z, int c_in, int c_out, int k);
int mdt_conv_s221_wgrad(const float *grad_out, const float *x, float *grad_weight, int batch, int Y, int X, int Z, int c_in, int c_out, int k,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ---- forward of the one-channel stem convolution (csrc/conv_stem_fwd.hip) -----------------------------------------------------
 * out[b, oy, ox, oz][co] = (bias[co] +) sum_{ky, kx, kz} weight[co][ky, kx, kz] * x_padded[b, 2 oy + ky, 2 ox + kx, oz + kz]
 * (optionally ReLU): C1 = conv(1 -> 18, ks 7, stride (2, 2, 1), pad 3) of models/backbone.py:66-68, which the reference gets from
 * cuDNN through nn.Conv3d.  x_padded = the ONE-channel input zero-padded by 3 on every face, [B, YP, XP, ZP = OZ + 6], 8-byte
 * aligned; out channels-last [B, OY, OX, OZ, c_out]; bias may be NULL.  Supported: k = 7, stride (2, 2), c_out <= 32, OX % 4 == 0,
 * OZ in {32, 64, 128} (`..._supported` answers 1); otherwise MDT_ERR_UNSUPPORTED.  fp32 MFMA. */
int mdt_conv_stem_forward_supported(int OY, int OX, int OZ, int c_out, int k, int sy, int sx);
int mdt_conv_stem_forward(const float *x_padded, const float *weight, const float *bias, float *out, int batch, int OY, int OX, int OZ,
                          int c_out, int k, int sy, int sx, int YP, int XP, int ZP, int relu, void *stream);

/*
 * 1x1(x1) convolution forward WITH its epilogue on channels-last fp32 rows (csrc/conv1x1_fwd.hip):
 *   out[v][n] = act((sum_k x[v][k] * w[n][k] + bias[n]) (+ res[v][n])),  act = ReLU when relu != 0;  w: [c_out][c_in] (the module's filter, 1x1 taps dropped).
 * The bottleneck layers of the reference's ResBlock (models/backbone.py:197-206: conv1 + ReLU, conv3 + residual + ReLU) in ONE pass over their operands
 * instead of a convolution plus mdt_bias_act_forward; fp32 MFMA (exact products, fixed order).  Shapes: (c_in, c_out) in {(18, 72), (72, 18), (36, 144)}
 * -- the C2 / C3 stages of the LIDC backbone; everything else: MDT_ERR_UNSUPPORTED (the caller keeps MIOpen + the epilogue kernel).
 * res may be NULL; out must not alias x (it may alias res).  x 16-byte aligned.
 */
int mdt_conv1x1_forward_supported(int c_in, int c_out);
int mdt_conv1x1_forward(const float *x, const float *w, const float *bias, const float *res, float *out, long long n_voxels, int c_in, int c_out, int relu,
                        void *stream);

/*
 * Backward of the same layer's epilogue AND its input gradient in one pass (channels-last fp32 rows):
 *   g[v][n] = gy[v][n] * (y[v][n] > 0)  (y == NULL: no activation -- g is gy, `g` is not written and may be NULL),
 *   gx[v][k] = sum_n g[v][n] * w[n][k],   gbias[n] = sum_v g[v][n]   (fixed summation order, deterministic).
 * Replaces mdt_bias_act_backward + the input-gradient convolution of conv3 in a C2 ResBlock (models/backbone.py:203-205): (c_in, c_out) = (18, 72) only.
 * gy, y, g 16-byte aligned.
 */
int mdt_conv1x1_backward_supported(int c_in, int c_out);
size_t mdt_conv1x1_backward_workspace_bytes(long long n_voxels, int c_out);
int mdt_conv1x1_backward(const float *gy, const float *y, const float *w, float *g, float *gx, float *gbias, long long n_voxels, int c_in, int c_out,
                         void *workspace, size_t workspace_bytes, void *stream);

/*
 * The RPN's two 1x1 heads (mrcnn.py:40-86) on the RAW output of conv_shared, forward only, channels-last fp32 rows:
 *   y[v][n] = sum_k relu(h[v][k] + bias_shared[k]) * w[n][k] + bias[n];   w = [conv_class.weight ; conv_bbox.weight] ([n_class + n_box][hidden]), bias alike.
 * Class logits go to logits[b][anchor_offset + v * A + a][0..1], box deltas to deltas[b][anchor_offset + v * A + a][0..2 dim - 1] (A = n_class / 2 anchors per
 * voxel; v = the voxel's index inside its batch element) -- the level's slice of the tensors the reference builds with torch.cat over the pyramid levels
 * (mrcnn.py:1030-1033; [batch][anchors_total][2] and [batch][anchors_total][n_box / A]).  One pass over the hidden map instead of conv_shared's bias / ReLU
 * pass, the head convolution, its bias pass, two slicing copies and the concatenations.  hidden == 128, n_class + n_box <= 32; h 16-byte aligned.
 */
int mdt_rpn_heads_forward_supported(int hidden, int n_class, int n_box);
int mdt_rpn_heads_forward(const float *h, const float *bias_shared, const float *w, const float *bias, float *logits, float *deltas, int batch,
                          long long voxels_per_element, int hidden, int n_class, int n_box, long long anchors_total, long long anchor_offset, void *stream);

/*
 * The first layer of the stride-1 backbone (models/backbone.py:60-63 with operate_stride1 -- the Retina U-Net: C0[0] = conv(1 -> 18, ks 3, pad 1) + ReLU on the
 * full-resolution one-channel volume), csrc/conv_c0.hip.  x: [batch][Y][X][Z] fp32, w: the filter as [27][18] for the forward (taps outermost: the module's [18][1][3][3][3] filter transposed), y: [batch][Y][X][Z][18]
 * (channels-last -- what the next layer reads; the library path produces it row-major and converts).
 *   forward:  y = act(conv(x) + bias)
 *   backward: grad_weight [18][27] (the module's layout) and grad_bias [18] (may be NULL) from gy and the forward output y (the ReLU mask; y may be NULL when relu == 0), one pass,
 *             fp32 MFMA, fixed summation order.  The input (the image) has no gradient.
 * c_in == 1, c_out == 18, k == 3, Z % 32 == 0 (mdt_conv_c0_supported); everything else: the caller keeps MIOpen.
 */
int mdt_conv_c0_supported(int c_in, int c_out, int k, int Z);
int mdt_conv_c0_forward(const float *x, const float *w, const float *bias, int relu, float *y, int batch, int Y, int X, int Z, int c_out, void *stream);
size_t mdt_conv_c0_wgrad_workspace_bytes(int batch, int Y, int X, int Z);
int mdt_conv_c0_backward(const float *gy, const float *y, const float *x, int relu, float *grad_weight, float *grad_bias, int batch, int Y, int X, int Z,
                         int c_out, void *workspace, size_t workspace_bytes, void *stream);

/*
 * 3x3x3 convolution (stride 1, padding 1) between a 36-channel and a 2-channel full-resolution map, channels-last fp32 (csrc/conv_seg.hip): the Retina U-Net's
 * segmentation branch  final_conv(P0_conv2(p0))  (models/retina_unet.py:483-486, models/backbone.py:160-176) as ONE layer -- P0_conv2 has no activation and no
 * other reader, so the two linear layers compose: W'[s][ci][tap] = sum_co Wf[s][co] W2[co][ci][tap], b' = Wf b2 + bf (the caller composes them with
 * differentiable tensor ops and hands W' here in the layouts below).
 *   forward:     y[v][s] = b[s] + sum x[v + tap][ci] wt[tap][ci][s]                wt = W'.permute(2, 3, 4, 1, 0) contiguous ([27][36][2])
 *   input grad:  gx[v][ci] = sum g[v + tap][s] wd[tap][s][ci]                      wd = flip(W', taps).permute(2, 3, 4, 0, 1) contiguous ([27][2][36])
 *   weight grad: grad_weight[s][ci][tap], grad_bias[s] (may be NULL) from g and x; fp32 MFMA, fixed summation order
 * c_big == 36, c_small == 2, Y % 2 == 0, X % 4 == 0, Z % 32 == 0 (mdt_conv_seg_supported).
 */
int mdt_conv_seg_supported(int c_big, int c_small, int Y, int X, int Z);
int mdt_conv_seg_forward(const float *x, const float *wt, const float *bias, float *y, int batch, int Y, int X, int Z, int c_big, int c_small, void *stream);
int mdt_conv_seg_input_grad(const float *g, const float *wd, float *gx, int batch, int Y, int X, int Z, int c_big, int c_small, void *stream);
size_t mdt_conv_seg_wgrad_workspace_bytes(int batch, int Y, int X, int Z);
int mdt_conv_seg_weight_grad(const float *g, const float *x, float *grad_weight, float *grad_bias, int batch, int Y, int X, int Z, int c_big, int c_small,
                             void *workspace, size_t workspace_bytes, void *stream);

/* ---- input gradient of a 1x1(x1) convolution added to another gradient of the same tensor (csrc/epilogue.hip, round 4) ------------------
 * out[v][ci] = res[v][ci] + sum_co gy[v][co] * w[co][ci] over n_voxels channels-last rows (res may be NULL: plain input gradient).
 * What autograd does in two steps for a ResBlock input (models/backbone.py:197-205: x feeds conv1 and the residual add): the
 * convolution's backward-data (cuDNN in the reference) and the accumulation of the two gradients.  w = the layer's weight [c_out, c_in]
 * (1x1x1 kernel squeezed); c_out even, c_in % 4 == 0, c_out * c_in <= 12288 (`..._supported`).  fp32, fixed summation order. */
int mdt_conv1x1_dgrad_add_supported(int c_out, int c_in);
int mdt_conv1x1_dgrad_add(const float *gy, const float *w, const float *res, float *out, long long n_voxels, int c_out, int c_in, void *stream);

/* ---- Adam over flat fp32 buffers (csrc/adam.hip) ------------------------------------------------------------------------------
 * One step of torch.optim.Adam (exec.py:39: Adam(lr = cf.learning_rate[0], weight_decay = cf.weight_decay); no amsgrad) for n
 * parameters whose values, gradients and moment estimates are four flat arrays: step >= 1 is the number of this update (bias
 * corrections 1 - beta^step), weight_decay is the L2 term added to the gradient; the hyper-parameters are doubles (derived scalars are
 * formed in double and rounded to fp32 once, like torch's python-side arithmetic).  grad_div > 0: every gradient is divided by it first
 * (IEEE fp32 division) -- the averaging step of the data-parallel gradient all-reduce (sum over ranks / world size) folded into the
 * update instead of a separate pass over the 19.75 MB buffer; 1.0 = gradients as they are.  In place on param / exp_avg / exp_avg_sq. */
int mdt_adam_flat(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, double lr, double beta1, double beta2,
                  double eps, double weight_decay, long long step, double grad_div, void *stream);

/* Per-segment form (round 5): torch.optim.Adam's per-PARAMETER semantics over the same flat buffers (exec.py:39,74).
 *   seg_off [nseg + 1] i64 (device): element offsets of the parameters inside the flat buffers, ascending, seg_off[0] = 0, seg_off[nseg] = n;
 *   seg_step [nseg] i32 (device, in/out): each parameter's own step counter (torch's state[p]['step']); a segment that is updated counts + 1 and its
 *            bias corrections 1 - beta^t use ITS t (formed in double on the device);
 *   present [nseg] u8 or NULL: 0 = the parameter has no gradient in this step as the HOST knows it (p.grad is None);
 *   cond_id [nseg] i32 or NULL, cond [*] f32 or NULL: a segment with cond_id >= 0 has a gradient only if cond[cond_id] > 0 -- a value the training
 *            step wrote on the device (number of positive RoIs / anchors ...): the reference's loss helpers return CONSTANTS when a step has no
 *            positive sample (models/mrcnn.py:233-234, 266-268, 287-288), so torch.optim.Adam leaves those heads alone in such a step;
 *   policy 0: a segment without a gradient is SKIPPED entirely, like torch.optim.Adam with zero_grad(set_to_none=True) (torch >= 2 default);
 *   policy 1: torch 0.4.1 (the reference's pinned version, requirements.txt:25): zero_grad() leaves ZERO tensors behind, so after its first gradient a
 *            parameter is updated in every step (with g = 0 when it has none);
 *   arith 0: every operation rounded separately; 1: the fma pattern of torch's foreach kernels (bit-equal to torch.optim.Adam on this stack);
 *   workspace: mdt_adam_flat_segments_workspace_bytes(nseg), 16-byte aligned.  Two launches (per-segment scalars, update); in place. */
size_t mdt_adam_flat_segments_workspace_bytes(int nseg);
int mdt_adam_flat_segments(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, const long long *seg_off, int nseg,
                           int *seg_step, const unsigned char *present, const int *cond_id, const float *cond, int policy, int arith, double lr,
                           double beta1, double beta2, double eps, double weight_decay, double grad_div, void *workspace, size_t workspace_bytes,
                           void *stream);

/* Forward of the same layer family on the window trick (round 6, csrc/conv_s221.hip): y [B, Y/2, X/2, Z, c_out] (channels-last storage of
 * [B, c_out, Y/2, X/2, Z]) = conv3d(x, w, stride (2, 2, 1), pad k / 2) (+ bias)(ReLU) for x [B, Y, X, Z, c_in] channels-last; wt = the filter as
 * [ky][kx][kz][ci][co] (w.permute(2, 3, 4, 1, 0) contiguous).  fp32 MFMA, deterministic.  Supported: odd k, even c_in, k * c_in <= 128, c_out <= 32,
 * Z % 64 == 0, (X / 2) % 4 == 0; otherwise MDT_ERR_UNSUPPORTED (the caller poses the space-to-depth problem to MIOpen).  What cuDNN computes for
 * models/backbone.py:84 (C1 of the Retina U-Net). */
int mdt_conv_s221_input_grad_supported(int Y, int X, int Z, int c_in, int c_out, int k);
/* the layer's input gradient on the same machine: gx [B, Y, X, Z, c_in] channels-last from gy [B, Y/2, X/2, Z, c_out] channels-last; wd = the filter as
 * [ky][kx][K-1-kz][co][ci] (w.flip(4).permute(2, 3, 4, 0, 1) contiguous); every element of gx is written.  Supported: odd k, even c_out, k * c_out <= 128,
 * c_in <= 32, Z % 64 == 0, X % 4 == 0. */
int mdt_conv_s221_input_grad(const float *gy, const float *wd, float *gx, int batch, int Y, int X, int Z, int c_in, int c_out, int k, void *stream);
/* the forward kernel at UNIT stride: y [B, Y, X, Z, c_out] = conv3d(x, w, stride 1, pad k / 2) (+ bias)(ReLU) for channels-last x, wt as above -- what cuDNN computes for
 * the size-preserving few-channel 3x3x3 layers of NDConvGenerator (utils/model_utils.py:751-765; the 18 -> 18 ResBlock.conv2 of stage C2, models/backbone.py:186-190), and,
 * on the flipped / transposed filter, for their input gradient.  Same conditions with the full-resolution extents (Z % 64 == 0, X % 4 == 0). */
/* ... and the weight gradient of those layers: mdt_conv_s221_wgrad's kernel at unit stride; gw in [c_out][ky][kx][kz][ci] memory order (channels_last_3d
 * storage of the [c_out, c_in, k, k, k] gradient).  k * c_in <= 128, c_out <= 32, Z % 16 == 0. */
int mdt_conv_win_wgrad_supported(int batch, int y, int x, int z, int c_in, int c_out, int k);
size_t mdt_conv_win_wgrad_workspace_bytes(int batch, int y, int x, int z, int c_in, int c_out, int k);
int mdt_conv_win_wgrad(const float *grad_out, const float *x, float *grad_weight, int batch, int y, int x_, int z, int c_in, int c_out, int k,
                       void *workspace, size_t workspace_bytes, void *stream);
int mdt_conv_win_forward_supported(int Y, int X, int Z, int c_in, int c_out, int k);
int mdt_conv_win_forward(const float *x, const float *wt, const float *bias, int relu, float *y, int batch, int Y, int X, int Z, int c_in, int c_out, int k,
                         void *stream);
int mdt_conv_s221_forward_supported(int Y, int X, int Z, int c_in, int c_out, int k);
int mdt_conv_s221_forward(const float *x, const float *wt, const float *bias, int relu, float *y, int batch, int Y, int X, int Z, int c_in, int c_out, int k,
                          void *stream);

/* ------------------------------------------------------------------------- */
/* Loss / target glue of the training step (csrc/glue.hip, round 6)           */
/* ------------------------------------------------------------------------- */
/* Each entry replaces a chain of small tensor operations of the reference's host code with one launch (two for mdt_rpn_sample), operation for
 * operation in the same precision.  Selections break ties towards the lower index.  Replaces, at the call sites of models/mrcnn.py:
 *   mdt_roi_levels            the level rule of pyramid_roi_align (reference models/mrcnn.py:394-409): rois [n, 2 dim + 1] (normalised box,
 *                             batch index) -> boxes [n, 2 dim], batch_ix [n] (truncated), level [n] = clamp(round(4 + log2(sqrt(h w))), lo, hi) - lo
 *                             (five_levels: h w > 0.65 -> level 5)
 *   mdt_rpn_sample            compute_rpn_class_loss's sampling (reference :176-214 + utils/model_utils.py:566-571, 674-691): per batch element the
 *                             n_pos_max best of {rand_pos[a] : match[a] > 0}, the SHEM pool = kpool best of {max fg soft-max prob : match[a] == -1},
 *                             of whose first poolsize * max(pos_count, 1) entries the n_pos_max best by rand_pool[rank] are drawn; nvalid[j] =
 *                             drawn && j < max(pos_count, 1); tgt_pos = max(match[pidx], 0).  MDT_ERR_UNSUPPORTED outside
 *                             mdt_rpn_sample_supported (n_pos_max, kpool <= 128, A x k within the merge block's LDS)
 *   mdt_anchor_delta_targets  utils/model_utils.py:575-617 on gathered rows, fp64 arithmetic, fp32 result [B, n, 2 dim]
 *   mdt_detection_targets     detection_target_layer (reference :461-613) up to the mask crop: one block per batch element; outputs in the slot
 *                             layout [P positives | Nn negatives] per element; pos_rois / box_ids feed the GT-mask RoIAlign. */
int mdt_roi_levels(const float *rois, int n, int dim, int level_lo, int level_hi, int five_levels,
                   float *boxes, int *batch_ix, int *level, void *stream);
int mdt_rpn_sample_supported(int A, int n_pos_max, int kpool);
size_t mdt_rpn_sample_workspace_bytes(int B, int A, int n_pos_max, int kpool);
int mdt_rpn_sample(const int *match, const float *logits, int K, const float *rand_pos, const float *rand_pool,
                   int B, int A, int n_pos_max, int poolsize, int kpool,
                   long long *pidx, unsigned char *pvalid, long long *nidx, unsigned char *nvalid, long long *pos_count, long long *tgt_pos,
                   void *workspace, size_t workspace_bytes, void *stream);
int mdt_anchor_delta_targets(const double *anchors, const double *gt_boxes, const int *argmax, const long long *pidx, const unsigned char *pvalid,
                             const double *std_dev, int B, int A, int G, int n, int dim, float *out, void *stream);
int mdt_detection_targets_supported(int pc, int G, int P, int pool_max, int Nn);
int mdt_detection_targets(const float *rois, int roi_stride, const float *scores, int n_classes, const double *gt_px, const float *scale,
                          const long long *gt_cls, const unsigned char *gt_valid, const int *gt_gidx, const float *rand_pos, const float *rand_pool,
                          const float *std_dev, int B, int pc, int G, int dim, int P, int pool_max, int Nn, int poolsize,
                          float pos_thr, float neg_thr, float ratio_r,
                          long long *sample_indices, unsigned char *valid, unsigned char *is_pos, long long *target_class_ids, float *target_deltas,
                          float *pos_rois, int *box_ids, long long *counts, void *stream);

/* rpn_at_anchors (models/mrcnn.py; the reference's RPN, models/mrcnn.py:40-86, evaluated at the sampled anchors only): idx[s] = anchor index in the
 * order of the concatenated pyramid levels ((y, x[, z], anchor) row-major per level), sample s belongs to batch element s / n_per_element.
 * gather: patches [S, 3^dim, C] = the voxel's neighbourhood on its own level (channels-last maps [B, Y, X, (Z), C]; zero outside the map),
 * k_anchor[s] = idx % anchors_per_voxel.  scatter_add: the adjoint -- adds grad_patches into the gradient maps (float atomics; the caller
 * zero-fills them or passes maps that already hold another gradient; row_major = 1: the maps are [B, C, Y, X, (Z)] row-major, the layout the RoIAlign
 * backward writes, so that both can land in one buffer).  C % 4 == 0, 16-byte aligned maps. */
int mdt_rpn_patch_gather(int n_levels, const float *const *maps_cl, const int *Y, const int *X, const int *Z, int dim, int channels, int anchors_per_voxel,
                         const long long *idx, int n_samples, int n_per_element, float *patches, long long *k_anchor, void *stream);
/* move_add: the deterministic way into maps that already hold a gradient -- a scatter_add (row_major = 0) first sums the rows into ZEROED channels-last
 * side maps (exact for up to two rows per voxel, as a scatter into a fresh tensor always was), then this launch takes every touched voxel's sum out of the
 * side maps by exchange with 0 (the first taker gets the sum, later ones 0; the side maps are all-zero again afterwards and can be kept for the next step)
 * and adds it to the row-major maps: base + sum exactly once, independent of the launch order of the rows. */
int mdt_rpn_patch_move_add(int n_levels, float *const *side_maps_cl, float *const *grad_maps_row_major, const int *Y, const int *X, const int *Z, int dim, int channels,
                           int anchors_per_voxel, const long long *idx, int n_samples, int n_per_element, void *stream);
/* scatter_add_ordered: the DETERMINISTIC adjoint (what the model uses): every voxel has one writer -- the first row that lands on it adds the later rows with the
 * same voxel in row order and read-modify-writes the voxel once; no atomics, works on zeroed maps and on maps that already hold a gradient, either layout.
 * ids_workspace: n_samples * 3^dim int64 (device). */
int mdt_rpn_patch_scatter_add_ordered(int n_levels, float *const *grad_maps, int row_major, const int *Y, const int *X, const int *Z, int dim, int channels,
                                      int anchors_per_voxel, const long long *idx, int n_samples, int n_per_element, const float *grad_patches,
                                      long long *ids_workspace, void *stream);
int mdt_rpn_patch_scatter_add(int n_levels, float *const *grad_maps, int row_major, const int *Y, const int *X, const int *Z, int dim, int channels, int anchors_per_voxel,
                              const long long *idx, int n_samples, int n_per_element, const float *grad_patches, void *stream);

/* refine_detections (reference models/mrcnn.py:620-714) around the batched NMS.  pre: dets [B * fg, pc, 2 dim + 1] -- per (element, foreground
 * class) the boxes decoded with that class's deltas x std_dev (utils/model_utils.py:318-370), x scale, clipped to `window`, rounded, sorted by
 * score (stable, descending; scores < min_confidence carry key -1) -- the input of mdt_nms_{2,3}d_batched.  post: per element the M best
 * survivors (keep [B * fg, pc] as the batched NMS wrote it) over its classes -> result [B * M, 2 dim + 3] = (box, batch index, class id, score),
 * zero rows + valid = 0 for empty slots; when nothing in the whole batch is valid, row 0 = roi 0 of element 0 with class 1 (reference :708-709).
 * std_dev / scale / window: HOST arrays of 2 dim floats.  any_valid_scratch: B device ints. */
int mdt_refine_detections_supported(int pc, int n_classes, int M);
int mdt_refine_detections_pre(const float *rois, const float *probs, const float *deltas, const float *std_dev_host, const float *scale_host,
                              const float *window_host, float min_confidence, int B, int pc, int dim, int n_classes, float *dets, void *stream);
int mdt_refine_detections_post(const float *rois, const float *probs, const float *deltas, const float *std_dev_host, const float *scale_host,
                               const float *window_host, float min_confidence, int B, int pc, int dim, int n_classes, int M,
                               const float *dets, const long long *keep, float *result, unsigned char *valid, int *any_valid_scratch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MDT_HIP_H */
