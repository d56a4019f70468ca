/*
 * mdt_oracle.c -- CPU oracle for the medicaldetectiontoolkit native hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker (or as the timed CPU baseline),
 * never as the thing shipped.  The product path (libmdt_hip.so) never links it.
 *
 * It restates, in plain C99, the arithmetic of the reference's CUDA kernels
 * (paths relative to the reference checkout):
 *   RoIAlign 3D fwd  cuda_functions/roi_align_3D/roi_align/src/cuda/crop_and_resize_kernel.cu:12-151
 *   RoIAlign 3D bwd  ... crop_and_resize_kernel.cu:154-304
 *   output init      cuda_functions/roi_align_3D/roi_align/src/crop_and_resize_gpu.c:26-27,61
 *   RoIAlign 2D      cuda_functions/roi_align_2D/roi_align/src/cuda/crop_and_resize_kernel.cu:11-99,102-194
 *   NMS IoU          cuda_functions/nms_3D/src/cuda/nms_kernel.cu:16-28 (2D: nms_2D/.../nms_kernel.cu:16-24)
 *   NMS mask         cuda_functions/nms_3D/src/cuda/nms_kernel.cu:30-78
 *   NMS greedy scan  cuda_functions/nms_3D/src/nms_cuda.c:47-61
 *   CPU NMS (>=)     cuda_functions/nms_3D/src/nms.c:35-70, cuda_functions/nms_2D/src/nms.c:35-65
 *
 * Parity pinning: the reference ships no golden vectors or tests for these ops
 * (SURVEY.md section 4).  This restatement is pinned instead against
 *   (a) the reference's own nms.c compiled where it lies (oracle/_ref/, see Makefile),
 *   (b) the reference's CUDA kernels compiled for gfx950 through a macro shim
 *       (oracle/_ref/libref_gpu_*.so, GPU box only),
 *   (c) torch.nn.functional.grid_sample (align_corners=False, border padding),
 *       which is the same sampling rule for pool extents > 1 (tests/).
 *
 * Build with:  gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math  (see Makefile).
 * fp32 throughout, no FMA contraction, operation order as in the CUDA source.
 * The bwd accumulates in out_idx order (the CUDA atomics have no defined order).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* sampling coordinate of bin p on one axis                            */
/* crop_and_resize_kernel.cu:51-75 (y), :79-91 (x), :94-106 (z)        */
/*   scale  = (P > 1) ? (a2 - a1) * L / P : 0            (float)       */
/*   in     = (P > 1) ? a1*L + p*scale + scale/2 - 0.5               */
/*                    : 0.5 * (a1 + a2) * L                            */
/* the 0.5 literals are double, so the final -0.5 (resp. the whole     */
/* P == 1 product) is evaluated in double and rounded to float once.   */
/* then clamp to [0, L-1].                                             */
/* ------------------------------------------------------------------ */
static float sample_coord(float a1, float a2, int L, int P, int p)
{
    const float scale = (P > 1) ? (a2 - a1) * (L) / (P) : 0;
    float in = (P > 1) ? a1 * (L) + p * scale + scale / 2 - 0.5
                       : 0.5 * (a1 + a2) * (L);
    if (in > L - 1) in = L - 1;
    if (in < 0) in = 0;
    return in;
}

/* exported so tests can pin the coordinate rule on its own */
float oracle_sample_coord(float a1, float a2, int L, int P, int p)
{
    return sample_coord(a1, a2, L, P, p);
}

/* ------------------------------------------------------------------ */
/* RoIAlign 3D forward                                                 */
/* image [B,C,H,W,D] (z contiguous), boxes [N,6]=(y1,x1,y2,x2,z1,z2)   */
/* normalised, box_ind [N] int32, crops [N,C,ch,cw,cd].                */
/* crops is zero-filled first (crop_and_resize_gpu.c:26-27); boxes with */
/* box_ind outside [0,B) are skipped (kernel.cu:43-47).                */
/* extrapolation_value is accepted and unused, as in the reference.    */
/* ------------------------------------------------------------------ */
void oracle_crop_and_resize_3d_forward(
    const float *image, const float *boxes, const int *box_ind,
    int num_boxes, int batch, int H, int W, int D,
    int ch, int cw, int cd, int depth, float extrapolation_value, float *crops)
{
    (void)extrapolation_value;
    const int64_t total = (int64_t)num_boxes * depth * ch * cw * cd;
    memset(crops, 0, (size_t)total * sizeof(float));
    /* every output element is independent: parallel over the flat index, like the reference's CPU path
       (crop_and_resize.c:30, `#pragma omp parallel for` over boxes); results are bit-identical to the serial loop */
#pragma omp parallel for schedule(static)
    for (int64_t out_idx = 0; out_idx < total; ++out_idx) {
        int64_t idx = out_idx;
        const int z = (int)(idx % cd); idx /= cd;
        const int x = (int)(idx % cw); idx /= cw;
        const int y = (int)(idx % ch); idx /= ch;
        const int d = (int)(idx % depth);
        const int b = (int)(idx / depth);

        const float y1 = boxes[b * 6], x1 = boxes[b * 6 + 1];
        const float y2 = boxes[b * 6 + 2], x2 = boxes[b * 6 + 3];
        const float z1 = boxes[b * 6 + 4], z2 = boxes[b * 6 + 5];
        const int b_in = box_ind[b];
        if (b_in < 0 || b_in >= batch) continue;

        const float in_y = sample_coord(y1, y2, H, ch, y);
        const float in_x = sample_coord(x1, x2, W, cw, x);
        const float in_z = sample_coord(z1, z2, D, cd, z);

        const int top = (int)floorf(in_y), bottom = (int)ceilf(in_y);
        const float y_lerp = in_y - top;
        const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
        const float x_lerp = in_x - left;
        const int front = (int)floorf(in_z), back = (int)ceilf(in_z);
        const float z_lerp = in_z - front;

        const float *pimage = image + ((int64_t)b_in * depth + d) * H * W * D;
        const float tlf = pimage[front + (int64_t)D * (left + W * top)];
        const float trf = pimage[front + (int64_t)D * (right + W * top)];
        const float blf = pimage[front + (int64_t)D * (left + W * bottom)];
        const float brf = pimage[front + (int64_t)D * (right + W * bottom)];
        const float tlb = pimage[back + (int64_t)D * (left + W * top)];
        const float trb = pimage[back + (int64_t)D * (right + W * top)];
        const float blb = pimage[back + (int64_t)D * (left + W * bottom)];
        const float brb = pimage[back + (int64_t)D * (right + W * bottom)];

        const float top_front = tlf + (trf - tlf) * x_lerp;
        const float bottom_front = blf + (brf - blf) * x_lerp;
        const float top_back = tlb + (trb - tlb) * x_lerp;
        const float bottom_back = blb + (brb - blb) * x_lerp;
        const float frontv = top_front + (bottom_front - top_front) * y_lerp;
        const float backv = top_back + (bottom_back - top_back) * y_lerp;
        crops[out_idx] = frontv + (backv - frontv) * z_lerp;
    }
}

/* ------------------------------------------------------------------ */
/* RoIAlign 3D backward: grads [N,C,ch,cw,cd] -> grads_image [B,C,H,W,D] */
/* zero-fill (crop_and_resize_gpu.c:61) then 8 adds per element in the  */
/* order of kernel.cu:256-301; weight product order (wx*wz)*wy*g.       */
/* ------------------------------------------------------------------ */
void oracle_crop_and_resize_3d_backward(
    const float *grads, const float *boxes, const int *box_ind,
    int num_boxes, int batch, int H, int W, int D,
    int ch, int cw, int cd, int depth, float *grads_image)
{
    memset(grads_image, 0, (size_t)batch * depth * H * W * D * sizeof(float));
    const int64_t total = (int64_t)num_boxes * depth * ch * cw * cd;
    for (int64_t out_idx = 0; out_idx < total; ++out_idx) {
        int64_t idx = out_idx;
        const int z = (int)(idx % cd); idx /= cd;
        const int x = (int)(idx % cw); idx /= cw;
        const int y = (int)(idx % ch); idx /= ch;
        const int d = (int)(idx % depth);
        const int b = (int)(idx / depth);

        const float y1 = boxes[b * 6], x1 = boxes[b * 6 + 1];
        const float y2 = boxes[b * 6 + 2], x2 = boxes[b * 6 + 3];
        const float z1 = boxes[b * 6 + 4], z2 = boxes[b * 6 + 5];
        const int b_in = box_ind[b];
        if (b_in < 0 || b_in >= batch) continue;

        const float in_y = sample_coord(y1, y2, H, ch, y);
        const float in_x = sample_coord(x1, x2, W, cw, x);
        const float in_z = sample_coord(z1, z2, D, cd, z);

        const int top = (int)floorf(in_y), bottom = (int)ceilf(in_y);
        const float y_lerp = in_y - top;
        const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
        const float x_lerp = in_x - left;
        const int front = (int)floorf(in_z), back = (int)ceilf(in_z);
        const float z_lerp = in_z - front;

        float *pimage = grads_image + ((int64_t)b_in * depth + d) * H * W * D;
        const float g = grads[out_idx];
        pimage[front + (int64_t)D * (left + W * top)]     += (1 - x_lerp) * (1 - z_lerp) * (1 - y_lerp) * g;
        pimage[back + (int64_t)D * (left + W * top)]      += (1 - x_lerp) * (z_lerp) * (1 - y_lerp) * g;
        pimage[front + (int64_t)D * (right + W * top)]    += (x_lerp) * (1 - z_lerp) * (1 - y_lerp) * g;
        pimage[back + (int64_t)D * (right + W * top)]     += (x_lerp) * (z_lerp) * (1 - y_lerp) * g;
        pimage[front + (int64_t)D * (left + W * bottom)]  += (1 - x_lerp) * (1 - z_lerp) * (y_lerp) * g;
        pimage[back + (int64_t)D * (left + W * bottom)]   += (1 - x_lerp) * (z_lerp) * (y_lerp) * g;
        pimage[front + (int64_t)D * (right + W * bottom)] += (x_lerp) * (1 - z_lerp) * (y_lerp) * g;
        pimage[back + (int64_t)D * (right + W * bottom)]  += (x_lerp) * (z_lerp) * (y_lerp) * g;
    }
}

/* ------------------------------------------------------------------ */
/* RoIAlign 2D forward / backward                                      */
/* roi_align_2D/.../crop_and_resize_kernel.cu:17-97, 108-192           */
/* image [B,C,H,W], boxes [N,4]=(y1,x1,y2,x2), crops [N,C,ch,cw]        */
/* ------------------------------------------------------------------ */
void oracle_crop_and_resize_2d_forward(
    const float *image, const float *boxes, const int *box_ind,
    int num_boxes, int batch, int H, int W,
    int ch, int cw, int depth, float extrapolation_value, float *crops)
{
    (void)extrapolation_value;
    const int64_t total = (int64_t)num_boxes * depth * ch * cw;
    memset(crops, 0, (size_t)total * sizeof(float));
    /* every output element is independent: parallel over the flat index, like the reference's CPU path
       (crop_and_resize.c:30, `#pragma omp parallel for` over boxes); results are bit-identical to the serial loop */
#pragma omp parallel for schedule(static)
    for (int64_t out_idx = 0; out_idx < total; ++out_idx) {
        int64_t idx = out_idx;
        const int x = (int)(idx % cw); idx /= cw;
        const int y = (int)(idx % ch); idx /= ch;
        const int d = (int)(idx % depth);
        const int b = (int)(idx / depth);

        const float y1 = boxes[b * 4], x1 = boxes[b * 4 + 1];
        const float y2 = boxes[b * 4 + 2], x2 = boxes[b * 4 + 3];
        const int b_in = box_ind[b];
        if (b_in < 0 || b_in >= batch) continue;

        const float in_y = sample_coord(y1, y2, H, ch, y);
        const float in_x = sample_coord(x1, x2, W, cw, x);
        const int top = (int)floorf(in_y), bottom = (int)ceilf(in_y);
        const float y_lerp = in_y - top;
        const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
        const float x_lerp = in_x - left;

        const float *pimage = image + ((int64_t)b_in * depth + d) * H * W;
        const float tl = pimage[top * W + left];
        const float tr = pimage[top * W + right];
        const float bl = pimage[bottom * W + left];
        const float br = pimage[bottom * W + right];
        const float topv = tl + (tr - tl) * x_lerp;
        const float bottomv = bl + (br - bl) * x_lerp;
        crops[out_idx] = topv + (bottomv - topv) * y_lerp;
    }
}

void oracle_crop_and_resize_2d_backward(
    const float *grads, const float *boxes, const int *box_ind,
    int num_boxes, int batch, int H, int W,
    int ch, int cw, int depth, float *grads_image)
{
    memset(grads_image, 0, (size_t)batch * depth * H * W * sizeof(float));
    const int64_t total = (int64_t)num_boxes * depth * ch * cw;
    for (int64_t out_idx = 0; out_idx < total; ++out_idx) {
        int64_t idx = out_idx;
        const int x = (int)(idx % cw); idx /= cw;
        const int y = (int)(idx % ch); idx /= ch;
        const int d = (int)(idx % depth);
        const int b = (int)(idx / depth);

        const float y1 = boxes[b * 4], x1 = boxes[b * 4 + 1];
        const float y2 = boxes[b * 4 + 2], x2 = boxes[b * 4 + 3];
        const int b_in = box_ind[b];
        if (b_in < 0 || b_in >= batch) continue;

        const float in_y = sample_coord(y1, y2, H, ch, y);
        const float in_x = sample_coord(x1, x2, W, cw, x);
        const int top = (int)floorf(in_y), bottom = (int)ceilf(in_y);
        const float y_lerp = in_y - top;
        const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
        const float x_lerp = in_x - left;

        float *pimage = grads_image + ((int64_t)b_in * depth + d) * H * W;
        const float g = grads[out_idx];
        /* kernel.cu:175-192: dtop first, then dbottom */
        const float dtop = (1 - y_lerp) * g;
        pimage[top * W + left] += (1 - x_lerp) * dtop;
        pimage[top * W + right] += x_lerp * dtop;
        const float dbottom = y_lerp * g;
        pimage[bottom * W + left] += (1 - x_lerp) * dbottom;
        pimage[bottom * W + right] += x_lerp * dbottom;
    }
}

/* ------------------------------------------------------------------ */
/* NMS                                                                 */
/* ------------------------------------------------------------------ */
/* nms_3D/src/cuda/nms_kernel.cu:16-28; a,b = (c0,c1,c2,c3,c4,c5,score);
 * extents are (c2-c0), (c3-c1), (c5-c4), +1 pixel convention. */
static float iou_3d(const float *a, const float *b)
{
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float front = fmaxf(a[4], b[4]), back = fminf(a[5], b[5]);
    float width = fmaxf(right - left + 1, 0.f);
    float height = fmaxf(bottom - top + 1, 0.f);
    float depth = fmaxf(back - front + 1, 0.f);
    float interS = width * height * depth;
    float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1) * (a[5] - a[4] + 1);
    float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) * (b[5] - b[4] + 1);
    return interS / (Sa + Sb - interS);
}

/* nms_2D/src/cuda/nms_kernel.cu:16-24 */
static float iou_2d(const float *a, const float *b)
{
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float width = fmaxf(right - left + 1, 0.f);
    float height = fmaxf(bottom - top + 1, 0.f);
    float interS = width * height;
    float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
    float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
    return interS / (Sa + Sb - interS);
}

float oracle_iou_3d(const float *a, const float *b) { return iou_3d(a, b); }
float oracle_iou_2d(const float *a, const float *b) { return iou_2d(a, b); }

#define TPB 64
#define DIVUP(m, n) ((m) / (n) + ((m) % (n) > 0))

/* pairwise suppression mask, nms_kernel.cu:30-78.  mask is [n, col_blocks]
 * u64.  upper_only != 0 leaves blocks with row_start > col_start at zero (the
 * reference fills them, but nms_cuda.c:54 never reads them). */
static void nms_mask(const float *dets, int n, int stride, float thresh,
                     int strict_gt, int upper_only, uint64_t *mask)
{
    const int col_blocks = DIVUP(n, TPB);
    memset(mask, 0, (size_t)n * col_blocks * sizeof(uint64_t));
    for (int row_start = 0; row_start < col_blocks; ++row_start) {
        for (int col_start = 0; col_start < col_blocks; ++col_start) {
            if (upper_only && row_start > col_start) continue;
            const int row_size = (n - row_start * TPB < TPB) ? n - row_start * TPB : TPB;
            const int col_size = (n - col_start * TPB < TPB) ? n - col_start * TPB : TPB;
            for (int t = 0; t < row_size; ++t) {
                const int cur = TPB * row_start + t;
                const float *cur_box = dets + (size_t)cur * stride;
                uint64_t bits = 0;
                int start = (row_start == col_start) ? t + 1 : 0;
                for (int i = start; i < col_size; ++i) {
                    const float *other = dets + (size_t)(TPB * col_start + i) * stride;
                    const float v = (stride == 7) ? iou_3d(cur_box, other) : iou_2d(cur_box, other);
                    if (strict_gt ? (v > thresh) : (v >= thresh)) bits |= 1ULL << i;
                }
                mask[(size_t)cur * col_blocks + col_start] = bits;
            }
        }
    }
}

void oracle_nms_mask_3d(const float *dets_sorted, int n, float thresh, uint64_t *mask)
{ nms_mask(dets_sorted, n, 7, thresh, 1, 0, mask); }
void oracle_nms_mask_2d(const float *dets_sorted, int n, float thresh, uint64_t *mask)
{ nms_mask(dets_sorted, n, 5, thresh, 1, 0, mask); }

/* host greedy scan, nms_cuda.c:47-61 */
static int64_t nms_scan(const uint64_t *mask, int n, int64_t *keep)
{
    const int col_blocks = DIVUP(n, TPB);
    uint64_t *remv = (uint64_t *)calloc((size_t)(col_blocks > 0 ? col_blocks : 1), sizeof(uint64_t));
    int64_t num_to_keep = 0;
    for (int i = 0; i < n; ++i) {
        const int nblock = i / TPB, inblock = i % TPB;
        if (!(remv[nblock] & (1ULL << inblock))) {
            keep[num_to_keep++] = i;
            const uint64_t *p = mask + (size_t)i * col_blocks;
            for (int j = nblock; j < col_blocks; ++j) remv[j] |= p[j];
        }
    }
    free(remv);
    return num_to_keep;
}

/* gpu_nms restated: boxes already sorted by descending score (pth_nms.py:10-12);
 * keep = positions in the sorted list, ascending.  strict_gt=1 is the GPU rule
 * (IoU > thresh, nms_kernel.cu:71); strict_gt=0 gives the cpu_nms rule (>=). */
int oracle_gpu_nms_3d(const float *dets_sorted, int n, float thresh, int strict_gt,
                      int64_t *keep, int64_t *num_out)
{
    if (n <= 0) { *num_out = 0; return 0; }
    const int col_blocks = DIVUP(n, TPB);
    uint64_t *mask = (uint64_t *)malloc((size_t)n * col_blocks * sizeof(uint64_t));
    if (!mask) return -1;
    nms_mask(dets_sorted, n, 7, thresh, strict_gt, 1, mask);
    *num_out = nms_scan(mask, n, keep);
    free(mask);
    return 0;
}

int oracle_gpu_nms_2d(const float *dets_sorted, int n, float thresh, int strict_gt,
                      int64_t *keep, int64_t *num_out)
{
    if (n <= 0) { *num_out = 0; return 0; }
    const int col_blocks = DIVUP(n, TPB);
    uint64_t *mask = (uint64_t *)malloc((size_t)n * col_blocks * sizeof(uint64_t));
    if (!mask) return -1;
    nms_mask(dets_sorted, n, 5, thresh, strict_gt, 1, mask);
    *num_out = nms_scan(mask, n, keep);
    free(mask);
    return 0;
}

/* cpu_nms restated (nms.c:35-70): dets in ORIGINAL order, `order` = indices by
 * descending score, `areas` precomputed by the caller (pth_nms.py:29-31) with
 * the +1 convention; suppress when ovr >= thresh; keep holds original indices. */
int oracle_cpu_nms_3d(const float *dets, int64_t n, int64_t boxes_dim,
                      const int64_t *order, const float *areas, float thresh,
                      int64_t *keep, int64_t *num_out)
{
    unsigned char *suppressed = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int64_t num_to_keep = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        const int64_t i = order[_i];
        if (suppressed[i] == 1) continue;
        keep[num_to_keep++] = i;
        const float ix1 = dets[i * boxes_dim], iy1 = dets[i * boxes_dim + 1];
        const float ix2 = dets[i * boxes_dim + 2], iy2 = dets[i * boxes_dim + 3];
        const float iz1 = dets[i * boxes_dim + 4], iz2 = dets[i * boxes_dim + 5];
        const float iarea = areas[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            const int64_t j = order[_j];
            if (suppressed[j] == 1) continue;
            const float xx1 = fmaxf(ix1, dets[j * boxes_dim]);
            const float yy1 = fmaxf(iy1, dets[j * boxes_dim + 1]);
            const float xx2 = fminf(ix2, dets[j * boxes_dim + 2]);
            const float yy2 = fminf(iy2, dets[j * boxes_dim + 3]);
            const float zz1 = fmaxf(iz1, dets[j * boxes_dim + 4]);
            const float zz2 = fminf(iz2, dets[j * boxes_dim + 5]);
            const float w = fmaxf(0.0, xx2 - xx1 + 1);
            const float h = fmaxf(0.0, yy2 - yy1 + 1);
            const float d = fmaxf(0.0, zz2 - zz1 + 1);
            const float inter = w * h * d;
            const float ovr = inter / (iarea + areas[j] - inter);
            if (ovr >= thresh) suppressed[j] = 1;
        }
    }
    *num_out = num_to_keep;
    free(suppressed);
    return 0;
}

int oracle_cpu_nms_2d(const float *dets, int64_t n, int64_t boxes_dim,
                      const int64_t *order, const float *areas, float thresh,
                      int64_t *keep, int64_t *num_out)
{
    unsigned char *suppressed = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int64_t num_to_keep = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        const int64_t i = order[_i];
        if (suppressed[i] == 1) continue;
        keep[num_to_keep++] = i;
        const float ix1 = dets[i * boxes_dim], iy1 = dets[i * boxes_dim + 1];
        const float ix2 = dets[i * boxes_dim + 2], iy2 = dets[i * boxes_dim + 3];
        const float iarea = areas[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            const int64_t j = order[_j];
            if (suppressed[j] == 1) continue;
            const float xx1 = fmaxf(ix1, dets[j * boxes_dim]);
            const float yy1 = fmaxf(iy1, dets[j * boxes_dim + 1]);
            const float xx2 = fminf(ix2, dets[j * boxes_dim + 2]);
            const float yy2 = fminf(iy2, dets[j * boxes_dim + 3]);
            const float w = fmaxf(0.0, xx2 - xx1 + 1);
            const float h = fmaxf(0.0, yy2 - yy1 + 1);
            const float inter = w * h;
            const float ovr = inter / (iarea + areas[j] - inter);
            if (ovr >= thresh) suppressed[j] = 1;
        }
    }
    *num_out = num_to_keep;
    free(suppressed);
    return 0;
}
