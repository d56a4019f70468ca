// anchors.hip -- anchor generation and anchor <-> ground-truth matching for gfx950.
//
// Follows (paths relative to the reference checkout):
//   generate_anchors / generate_anchors_3D   utils/model_utils.py:190-226, 230-272
//   gt_anchor_matching steps 1-3             utils/model_utils.py:505-563
//   compute_overlaps / compute_iou_{2D,3D}   utils/model_utils.py:35-110
// Everything is float64 like the numpy reference, built with -ffp-contract=off,
// and uses only IEEE add/sub/mul/div/compare on the device (the sqrt of the
// anchor ratios is taken on the host), so results equal numpy's bit for bit.
// In the reference the matching runs in numpy on one host core per batch element
// (models/mrcnn.py:894; 0.118 s for 449k anchors); here it is one HBM-bound pass
// over the anchors plus a tiny second-stage reduction.

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "mdt_hip.h"

namespace {

constexpr int MAX_K = 64;  // anchors per position

struct AnchorGenParams {
    double h[MAX_K], w[MAX_K], d[MAX_K];
    int K, Y, X, Z;  // positions per axis (after anchor_stride)
    double stride_xy, stride_z;  // feature_stride * anchor_stride step in input pixels
};

template <int DIM>
__global__ __launch_bounds__(256) void gen_anchors_kernel(AnchorGenParams g, long long n_rows,
                                                          double *__restrict__ out, float *__restrict__ out_f32)
{
    const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const int k = (int)(row % g.K);
    long long m = row / g.K;
    int z = 0;
    if (DIM == 3) { z = (int)(m % g.Z); m /= g.Z; }
    const int x = (int)(m % g.X);
    const int y = (int)(m / g.X);
    // centres: arange(0, shape, anchor_stride) * feature_stride  (integers times stride: exact)
    const double cy = (double)y * g.stride_xy;
    const double cx = (double)x * g.stride_xy;
    const double y1 = cy - 0.5 * g.h[k], y2 = cy + 0.5 * g.h[k];
    const double x1 = cx - 0.5 * g.w[k], x2 = cx + 0.5 * g.w[k];
    double *o = out + row * (2 * DIM);
    o[0] = y1; o[1] = x1; o[2] = y2; o[3] = x2;
    if (DIM == 3) {
        const double cz = (double)z * g.stride_z;
        o[4] = cz - 0.5 * g.d[k];
        o[5] = cz + 0.5 * g.d[k];
    }
    if (out_f32) {
        float *f = out_f32 + row * (2 * DIM);
#pragma unroll
        for (int q = 0; q < 2 * DIM; ++q) f[q] = (float)o[q];
    }
}

// ---- matching -------------------------------------------------------------
constexpr int MATCH_THREADS = 256;
constexpr int MATCH_MAX_BLOCKS = 1024;

template <int DIM>
__device__ __forceinline__ double iou_f64(const double *gt, double gt_vol, const double *a, double a_vol)
{
    // compute_iou_3D (model_utils.py:58-79): box = gt, boxes = anchors
    const double y1 = fmax(gt[0], a[0]);
    const double y2 = fmin(gt[2], a[2]);
    const double x1 = fmax(gt[1], a[1]);
    const double x2 = fmin(gt[3], a[3]);
    double inter = fmax(x2 - x1, 0.0) * fmax(y2 - y1, 0.0);
    if (DIM == 3) {
        const double z1 = fmax(gt[4], a[4]);
        const double z2 = fmin(gt[5], a[5]);
        inter = inter * fmax(z2 - z1, 0.0);
    }
    const double uni = gt_vol + a_vol - inter;
    return inter / uni;
}

template <int DIM>
__device__ __forceinline__ double box_vol(const double *b)
{
    double v = (b[2] - b[0]) * (b[3] - b[1]);
    if (DIM == 3) v = v * (b[5] - b[4]);
    return v;
}

// pass 1: per anchor max / argmax over GT, step-1 and step-3 labels; per block and
// per GT the best (iou, first anchor index) into `part`.
// Batched form (blockIdx.y = batch element): the element's GT rows start at gt + e * gmax * 2 DIM, its count is read from
// n_gt_dev[e] ON THE DEVICE (so the launch does not depend on host-side GT counts: the training step is capturable in a hipGraph
// and needs no per-element launches); n_gt_dev == NULL is the single-problem form with the host count G (= gmax).
template <int DIM>
__global__ __launch_bounds__(MATCH_THREADS) void match_pass1_kernel(
    const double *__restrict__ anchors, int A, const double *__restrict__ gt, const int *__restrict__ gt_cls,
    int G, const int *__restrict__ n_gt_dev, int gmax, double neg_thresh, double pos_thresh,
    int *__restrict__ matches, int *__restrict__ iou_argmax, double *__restrict__ iou_max,
    double *__restrict__ part_val, int *__restrict__ part_idx)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *s_gt = reinterpret_cast<double *>(smem_raw);           // [gmax][2*DIM]
    double *s_vol = s_gt + (size_t)gmax * 2 * DIM;                  // [gmax]
    double *s_red_v = s_vol + gmax;                                 // [MATCH_THREADS/64]
    int *s_red_i = reinterpret_cast<int *>(s_red_v + MATCH_THREADS / 64);
    {
        const size_t e = blockIdx.y;
        gt += e * (size_t)gmax * 2 * DIM;
        if (gt_cls) gt_cls += e * (size_t)gmax;
        matches += e * (size_t)A;
        iou_argmax += e * (size_t)A;
        iou_max += e * (size_t)A;
        part_val += e * (size_t)gridDim.x * gmax;
        part_idx += e * (size_t)gridDim.x * gmax;
        if (n_gt_dev) G = min(max(n_gt_dev[e], 0), gmax);
    }
    if (G == 0) {   // "gt_boxes is None": every anchor negative (model_utils.py:524-526)
        const int per = (A + gridDim.x - 1) / gridDim.x;
        const int b0 = blockIdx.x * per, b1 = min(A, b0 + per);
        for (int a = b0 + threadIdx.x; a < b1; a += MATCH_THREADS) { matches[a] = -1; iou_argmax[a] = 0; iou_max[a] = 0.0; }
        return;
    }

    for (int t = threadIdx.x; t < G * 2 * DIM; t += MATCH_THREADS) s_gt[t] = gt[t];
    __syncthreads();
    for (int t = threadIdx.x; t < G; t += MATCH_THREADS) s_vol[t] = box_vol<DIM>(s_gt + (size_t)t * 2 * DIM);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_block = (A + gridDim.x - 1) / gridDim.x;
    const int a0 = blockIdx.x * per_block;
    const int a1 = min(A, a0 + per_block);

    // each thread keeps, for the anchors it visits, its own per-GT best only for the
    // GT currently being reduced -> loop GTs outermost would re-read anchors G times.
    // Instead: first the per-anchor pass (stores iou_max/argmax), GT-best handled in a
    // second sweep over the block's anchors per GT using the cached anchor data in L2.
    for (int a = a0 + threadIdx.x; a < a1; a += MATCH_THREADS) {
        double ab[2 * DIM];
#pragma unroll
        for (int q = 0; q < 2 * DIM; ++q) ab[q] = anchors[(long long)a * 2 * DIM + q];
        const double av = box_vol<DIM>(ab);
        double best = 0.0;
        int best_g = 0;
        for (int g = 0; g < G; ++g) {
            const double v = iou_f64<DIM>(s_gt + (size_t)g * 2 * DIM, s_vol[g], ab, av);
            if (g == 0 || v > best) { best = v; best_g = g; }  // np.argmax: first maximum
        }
        int label = 0;
        if (best < neg_thresh) label = -1;                       // model_utils.py:549-552
        if (best >= pos_thresh) label = gt_cls ? gt_cls[best_g] : 1;  // :562-563
        matches[a] = label;
        iou_argmax[a] = best_g;
        iou_max[a] = best;
    }

    for (int g = 0; g < G; ++g) {
        double bv = -1.0;
        int bi = 0x7fffffff;
        for (int a = a0 + threadIdx.x; a < a1; a += MATCH_THREADS) {
            double ab[2 * DIM];
#pragma unroll
            for (int q = 0; q < 2 * DIM; ++q) ab[q] = anchors[(long long)a * 2 * DIM + q];
            const double v = iou_f64<DIM>(s_gt + (size_t)g * 2 * DIM, s_vol[g], ab, box_vol<DIM>(ab));
            if (v > bv) { bv = v; bi = a; }  // ascending a per thread: first maximum kept
        }
        // wave reduce: larger value wins, ties -> smaller index
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_down(bv, off);
            const int oi = __shfl_down(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_red_v[wave] = bv; s_red_i[wave] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < MATCH_THREADS / 64; ++w)
                if (s_red_v[w] > bv || (s_red_v[w] == bv && s_red_i[w] < bi)) { bv = s_red_v[w]; bi = s_red_i[w]; }
            part_val[(size_t)blockIdx.x * G + g] = bv;
            part_idx[(size_t)blockIdx.x * G + g] = bi;
        }
        __syncthreads();
    }
}

// pass 2 (one block): reduce partials per GT (all threads), then apply step 2 (model_utils.py:556-559)
// in GT order, not overriding step-3 positives (:562-563 runs after step 2).
__global__ __launch_bounds__(256) void match_pass2_kernel(
    const double *__restrict__ part_val, const int *__restrict__ part_idx, int n_blocks, int G, const int *__restrict__ n_gt_dev, int gmax,
    int A, const int *__restrict__ gt_cls, double pos_thresh, const double *__restrict__ iou_max,
    int *__restrict__ matches, int *__restrict__ gt_best)
{
    __shared__ double s_v[4];
    __shared__ int s_i[4];
    {
        const size_t e = blockIdx.x;      // one block per batch element
        part_val += e * (size_t)n_blocks * gmax;
        part_idx += e * (size_t)n_blocks * gmax;
        if (gt_cls) gt_cls += e * (size_t)gmax;
        iou_max += e * (size_t)A;
        matches += e * (size_t)A;
        gt_best += e * (size_t)gmax;
        if (n_gt_dev) G = min(max(n_gt_dev[e], 0), gmax);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int g = 0; g < G; ++g) {
        double bv = -1.0;
        int bi = 0x7fffffff;
        for (int b = threadIdx.x; b < n_blocks; b += blockDim.x) {
            const double v = part_val[(size_t)b * G + g];
            const int i = part_idx[(size_t)b * G + g];
            if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_down(bv, off);
            const int oi = __shfl_down(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w)
                if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) { bv = s_v[w]; bi = s_i[w]; }
            gt_best[g] = bi;
            // step 2 (model_utils.py:556-559) in GT order; step-3 positives (:562-563) are not overridden
            if (bi != 0x7fffffff && !(iou_max[bi] >= pos_thresh)) matches[bi] = gt_cls ? gt_cls[g] : 1;
        }
        __syncthreads();
    }
}

inline int check_launch()
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

inline int match_blocks(int A)
{
    int b = (A + MATCH_THREADS * 4 - 1) / (MATCH_THREADS * 4);
    if (b > MATCH_MAX_BLOCKS) b = MATCH_MAX_BLOCKS;
    if (b < 1) b = 1;
    return b;
}

}  // namespace

extern "C" {

int mdt_generate_anchors(int dim, const double *scales_xy_host, const double *scales_z_host, int n_scales,
                         const double *ratios_host, int n_ratios, const int *shape_host,
                         double feature_stride_xy, double feature_stride_z, int anchor_stride,
                         double *out, float *out_f32, void *stream)
{
    (void)hipGetLastError();   // drop stale error state of earlier runtime calls on this thread
    if ((dim != 2 && dim != 3) || n_scales <= 0 || n_ratios <= 0 || anchor_stride <= 0 || !scales_xy_host ||
        !ratios_host || !shape_host || (dim == 3 && !scales_z_host))
        return MDT_ERR_INVALID_ARGUMENT;
    const int K = n_scales * n_ratios;
    if (K > MAX_K) return MDT_ERR_UNSUPPORTED;
    AnchorGenParams g;
    g.K = K;
    for (int k = 0; k < K; ++k) {
        const double s = scales_xy_host[k % n_scales];
        const double r = ratios_host[k / n_scales];
        g.h[k] = s / sqrt(r);   // model_utils.py:247 (2D :207)
        g.w[k] = s * sqrt(r);   // :248 (:208)
        g.d[k] = dim == 3 ? scales_z_host[k % n_scales] : 0.0;  // np.tile, :249
    }
    g.Y = (shape_host[0] + anchor_stride - 1) / anchor_stride;
    g.X = (shape_host[1] + anchor_stride - 1) / anchor_stride;
    g.Z = dim == 3 ? (shape_host[2] + anchor_stride - 1) / anchor_stride : 1;
    g.stride_xy = (double)anchor_stride * feature_stride_xy;
    g.stride_z = (double)anchor_stride * feature_stride_z;
    const long long n_rows = (long long)g.Y * g.X * g.Z * K;
    if (n_rows == 0) return MDT_OK;
    const long long blocks = (n_rows + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    if (dim == 3) hipLaunchKernelGGL(gen_anchors_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, s, g, n_rows, out, out_f32);
    else hipLaunchKernelGGL(gen_anchors_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, g, n_rows, out, out_f32);
    return check_launch();
}

size_t mdt_anchor_match_workspace_bytes(int n_anchors, int n_gt)
{
    if (n_anchors <= 0 || n_gt <= 0) return 16;
    const size_t nb = (size_t)match_blocks(n_anchors);
    size_t bytes = nb * n_gt * sizeof(double);            // part_val
    bytes += nb * n_gt * sizeof(int);                      // part_idx
    bytes = (bytes + 15) & ~(size_t)15;
    bytes += (size_t)n_anchors * sizeof(double);           // iou_max when the caller passes NULL
    return (bytes + 255) & ~(size_t)255;
}

int mdt_anchor_match(const double *anchors, int n_anchors, int dim,
                     const double *gt_boxes, const int *gt_class_ids, int n_gt,
                     double neg_thresh, double pos_thresh,
                     int *matches, int *iou_argmax, double *iou_max, int *gt_best_anchor,
                     void *workspace, size_t workspace_bytes, void *stream)
{
    (void)hipGetLastError();   // drop stale error state of earlier runtime calls on this thread
    if (n_anchors < 0 || n_gt < 0 || (dim != 2 && dim != 3) || !matches || !iou_argmax)
        return MDT_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    if (n_anchors == 0) return MDT_OK;
    if (n_gt == 0) {  // "gt_boxes is None": every anchor negative (model_utils.py:524-526)
        if (hipMemsetAsync(matches, 0xff, sizeof(int) * (size_t)n_anchors, s) != hipSuccess) return MDT_ERR_LAUNCH_FAILED;
        if (hipMemsetAsync(iou_argmax, 0, sizeof(int) * (size_t)n_anchors, s) != hipSuccess) return MDT_ERR_LAUNCH_FAILED;
        if (iou_max && hipMemsetAsync(iou_max, 0, sizeof(double) * (size_t)n_anchors, s) != hipSuccess) return MDT_ERR_LAUNCH_FAILED;
        return MDT_OK;
    }
    if (!gt_best_anchor) return MDT_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < mdt_anchor_match_workspace_bytes(n_anchors, n_gt))
        return MDT_ERR_WORKSPACE_TOO_SMALL;
    const int nb = match_blocks(n_anchors);
    char *ws = reinterpret_cast<char *>(workspace);
    double *part_val = reinterpret_cast<double *>(ws);
    int *part_idx = reinterpret_cast<int *>(ws + (size_t)nb * n_gt * sizeof(double));
    size_t off = ((size_t)nb * n_gt * (sizeof(double) + sizeof(int)) + 15) & ~(size_t)15;
    double *iou_max_buf = iou_max ? iou_max : reinterpret_cast<double *>(ws + off);
    const size_t lds = ((size_t)n_gt * (2 * dim + 1) + MATCH_THREADS / 64) * sizeof(double) + (MATCH_THREADS / 64) * sizeof(int);
    if (lds > 60 * 1024) return MDT_ERR_UNSUPPORTED;
    if (dim == 3) hipLaunchKernelGGL(match_pass1_kernel<3>, dim3(nb), dim3(MATCH_THREADS), lds, s, anchors, n_anchors, gt_boxes,
                           gt_class_ids, n_gt, (const int *)nullptr, n_gt, neg_thresh, pos_thresh, matches, iou_argmax, iou_max_buf, part_val, part_idx);
    else hipLaunchKernelGGL(match_pass1_kernel<2>, dim3(nb), dim3(MATCH_THREADS), lds, s, anchors, n_anchors, gt_boxes,
                           gt_class_ids, n_gt, (const int *)nullptr, n_gt, neg_thresh, pos_thresh, matches, iou_argmax, iou_max_buf, part_val, part_idx);
    if (check_launch() != MDT_OK) return MDT_ERR_LAUNCH_FAILED;
    (void)hipGetLastError(); hipLaunchKernelGGL(match_pass2_kernel, dim3(1), dim3(256), 0, s, part_val, part_idx, nb, n_gt, (const int *)nullptr, n_gt,
                       n_anchors, gt_class_ids, pos_thresh, iou_max_buf, matches, gt_best_anchor);
    return check_launch();
}

size_t mdt_anchor_match_batched_workspace_bytes(int n_anchors, int batch, int gmax)
{
    if (n_anchors <= 0 || gmax <= 0 || batch <= 0) return 16;
    const size_t nb = (size_t)match_blocks(n_anchors);
    size_t bytes = (size_t)batch * nb * gmax * (sizeof(double) + sizeof(int));   // part_val, part_idx per element
    bytes = (bytes + 15) & ~(size_t)15;
    bytes += (size_t)batch * n_anchors * sizeof(double);                          // iou_max when the caller passes NULL
    return (bytes + 255) & ~(size_t)255;
}

int mdt_anchor_match_batched(const double *anchors, int n_anchors, int dim, int batch,
                             const double *gt_boxes, const int *gt_class_ids, const int *n_gt_dev, int gmax,
                             double neg_thresh, double pos_thresh,
                             int *matches, int *iou_argmax, double *iou_max, int *gt_best_anchor,
                             void *workspace, size_t workspace_bytes, void *stream)
{
    (void)hipGetLastError();
    if (n_anchors < 0 || batch < 0 || gmax < 1 || (dim != 2 && dim != 3) || !matches || !iou_argmax || !gt_boxes || !n_gt_dev || !gt_best_anchor)
        return MDT_ERR_INVALID_ARGUMENT;
    if (n_anchors == 0 || batch == 0) return MDT_OK;
    if (batch > 65535) return MDT_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < mdt_anchor_match_batched_workspace_bytes(n_anchors, batch, gmax)) return MDT_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t s = (hipStream_t)stream;
    const int nb = match_blocks(n_anchors);
    char *ws = reinterpret_cast<char *>(workspace);
    double *part_val = reinterpret_cast<double *>(ws);
    int *part_idx = reinterpret_cast<int *>(ws + (size_t)batch * nb * gmax * sizeof(double));
    const size_t off = ((size_t)batch * nb * gmax * (sizeof(double) + sizeof(int)) + 15) & ~(size_t)15;
    double *iou_max_buf = iou_max ? iou_max : reinterpret_cast<double *>(ws + off);
    const size_t lds = ((size_t)gmax * (2 * dim + 1) + MATCH_THREADS / 64) * sizeof(double) + (MATCH_THREADS / 64) * sizeof(int);
    if (lds > 60 * 1024) return MDT_ERR_UNSUPPORTED;
    if (dim == 3) hipLaunchKernelGGL(match_pass1_kernel<3>, dim3(nb, batch), dim3(MATCH_THREADS), lds, s, anchors, n_anchors, gt_boxes,
                           gt_class_ids, gmax, n_gt_dev, gmax, neg_thresh, pos_thresh, matches, iou_argmax, iou_max_buf, part_val, part_idx);
    else hipLaunchKernelGGL(match_pass1_kernel<2>, dim3(nb, batch), dim3(MATCH_THREADS), lds, s, anchors, n_anchors, gt_boxes,
                           gt_class_ids, gmax, n_gt_dev, gmax, neg_thresh, pos_thresh, matches, iou_argmax, iou_max_buf, part_val, part_idx);
    if (check_launch() != MDT_OK) return MDT_ERR_LAUNCH_FAILED;
    (void)hipGetLastError(); hipLaunchKernelGGL(match_pass2_kernel, dim3(batch), dim3(256), 0, s, part_val, part_idx, nb, gmax, n_gt_dev, gmax,
                       n_anchors, gt_class_ids, pos_thresh, iou_max_buf, matches, gt_best_anchor);
    return check_launch();
}

}  // extern "C"
