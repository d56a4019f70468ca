// boxes.hip -- fused box decode + clip (+ score append) for gfx950.
//
// Replaces the chain of ~20 tiny elementwise launches in the reference:
//   apply_box_deltas_{2D,3D}  utils/model_utils.py:318-370
//   clip_boxes_{2D,3D}        utils/model_utils.py:374-398
//   deltas * std_dev, gather by `order`, cat(boxes, scores)   models/mrcnn.py:320-345
// Each arithmetic step is kept as its own fp32 operation in the reference's
// order (this TU is built with -ffp-contract=off); the only op that may differ
// from torch in the last ulp is expf.  HBM-bound elementwise work.

#include <hip/hip_runtime.h>
#include "mdt_hip.h"

namespace {

struct DecodeParams {
    float std_dev[6];
    float window[6];
};

template <int DIM>
__global__ __launch_bounds__(256) void decode_clip_kernel(
    const float *__restrict__ boxes, const float *__restrict__ deltas,
    const long long *__restrict__ order, const float *__restrict__ scores,
    int n, DecodeParams prm, float *__restrict__ out, int out_stride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long src = order ? order[i] : (long long)i;
    const float *b = boxes + src * (2 * DIM);
    const float *d = deltas + src * (2 * DIM);
    float *o = out + (long long)i * out_stride;

    // model_utils.py:349-369 (3D) / :324-339 (2D)
    float height = b[2] - b[0];
    float width = b[3] - b[1];
    float center_y = b[0] + 0.5f * height;
    float center_x = b[1] + 0.5f * width;
    if (DIM == 3) {
        float depth = b[5] - b[4];
        float center_z = b[4] + 0.5f * depth;
        center_y = center_y + (d[0] * prm.std_dev[0]) * height;
        center_x = center_x + (d[1] * prm.std_dev[1]) * width;
        center_z = center_z + (d[2] * prm.std_dev[2]) * depth;
        height = height * expf(d[3] * prm.std_dev[3]);
        width = width * expf(d[4] * prm.std_dev[4]);
        depth = depth * expf(d[5] * prm.std_dev[5]);
        const float y1 = center_y - 0.5f * height;
        const float x1 = center_x - 0.5f * width;
        const float z1 = center_z - 0.5f * depth;
        const float y2 = y1 + height;
        const float x2 = x1 + width;
        const float z2 = z1 + depth;
        // clip_boxes_3D: (y1,y2) to window[0],[2]; (x1,x2) to [1],[3]; (z1,z2) to [4],[5]
        o[0] = fminf(fmaxf(y1, prm.window[0]), prm.window[2]);
        o[1] = fminf(fmaxf(x1, prm.window[1]), prm.window[3]);
        o[2] = fminf(fmaxf(y2, prm.window[0]), prm.window[2]);
        o[3] = fminf(fmaxf(x2, prm.window[1]), prm.window[3]);
        o[4] = fminf(fmaxf(z1, prm.window[4]), prm.window[5]);
        o[5] = fminf(fmaxf(z2, prm.window[4]), prm.window[5]);
        if (out_stride > 6) o[6] = scores ? scores[i] : 0.0f;
    } else {
        center_y = center_y + (d[0] * prm.std_dev[0]) * height;
        center_x = center_x + (d[1] * prm.std_dev[1]) * width;
        height = height * expf(d[2] * prm.std_dev[2]);
        width = width * expf(d[3] * prm.std_dev[3]);
        const float y1 = center_y - 0.5f * height;
        const float x1 = center_x - 0.5f * width;
        const float y2 = y1 + height;
        const float x2 = x1 + width;
        o[0] = fminf(fmaxf(y1, prm.window[0]), prm.window[2]);
        o[1] = fminf(fmaxf(x1, prm.window[1]), prm.window[3]);
        o[2] = fminf(fmaxf(y2, prm.window[0]), prm.window[2]);
        o[3] = fminf(fmaxf(x2, prm.window[1]), prm.window[3]);
        if (out_stride > 4) o[4] = scores ? scores[i] : 0.0f;
    }
}

}  // namespace

extern "C" int mdt_decode_clip_boxes(const float *boxes, const float *deltas, const long long *order,
                                     const float *scores, int n, int dim,
                                     const float *std_dev_host, const float *window_host,
                                     float *out, int out_stride, void *stream)
{
    (void)hipGetLastError();   // drop stale error state of earlier runtime calls on this thread
    if (n < 0 || (dim != 2 && dim != 3) || out_stride < 2 * dim || !std_dev_host || !window_host)
        return MDT_ERR_INVALID_ARGUMENT;
    if (n == 0) return MDT_OK;
    DecodeParams prm;
    for (int i = 0; i < 6; ++i) {
        prm.std_dev[i] = i < 2 * dim ? std_dev_host[i] : 1.0f;
        prm.window[i] = i < 2 * dim ? window_host[i] : 0.0f;
    }
    const int blocks = (n + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    if (dim == 3) hipLaunchKernelGGL(decode_clip_kernel<3>, dim3(blocks), dim3(256), 0, s, boxes, deltas, order, scores, n, prm, out, out_stride);
    else hipLaunchKernelGGL(decode_clip_kernel<2>, dim3(blocks), dim3(256), 0, s, boxes, deltas, order, scores, n, prm, out, out_stride);
    return hipGetLastError() == hipSuccess ? MDT_OK : MDT_ERR_LAUNCH_FAILED;
}
