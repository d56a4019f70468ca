// upsample.hip -- linear x2 up-sampling of the (y, x) axes on channels-last storage, forward and backward, for gfx950:
// the P2 -> P1 -> P0 top-down path of the Retina U-Net decoder (models/backbone.py: Interpolate(scale_factor=(2, 2, 1),
// mode='trilinear', align_corners=False); 2D: scale 2 'bilinear').  With the z scale at 1 the trilinear stencil is the
// identity along z, so a channels_last_3d tensor [B, C, Y, X, Z] is a [B, Y, X, inner = Z * C] array whose output slab
// (oy, ox, :) is a fixed 2 x 2 blend of input slabs: pure 16-byte streaming.  torch's upsample_trilinear3d kernels take
// 14 ms forward + 11 ms backward per call on the 8 x 36 x 128^3 output (2.4 GB: 0.5 ms at the copy roofline;
// profiles/r03_retina_unet_step_kernels.csv: 11 % of the Retina U-Net step).
//
// Arithmetic = torch's (aten/src/ATen/native/cuda/UpSample.cuh area_pixel_compute_source_index, align_corners = false,
// scale 1/2): src = max(0.5 * (dst + 0.5) - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, n - 1), l1 = src - i0:
//   out[2k] = 0.25 in[k-1] + 0.75 in[k] (out[0] = in[0]),  out[2k+1] = 0.75 in[k] + 0.25 in[k+1] (last: in[n-1]).
// The backward is the gather form of the adjoint (every input slab sums its <= 4 x 4 output taps): no atomics, deterministic.
// HBM-bound; algorithmic bytes: forward 4 * (out + in), backward the same.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mdt_hip.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void src_index(int dst, int n, int &i0, int &i1, float &l1)
{
    float s = 0.5f * ((float)dst + 0.5f) - 0.5f;
    if (s < 0.0f) s = 0.0f;
    i0 = (int)s;
    i1 = min(i0 + 1, n - 1);
    l1 = s - (float)i0;
}

// VEC floats per thread (4 when inner % 4 == 0 and the pointers are 16-byte aligned, else 1)
template <int VEC>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const float *__restrict__ in, float *__restrict__ out, int Y, int X,
                                                           long long inner_v, long long total_v)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total_v; t += stride) {
        const long long slab = t / inner_v;                 // (b, oy, ox)
        const long long i = t - slab * inner_v;
        const int ox = (int)(slab % (2 * X));
        const long long r = slab / (2 * X);
        const int oy = (int)(r % (2 * Y));
        const long long b = r / (2 * Y);
        int y0, y1, x0, x1;
        float ly, lx;
        src_index(oy, Y, y0, y1, ly);
        src_index(ox, X, x0, x1, lx);
        const long long base = b * Y;
        const long long p00 = ((base + y0) * X + x0) * inner_v + i, p01 = ((base + y0) * X + x1) * inner_v + i;
        const long long p10 = ((base + y1) * X + x0) * inner_v + i, p11 = ((base + y1) * X + x1) * inner_v + i;
        if (VEC == 4) {
            const v4f a = reinterpret_cast<const v4f *>(in)[p00], bq = reinterpret_cast<const v4f *>(in)[p01];
            const v4f c = reinterpret_cast<const v4f *>(in)[p10], d = reinterpret_cast<const v4f *>(in)[p11];
            const v4f top = a * (1.0f - lx) + bq * lx, bot = c * (1.0f - lx) + d * lx;
            reinterpret_cast<v4f *>(out)[t] = top * (1.0f - ly) + bot * ly;
        } else {
            const float top = in[p00] * (1.0f - lx) + in[p01] * lx, bot = in[p10] * (1.0f - lx) + in[p11] * lx;
            out[t] = top * (1.0f - ly) + bot * ly;
        }
    }
}

// the <= 4 output indices that read input index k along one axis (n inputs, 2n outputs) and their weights
__device__ __forceinline__ int taps(int k, int n, int *o, float *w)
{
    int c = 0;
    if (k >= 1) { o[c] = 2 * k - 1; w[c] = 0.25f; ++c; }                // out[2(k-1)+1] = 0.75 in[k-1] + 0.25 in[k]
    o[c] = 2 * k; w[c] = (k == 0) ? 1.0f : 0.75f; ++c;                    // out[2k]   = 0.25 in[k-1] + 0.75 in[k]  (out[0] = in[0])
    o[c] = 2 * k + 1; w[c] = (k == n - 1) ? 1.0f : 0.75f; ++c;            // out[2k+1] = 0.75 in[k] + 0.25 in[k+1]  (last: in[n-1])
    if (k + 1 <= n - 1) { o[c] = 2 * k + 2; w[c] = 0.25f; ++c; }          // out[2(k+1)] = 0.25 in[k] + 0.75 in[k+1]
    return c;
}

template <int VEC>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float *__restrict__ gout, float *__restrict__ gin, int Y, int X,
                                                           long long inner_v, long long total_v)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total_v; t += stride) {
        const long long slab = t / inner_v;                 // (b, iy, ix)
        const long long i = t - slab * inner_v;
        const int ix = (int)(slab % X);
        const long long r = slab / X;
        const int iy = (int)(r % Y);
        const long long b = r / Y;
        int oy[4], ox[4];
        float wy[4], wx[4];
        const int ny = taps(iy, Y, oy, wy), nx = taps(ix, X, ox, wx);
        const long long base = b * 2 * Y;
        if (VEC == 4) {
            v4f acc = {0.f, 0.f, 0.f, 0.f};
            for (int a = 0; a < ny; ++a) {
                v4f row = {0.f, 0.f, 0.f, 0.f};
                for (int c = 0; c < nx; ++c)
                    row = row + reinterpret_cast<const v4f *>(gout)[((base + oy[a]) * (2 * X) + ox[c]) * inner_v + i] * wx[c];
                acc = acc + row * wy[a];
            }
            reinterpret_cast<v4f *>(gin)[t] = acc;
        } else {
            float acc = 0.0f;
            for (int a = 0; a < ny; ++a) {
                float row = 0.0f;
                for (int c = 0; c < nx; ++c) row = row + gout[((base + oy[a]) * (2 * X) + ox[c]) * inner_v + i] * wx[c];
                acc = acc + row * wy[a];
            }
            gin[t] = acc;
        }
    }
}

inline int check()
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

inline unsigned blocks_for(long long total)
{
    long long b = (total + 255) / 256;
    if (b > 65536) b = 65536;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" {

int mdt_upsample2x_yx_cl_forward(const float *in, float *out, long long batch, int Y, int X, long long inner, void *stream)
{
    if (!in || !out || batch <= 0 || Y <= 0 || X <= 0 || inner <= 0) return MDT_ERR_INVALID_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool v4 = (inner % 4 == 0) && ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0;
    const long long total = batch * 4 * Y * X * inner;
    (void)hipGetLastError();
    if (v4) hipLaunchKernelGGL(upsample2x_fwd_kernel<4>, dim3(blocks_for(total / 4)), dim3(256), 0, s, in, out, Y, X, inner / 4, total / 4);
    else hipLaunchKernelGGL(upsample2x_fwd_kernel<1>, dim3(blocks_for(total)), dim3(256), 0, s, in, out, Y, X, inner, total);
    return check();
}

int mdt_upsample2x_yx_cl_backward(const float *grad_out, float *grad_in, long long batch, int Y, int X, long long inner, void *stream)
{
    if (!grad_out || !grad_in || batch <= 0 || Y <= 0 || X <= 0 || inner <= 0) return MDT_ERR_INVALID_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool v4 = (inner % 4 == 0) && ((((uintptr_t)grad_out) | ((uintptr_t)grad_in)) & 15) == 0;
    const long long total = batch * Y * X * inner;
    (void)hipGetLastError();
    if (v4) hipLaunchKernelGGL(upsample2x_bwd_kernel<4>, dim3(blocks_for(total / 4)), dim3(256), 0, s, grad_out, grad_in, Y, X, inner / 4, total / 4);
    else hipLaunchKernelGGL(upsample2x_bwd_kernel<1>, dim3(blocks_for(total)), dim3(256), 0, s, grad_out, grad_in, Y, X, inner, total);
    return check();
}

}  // extern "C"
