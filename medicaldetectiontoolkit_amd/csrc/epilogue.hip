// epilogue.hip -- fused convolution epilogues for the FPN / ResNet conv path (gfx950).
//
// The backbone convolutions themselves stay on MIOpen / CK (north star).  What torch wraps around every one of them
// is not: a broadcast bias add, the residual / top-down add and the ReLU are three separate full passes over the
// activation (reference graph: models/backbone.py:147-153,197-206, utils/model_utils.py:751-781), and in backward the
// ReLU mask and the bias-gradient reduction are two more.  At fp32 these passes are pure HBM traffic
// (profiles/r01_bench_train_step_steady_state_kernels.csv: ~10 % of the step).  Here:
//
//   forward   y = act(x + bias[c] (+ residual))            one pass, 16-byte accesses, in place on the conv output
//   backward  gx = gy * (y > 0)   and   gbias[c] = sum gx  one pass + a tiny deterministic second stage
//
// Both memory orders of torch are handled through `inner`: channel(i) = (i / inner) % C with inner = 1 for
// channels_last(_3d) storage and inner = prod(spatial) for contiguous NC(D)HW.  No atomics: the bias gradient is
// reduced in a fixed order (per-thread running sums -> per-block partials -> one thread per channel), so training
// stays run-to-run deterministic.  HBM-bound elementwise work, no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mdt_hip.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int EP_THREADS = 256;
constexpr int EP_MAX_BLOCKS = 1024;

inline int ep_check()
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

// ---- forward ---------------------------------------------------------------------------------------------------
// CL: channels-last storage (channel = i % C); else contiguous (channel = (i / inner) % C, inner % 4 == 0 on the vector path)
template <bool CL, bool RES, bool RELU>
// y and x (and res) may alias: the Python side runs the epilogue IN PLACE (y == x); each thread reads its elements before it
// writes the same ones, so no restrict qualifier on them
__global__ __launch_bounds__(EP_THREADS) void bias_act_fwd_kernel(float *y, const float *x,
                                                                  const float *__restrict__ bias, const float *res,
                                                                  long long n4, long long n, int C, long long inner)
{
    const long long stride = (long long)gridDim.x * EP_THREADS;
    for (long long i4 = (long long)blockIdx.x * EP_THREADS + threadIdx.x; i4 < n4; i4 += stride) {
        const long long i = i4 * 4;
        v4f v = reinterpret_cast<const v4f *>(x)[i4];
        float b[4];
        if (CL) {
            int c = (int)(i % C);
#pragma unroll
            for (int k = 0; k < 4; ++k) { b[k] = bias[c]; c = (c + 1 == C) ? 0 : c + 1; }
        } else {
            const float bb = bias[(int)((i / inner) % C)];      // inner % 4 == 0: the four share one channel
            b[0] = b[1] = b[2] = b[3] = bb;
        }
        if (RES) {
            const v4f r = reinterpret_cast<const v4f *>(res)[i4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] + b[k] + r[k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] + b[k];
        }
        if (RELU) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.0f ? v[k] : 0.0f;
        }
        reinterpret_cast<v4f *>(y)[i4] = v;
    }
    // scalar tail (n % 4 elements) and the fully scalar fallback are handled by the launcher with n4 = 0 .. see below
    if (blockIdx.x == 0) {
        for (long long i = n4 * 4 + threadIdx.x; i < n; i += EP_THREADS) {
            const int c = CL ? (int)(i % C) : (int)((i / inner) % C);
            float v = x[i] + bias[c];
            if (RES) v = v + res[i];
            if (RELU) v = v > 0.0f ? v : 0.0f;
            y[i] = v;
        }
    }
}

// ---- backward, channels-last -----------------------------------------------------------------------------------
// Thread t of a block owns channel slots; a block walks a contiguous range of pixels.  C < 256: ppb = 256 / C pixels are
// processed side by side (thread t -> pixel lane t / C, channel t % C) and the lanes are folded in LDS in lane order;
// C >= 256: thread t owns channels t, t + 256, ... of every pixel.  Per-block partials go to `partial[block][C]`.
template <bool RELU>
__global__ __launch_bounds__(EP_THREADS) void bias_act_bwd_cl_kernel(float *__restrict__ gx, const float *__restrict__ gy,
                                                                     const float *__restrict__ y, float *__restrict__ partial,
                                                                     long long npix, int C, int ppb, int ktiles, long long pix_per_block)
{
    __shared__ float s_acc[EP_THREADS];
    const int t = threadIdx.x;
    const long long p0 = (long long)blockIdx.x * pix_per_block;
    long long p1 = p0 + pix_per_block;
    if (p1 > npix) p1 = npix;
    if (ppb >= 1 && ktiles == 1) {
        const int lanes = ppb * C;                 // active threads
        const int pl = t / C, c = t - pl * C;
        float acc = 0.0f;
        if (t < lanes) {
            long long p = p0 + pl;
            for (; p + 3LL * ppb < p1; p += 4LL * ppb) {          // four independent loads in flight
                float g[4], v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const long long i = (p + (long long)u * ppb) * C + c; g[u] = gy[i]; v[u] = RELU ? y[i] : 1.0f; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (RELU) g[u] = (v[u] > 0.0f) ? g[u] : 0.0f;
                    gx[(p + (long long)u * ppb) * C + c] = g[u];
                    acc = acc + g[u];
                }
            }
            for (; p < p1; p += ppb) {
                const long long i = p * C + c;
                float g = gy[i];
                if (RELU) g = (y[i] > 0.0f) ? g : 0.0f;
                gx[i] = g;
                acc = acc + g;
            }
        }
        s_acc[t] = acc;
        __syncthreads();
        if (t < C) {
            float s = 0.0f;
            for (int l = 0; l < ppb; ++l) s = s + s_acc[l * C + t];     // fixed lane order
            partial[(long long)blockIdx.x * C + t] = s;
        }
    } else {
        for (int k = 0; k < ktiles; ++k) {
            const int c = t + k * EP_THREADS;
            float acc = 0.0f;
            if (c < C) {
                long long p = p0;
                for (; p + 3 < p1; p += 4) {
                    float g[4], v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const long long i = (p + u) * C + c; g[u] = gy[i]; v[u] = RELU ? y[i] : 1.0f; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (RELU) g[u] = (v[u] > 0.0f) ? g[u] : 0.0f;
                        gx[(p + u) * C + c] = g[u];
                        acc = acc + g[u];
                    }
                }
                for (; p < p1; ++p) {
                    const long long i = p * C + c;
                    float g = gy[i];
                    if (RELU) g = (y[i] > 0.0f) ? g : 0.0f;
                    gx[i] = g;
                    acc = acc + g;
                }
                partial[(long long)blockIdx.x * C + c] = acc;
            }
        }
    }
}

// ---- backward, contiguous NC(D)HW: one block per chunk of one (n, c) row ---------------------------------------
template <bool RELU>
__global__ __launch_bounds__(EP_THREADS) void bias_act_bwd_nc_kernel(float *__restrict__ gx, const float *__restrict__ gy,
                                                                     const float *__restrict__ y, float *__restrict__ partial,
                                                                     long long inner, int chunks, long long chunk_len)
{
    __shared__ float s_acc[EP_THREADS];
    const long long row = blockIdx.x / chunks;         // n * C + c
    const int chunk = blockIdx.x % chunks;
    const long long e0 = (long long)chunk * chunk_len;
    long long e1 = e0 + chunk_len;
    if (e1 > inner) e1 = inner;
    float acc = 0.0f;
    for (long long e = e0 + threadIdx.x; e < e1; e += EP_THREADS) {
        const long long i = row * inner + e;
        float g = gy[i];
        if (RELU) g = (y[i] > 0.0f) ? g : 0.0f;
        gx[i] = g;
        acc = acc + g;
    }
    s_acc[threadIdx.x] = acc;
    __syncthreads();
    for (int s = EP_THREADS / 2; s > 0; s >>= 1) {      // fixed tree
        if ((int)threadIdx.x < s) s_acc[threadIdx.x] = s_acc[threadIdx.x] + s_acc[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s_acc[0];
}

// second stage: gbias[c] = sum of the partials of channel c.  One block per channel: every thread adds a fixed strided
// subset in index order, then a fixed LDS tree -- deterministic, and no serial chain of dependent global loads
__global__ __launch_bounds__(EP_THREADS) void bias_grad_finish_kernel(float *__restrict__ gbias, const float *__restrict__ partial,
                                                                      int C, long long count, long long stride_c, long long stride_k,
                                                                      long long groups, long long stride_g)
{
    __shared__ float s_acc[EP_THREADS];
    const int c = blockIdx.x;
    const long long total = groups * count;
    float s = 0.0f;
    for (long long j = threadIdx.x; j < total; j += EP_THREADS) {
        const long long g = j / count, k = j - g * count;
        s = s + partial[g * stride_g + (long long)c * stride_c + k * stride_k];
    }
    s_acc[threadIdx.x] = s;
    __syncthreads();
    for (int h = EP_THREADS / 2; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) s_acc[threadIdx.x] = s_acc[threadIdx.x] + s_acc[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) gbias[c] = s_acc[0];
}

}  // namespace

extern "C" {

int mdt_bias_act_forward(float *y, const float *x, const float *bias, const float *residual,
                         long long n, int channels, long long inner, int relu, void *stream)
{
    if (n < 0 || channels <= 0 || inner <= 0 || (n % ((long long)channels * inner)) != 0) return MDT_ERR_INVALID_ARGUMENT;
    if (n == 0) return MDT_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool cl = inner == 1;
    const bool aligned = ((((uintptr_t)y) | ((uintptr_t)x) | ((uintptr_t)residual)) & 15) == 0;
    const long long n4 = (aligned && (cl || inner % 4 == 0)) ? n / 4 : 0;
    long long blocks = (n4 + EP_THREADS - 1) / EP_THREADS;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    (void)hipGetLastError();
#define LAUNCH_FWD(CL, RES, RELU) hipLaunchKernelGGL((bias_act_fwd_kernel<CL, RES, RELU>), dim3((unsigned)blocks), dim3(EP_THREADS), 0, s, \
                                                     y, x, bias, residual, n4, n, channels, inner)
    const bool res = residual != nullptr;
    if (cl) {
        if (res) { if (relu) LAUNCH_FWD(true, true, true); else LAUNCH_FWD(true, true, false); }
        else { if (relu) LAUNCH_FWD(true, false, true); else LAUNCH_FWD(true, false, false); }
    } else {
        if (res) { if (relu) LAUNCH_FWD(false, true, true); else LAUNCH_FWD(false, true, false); }
        else { if (relu) LAUNCH_FWD(false, false, true); else LAUNCH_FWD(false, false, false); }
    }
#undef LAUNCH_FWD
    return ep_check();
}

size_t mdt_bias_act_backward_workspace_bytes(long long n, int channels, long long inner)
{
    if (n <= 0 || channels <= 0 || inner <= 0) return 256;
    if (inner == 1) return (size_t)4 * EP_MAX_BLOCKS * channels * sizeof(float) + 256;
    const long long rows = n / inner;
    long long chunks = (EP_MAX_BLOCKS * 2 + rows - 1) / rows;
    if (chunks < 1) chunks = 1;
    return (size_t)(rows * chunks) * sizeof(float) + 256;
}

int mdt_bias_act_backward(float *gx, const float *gy, const float *y, float *gbias,
                          long long n, int channels, long long inner, int relu,
                          void *workspace, size_t workspace_bytes, void *stream)
{
    if (n < 0 || channels <= 0 || inner <= 0 || (n % ((long long)channels * inner)) != 0) return MDT_ERR_INVALID_ARGUMENT;
    if (relu && y == nullptr) return MDT_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    if (workspace == nullptr || workspace_bytes < mdt_bias_act_backward_workspace_bytes(n, channels, inner)) return MDT_ERR_WORKSPACE_TOO_SMALL;
    float *partial = reinterpret_cast<float *>(workspace);
    (void)hipGetLastError();
    const int fin_blocks = channels;
    if (n == 0) {
        hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(fin_blocks), dim3(EP_THREADS), 0, s, gbias, partial, channels, 0LL, 0LL, 0LL, 0LL, 0LL);
        return ep_check();
    }
    if (inner == 1) {
        const long long npix = n / channels;
        const int ktiles = (channels + EP_THREADS - 1) / EP_THREADS;
        const int ppb = (ktiles == 1) ? EP_THREADS / channels : 1;
        // enough blocks to fill the chip, each with a contiguous pixel range (a multiple of ppb)
        long long blocks = (npix + 8LL * ppb - 1) / (8LL * ppb);      // ~8 pixel iterations per thread
        if (blocks > 4 * EP_MAX_BLOCKS) blocks = 4 * EP_MAX_BLOCKS;
        if (blocks < 1) blocks = 1;
        long long ppblock = (npix + blocks - 1) / blocks;
        ppblock = ((ppblock + ppb - 1) / ppb) * ppb;
        blocks = (npix + ppblock - 1) / ppblock;
        if (relu) hipLaunchKernelGGL(bias_act_bwd_cl_kernel<true>, dim3((unsigned)blocks), dim3(EP_THREADS), 0, s, gx, gy, y, partial, npix, channels, ppb, ktiles, ppblock);
        else hipLaunchKernelGGL(bias_act_bwd_cl_kernel<false>, dim3((unsigned)blocks), dim3(EP_THREADS), 0, s, gx, gy, y, partial, npix, channels, ppb, ktiles, ppblock);
        if (ep_check() != MDT_OK) return MDT_ERR_LAUNCH_FAILED;
        // partial[block][c]: channel stride 1, block stride C
        hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(fin_blocks), dim3(EP_THREADS), 0, s, gbias, partial, channels, blocks, 1LL, (long long)channels, 1LL, 0LL);
        return ep_check();
    }
    const long long rows = n / inner;                       // batch * C
    long long chunks = (EP_MAX_BLOCKS * 2 + rows - 1) / rows;
    if (chunks < 1) chunks = 1;
    if (chunks > (inner + 1023) / 1024) chunks = (inner + 1023) / 1024;
    if (chunks < 1) chunks = 1;
    const long long chunk_len = (inner + chunks - 1) / chunks;
    const long long blocks = rows * chunks;
    if (blocks > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    if (relu) hipLaunchKernelGGL(bias_act_bwd_nc_kernel<true>, dim3((unsigned)blocks), dim3(EP_THREADS), 0, s, gx, gy, y, partial, inner, (int)chunks, chunk_len);
    else hipLaunchKernelGGL(bias_act_bwd_nc_kernel<false>, dim3((unsigned)blocks), dim3(EP_THREADS), 0, s, gx, gy, y, partial, inner, (int)chunks, chunk_len);
    if (ep_check() != MDT_OK) return MDT_ERR_LAUNCH_FAILED;
    // partial[(n * C + c) * chunks + k]: channel stride chunks, chunk stride 1, batch groups stride C * chunks
    hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(fin_blocks), dim3(EP_THREADS), 0, s, gbias, partial, channels, chunks, chunks, 1LL,
                       rows / channels, (long long)channels * chunks);
    return ep_check();
}

}  // extern "C"
