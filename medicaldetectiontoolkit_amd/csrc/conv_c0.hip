// conv_c0.hip -- the FIRST layer of the stride-1 backbone (models/backbone.py:60-63, operate_stride1: C0 = conv(1 -> 18, ks 3, pad 1) + ReLU on the full-
// resolution volume; the Retina U-Net's first convolution), forward and backward, for gfx950.
//
// A one-channel volume is "contiguous" in both of torch's layouts, so the library runs this layer row-major: at 8 x 128^3 the step spent 1.1 + 0.7 ms in
// MIOpen's layout transposes around a 1.4 ms convolution, 0.9 ms converting the 2.4 GB output to channels-last for the next layer, and in the backward
// 3.2 + 0.9 ms converting the gradient back, 0.9 ms on the ReLU / bias pass and 2.9 ms on the weight gradient: 12.5 ms for a layer of 16 GFLOP whose
// operands are 67 MB in and 1.2 GB out.  Here:
//   forward   y[v][co] = relu(bias[co] + sum_tap w[co][tap] * x[v + tap])   one thread per voxel on the VALU (486 FMAs), the 256-voxel x 18 result tile goes
//             through LDS and leaves as one contiguous 18 KB run of the channels-last output.  HBM-bound: 4 * (1 + C_out) bytes per voxel.
//   backward  gw[co][tap] = sum_v g[v][co] * x[v + tap],  gbias[co] = sum_v g[v][co],  g = gy * (y > 0)   -- ONE pass over gy and y on the fp32 matrix cores:
//             D[tap | ones][co] += A[tap | ones][voxel] * B[voxel][co]; A rows are the 27 shifted copies of the volume (z-contiguous: 16-byte loads feed
//             four MFMAs) and a row of ones (the bias gradient), B is the masked gradient, staged per wave through LDS from contiguous 16-byte loads.
//             Per-wave partial 32 x 32 blocks are folded in a fixed order by a second kernel (deterministic).  HBM-bound: 8 * C_out bytes per voxel.
// The input has no gradient (it is the image).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mdt_hip.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

inline int c0_check()
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

constexpr int C0_CO = 18;
constexpr int C0_THREADS = 256;

// ---- forward ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(C0_THREADS) void conv_c0_fwd_kernel(float *__restrict__ y, const float *__restrict__ x, const float *__restrict__ wt,
                                                                 const float *__restrict__ bias, int relu, int Y, int X, int Z, long long V)
{
    __shared__ __attribute__((aligned(16))) float s_t[C0_THREADS * C0_CO];      // the block's output tile
    const long long v0 = (long long)blockIdx.x * C0_THREADS;
    const long long v = v0 + threadIdx.x;
    if (v < V) {
        const int z = (int)(v % Z);
        const long long t = v / Z;
        const int xx = (int)(t % X);
        const long long t2 = t / X;
        const int yy = (int)(t2 % Y);
        float acc[C0_CO];
#pragma unroll
        for (int co = 0; co < C0_CO; ++co) acc[co] = 0.0f;
#pragma unroll 1                // (the filter values are wave-uniform: scalar loads of wt[tap][0 .. 17]; fully unrolled they would not fit the scalar registers)
        for (int tap = 0; tap < 27; ++tap) {
            const int dz = tap % 3 - 1, dx = (tap / 3) % 3 - 1, dy = tap / 9 - 1;
            const bool ok = (unsigned)(yy + dy) < (unsigned)Y && (unsigned)(xx + dx) < (unsigned)X && (unsigned)(z + dz) < (unsigned)Z;
            float xv = x[ok ? v + ((long long)dy * X + dx) * Z + dz : v];
            if (!ok) xv = 0.0f;
            const float *wp = wt + tap * C0_CO;
#pragma unroll
            for (int co = 0; co < C0_CO; ++co) acc[co] = acc[co] + xv * wp[co];                  // taps ascending: a fixed order
        }
#pragma unroll
        for (int co = 0; co < C0_CO; ++co) {
            float r = acc[co] + (bias ? bias[co] : 0.0f);
            if (relu) r = r > 0.0f ? r : 0.0f;
            s_t[threadIdx.x * C0_CO + co] = r;
        }
    }
    __syncthreads();
    const long long nv = min((long long)C0_THREADS, V - v0);
    const int n4 = (int)(nv * C0_CO / 4);                    // (nv * 18) % 4 != 0 only in a ragged last block: scalar tail
    v4f *dst = reinterpret_cast<v4f *>(y + v0 * C0_CO);      // v0 * 18 floats: 256 * 72 bytes per block, 16-byte aligned
    for (int e = threadIdx.x; e < n4; e += C0_THREADS) dst[e] = reinterpret_cast<const v4f *>(s_t)[e];
    for (int e = n4 * 4 + threadIdx.x; e < (int)(nv * C0_CO); e += C0_THREADS) y[v0 * C0_CO + e] = s_t[e];
}

// ---- backward: weight + bias gradient ---------------------------------------------------------------------------------------------------------
// A wave walks chunks of 32 consecutive voxels of one z line (Z % 32 == 0): 16 MFMAs of 2 voxels each.  K is permuted inside a chunk: half-wave h of MFMA j
// multiplies voxel 16 h + j, so that a lane's A operands (one tap, 16 consecutive z) are four 16-byte loads.
constexpr int C0_BS = C0_CO + 1;            // LDS row stride of the masked-gradient tile (odd: conflict-free fragment reads)

__global__ __launch_bounds__(C0_THREADS) void conv_c0_wgrad_kernel(float *__restrict__ partial, const float *__restrict__ gy, const float *__restrict__ yout,
                                                                   const float *__restrict__ x, int relu, int Y, int X, int Z, long long chunks,
                                                                   long long chunks_per_wave)
{
    __shared__ float s_g[C0_THREADS / 64][32 * C0_BS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const long long gw = (long long)blockIdx.x * (C0_THREADS / 64) + wave;
    const long long c0 = gw * chunks_per_wave, c1 = min(chunks, c0 + chunks_per_wave);
    float *sg = s_g[wave];
    // row `col` of A: tap col (< 27), the ones row (27: bias gradient), or nothing (28 .. 31)
    const int tap = col < 27 ? col : 13;
    const int dz = tap % 3 - 1, dx = (tap / 3) % 3 - 1, dy = tap / 9 - 1;
    const int zpc = Z / 32;                                  // chunks per z line
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (long long c = c0; c < c1; ++c) {
        const long long line = c / zpc;                      // (b, y, x) line
        const int zc = (int)(c - line * zpc) * 32;
        const int xx = (int)(line % X);
        const int yy = (int)((line / X) % Y);
        const long long vbase = line * Z + zc;               // first voxel of the chunk
        // B: the chunk's masked gradient, 32 x 18 floats = 144 contiguous 16-byte pieces -> LDS [voxel][18 | 1]
        const v4f *g4 = reinterpret_cast<const v4f *>(gy + vbase * C0_CO);
        const v4f *y4 = reinterpret_cast<const v4f *>(yout + vbase * C0_CO);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e4 = lane + 64 * i;
            if (e4 < 144) {
                v4f g = g4[e4];
                if (relu) {
                    const v4f yv = y4[e4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = yv[j] > 0.0f ? g[j] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int e = e4 * 4 + j, vx = e / C0_CO, co = e - vx * C0_CO; sg[vx * C0_BS + co] = g[j]; }
            }
        }
        // A: this lane's tap at voxels zc + 16 half .. + 15 (one z line of the volume; the ends of the line and the y / x borders are zero padding)
        float a[16];
        if (col < 27) {
            const bool lok = (unsigned)(yy + dy) < (unsigned)Y && (unsigned)(xx + dx) < (unsigned)X;
            const long long src = (line + (lok ? (long long)dy * X + dx : 0)) * Z;       // the source line
            const int z0 = zc + 16 * half + dz;                                          // first source z of this lane
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int zz = z0 + j;
                const bool ok = lok && (unsigned)zz < (unsigned)Z;
                const float v = x[src + (ok ? zz : 0)];
                a[j] = ok ? v : 0.0f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = col == 27 ? 1.0f : 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float b = col < C0_CO ? sg[(16 * half + j) * C0_BS + col] : 0.0f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b, acc, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                     // the tile is rewritten by the next chunk
    }
    // D[row = tap | ones][col = co] -> partial[wave][row][col] (28 x 18 used)
    float *pw = partial + gw * (32 * 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        pw[row * 32 + col] = acc[r];
    }
}

// gw[co][tap] = sum over waves (ascending within a thread's strided subset, then a fixed LDS tree) of partial[wave][tap][co]; row 27 -> gbias[co]
__global__ __launch_bounds__(256) void conv_c0_wgrad_fold_kernel(float *__restrict__ gw, float *__restrict__ gbias, const float *__restrict__ partial, long long nw)
{
    __shared__ float s_acc[256];
    const int row = blockIdx.x / C0_CO, co = blockIdx.x - row * C0_CO;        // row 0 .. 27
    float s = 0.0f;
    for (long long j = threadIdx.x; j < nw; j += 256) s = s + partial[j * 1024 + row * 32 + co];
    s_acc[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) s_acc[threadIdx.x] = s_acc[threadIdx.x] + s_acc[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (row < 27) gw[co * 27 + row] = s_acc[0];
        else if (gbias) gbias[co] = s_acc[0];
    }
}

long long c0_waves(long long chunks)
{
    long long w = 256LL * 16;                               // four waves per SIMD
    if (w > chunks) w = chunks;
    if (w < 1) w = 1;
    return ((w + 3) / 4) * 4;
}

}  // namespace

extern "C" {

int mdt_conv_c0_supported(int c_in, int c_out, int k, int Z) { return (c_in == 1 && c_out == C0_CO && k == 3 && Z > 0 && Z % 32 == 0) ? 1 : 0; }

int mdt_conv_c0_forward(const float *x, const float *w, const float *bias, int relu, float *y, int batch, int Y, int X, int Z, int c_out, void *stream)
{
    if (!x || !w || !y || batch < 0 || Y <= 0 || X <= 0 || Z <= 0) return MDT_ERR_INVALID_ARGUMENT;
    if (c_out != C0_CO) return MDT_ERR_UNSUPPORTED;
    const long long V = (long long)batch * Y * X * Z;
    if (V == 0) return MDT_OK;
    if (((uintptr_t)y & 15) != 0 || (V + C0_THREADS - 1) / C0_THREADS > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    (void)hipGetLastError();
    hipLaunchKernelGGL(conv_c0_fwd_kernel, dim3((unsigned)((V + C0_THREADS - 1) / C0_THREADS)), dim3(C0_THREADS), 0, static_cast<hipStream_t>(stream), y, x, w, bias,
                       relu ? 1 : 0, Y, X, Z, V);
    return c0_check();
}

size_t mdt_conv_c0_wgrad_workspace_bytes(int batch, int Y, int X, int Z)
{
    if (batch <= 0 || Y <= 0 || X <= 0 || Z <= 0) return 256;
    const long long chunks = (long long)batch * Y * X * (Z / 32);
    return (size_t)c0_waves(chunks) * 1024 * sizeof(float) + 256;
}

int mdt_conv_c0_backward(const float *gy, const float *y, const float *x, int relu, float *grad_weight, float *grad_bias, int batch, int Y, int X, int Z,
                         int c_out, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!gy || !x || !grad_weight || batch < 0 || Y <= 0 || X <= 0 || Z <= 0 || (relu && !y)) return MDT_ERR_INVALID_ARGUMENT;
    if (!mdt_conv_c0_supported(1, c_out, 3, Z)) return MDT_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < mdt_conv_c0_wgrad_workspace_bytes(batch, Y, X, Z)) return MDT_ERR_WORKSPACE_TOO_SMALL;
    if ((((uintptr_t)gy | (uintptr_t)y) & 15) != 0) return MDT_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *partial = static_cast<float *>(workspace);
    const long long chunks = (long long)batch * Y * X * (Z / 32);
    const long long nw = chunks > 0 ? c0_waves(chunks) : 0;
    (void)hipGetLastError();
    if (nw > 0) {
        const long long cpw = (chunks + nw - 1) / nw;
        hipLaunchKernelGGL(conv_c0_wgrad_kernel, dim3((unsigned)(nw / 4)), dim3(C0_THREADS), 0, s, partial, gy, y ? y : gy, x, relu ? 1 : 0, Y, X, Z, chunks, cpw);
        if (c0_check() != MDT_OK) return MDT_ERR_LAUNCH_FAILED;
    }
    hipLaunchKernelGGL(conv_c0_wgrad_fold_kernel, dim3(28 * C0_CO), dim3(256), 0, s, grad_weight, grad_bias, partial, nw);
    return c0_check();
}

}  // extern "C"
