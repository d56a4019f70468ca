// conv_seg.hip -- 3x3x3 convolution (stride 1, padding 1) between a MANY-channel and a TWO-channel full-resolution map, channels-last fp32 (gfx950):
// the Retina U-Net's segmentation branch.  The reference computes  seg_logits = final_conv(P0_conv2(p0))  (models/retina_unet.py:483-486 with
// models/backbone.py:160-176): a 36 -> 36 3x3x3 layer WITHOUT activation followed by a 36 -> 2 1x1x1 layer, and nothing else reads P0_conv2's output.  Two
// linear layers in a row are one: W'[s][ci][tap] = sum_co Wf[s][co] * W2[co][ci][tap], b' = Wf b2 + bf (composed with torch ops from the two modules'
// parameters, so autograd hands each its gradient; models/retina_unet.py).  At 8 x 128^3 that replaces 58 ms of library convolutions (36 -> 36: 17.4 forward +
// 17.5 input gradient + 23.2 weight gradient, 131 GFLOP each) by 36 <-> 2 layers of 65 GFLOP each that no library kernel is shaped for (N = 2 pads to 32 on the
// matrix cores).  Three kernels:
//   forward  (36 -> 2)  VALU.  A workgroup owns a 2 x 4 x 16 voxel tile; its 4 x 6 x 18 input lines sit in LDS once (3.4 x the tile's own bytes instead of 27 x);
//            three thread groups of 12 input channels each (16-byte LDS reads, filter values wave-uniform: scalar registers), partial sums folded in group order.
//   input gradient (2 -> 36)  VALU.  One thread per voxel: its 27 x 2 neighbourhood of the tiny output gradient from the caches, 1 944 FMAs against wave-uniform
//            filter values, the 36 results leave through an LDS tile as one contiguous run.
//   weight gradient  fp32 MFMA.  D[(tap, s)][ci] += g[v + tap][s] * x[v][ci] (= the filter gradient at the mirrored tap): rows = the 54 shifted copies of the gradient, columns = the input channels (+ a ones
//            column whose centre-tap rows are the bias gradient), K = voxels; x chunks go through a per-wave LDS tile, per-wave partial blocks are folded in a
//            fixed order (deterministic).
// HBM-bound layers (one pass over the 36-channel map each); S = 2 output classes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mdt_hip.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

inline int sg_check()
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

constexpr int SG_C = 36;           // channels of the big map
constexpr int SG_S = 2;            // channels of the small map

// ---- forward: y[v][s] = b[s] + sum_tap sum_ci x[v + tap][ci] * wt[tap][ci][s] -------------------------------------------------------------------
constexpr int SF_TY = 2, SF_TX = 4, SF_TZ = 16;
constexpr int SF_LY = SF_TY + 2, SF_LX = SF_TX + 2, SF_LZ = SF_TZ + 2;
constexpr int SF_VOX = SF_TY * SF_TX * SF_TZ;          // 128 voxels per workgroup
constexpr int SF_GROUPS = 3, SF_GC = SG_C / SF_GROUPS;  // three groups of 12 channels
constexpr int SF_VPT = 2;                               // voxels per thread (z and z + 8): the wave-uniform filter values are fetched once for both
constexpr int SF_TPG = SF_VOX / SF_VPT;                 // threads per channel group: 64 = one wave
constexpr int SF_THREADS = SF_TPG * SF_GROUPS;          // 192

__global__ __launch_bounds__(SF_THREADS) void conv_seg_fwd_kernel(float *__restrict__ y, const float *__restrict__ x, const float *__restrict__ wt,
                                                                  const float *__restrict__ bias, int Y, int X, int Z, int ty_n, int tx_n, int tz_n)
{
    __shared__ __attribute__((aligned(16))) float img[SF_LY * SF_LX * SF_LZ * SG_C];       // 62 208 bytes
    __shared__ float red[SF_GROUPS][SF_VOX][SG_S];
    int t = blockIdx.x;
    const int tz = t % tz_n; t /= tz_n;
    const int tx = t % tx_n; t /= tx_n;
    const int ty = t % ty_n;
    const int b = t / ty_n;
    const int y0 = ty * SF_TY, x0 = tx * SF_TX, z0 = tz * SF_TZ;
    // the tile's 4 x 6 input lines, z0 - 1 .. z0 + 16: contiguous runs of 18 * 36 floats (zero outside the map)
    constexpr int LQ = SF_LZ * SG_C / 4;                 // 16-byte pieces per line: 162
    for (int e = threadIdx.x; e < SF_LY * SF_LX * LQ; e += SF_THREADS) {
        const int line = e / LQ, piece = e - line * LQ;
        const int ly = line / SF_LX, lx = line - ly * SF_LX;
        const int yy = y0 + ly - 1, xx = x0 + lx - 1;
        const int zz = z0 - 1 + piece / (SG_C / 4);      // the voxel of this piece
        const bool ok = (unsigned)yy < (unsigned)Y && (unsigned)xx < (unsigned)X && (unsigned)zz < (unsigned)Z;
        v4f v = *reinterpret_cast<const v4f *>(x + ((((long long)b * Y + (ok ? yy : 0)) * X + (ok ? xx : 0)) * Z + (ok ? zz : 0)) * SG_C + (piece % (SG_C / 4)) * 4);
        if (!ok) v = v4f{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<v4f *>(img + (line * SF_LZ) * SG_C + piece * 4) = v;
    }
    __syncthreads();
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x / SF_TPG);      // wave-uniform: one wave per channel group
    const int tl = threadIdx.x - grp * SF_TPG;
    const int lz = tl % (SF_TZ / SF_VPT), lx = (tl / (SF_TZ / SF_VPT)) % SF_TX, ly = tl / ((SF_TZ / SF_VPT) * SF_TX);
    float acc[SF_VPT][SG_S];
#pragma unroll
    for (int u = 0; u < SF_VPT; ++u) { acc[u][0] = 0.0f; acc[u][1] = 0.0f; }
    const float *wg = wt + grp * SF_GC * SG_S;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
        const int dz = tap % 3, dx = (tap / 3) % 3, dy = tap / 9;              // (offsets + 1: the image starts one voxel before the tile)
        const float *wp = wg + tap * SG_C * SG_S;
#pragma unroll
        for (int u = 0; u < SF_VPT; ++u) {
            const float *src = img + (((ly + dy) * SF_LX + (lx + dx)) * SF_LZ + (lz + u * (SF_TZ / SF_VPT) + dz)) * SG_C + grp * SF_GC;
            const v4f a0 = *reinterpret_cast<const v4f *>(src), a1 = *reinterpret_cast<const v4f *>(src + 4), a2 = *reinterpret_cast<const v4f *>(src + 8);
            const float xs[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
#pragma unroll
            for (int c = 0; c < SF_GC; ++c) {
                acc[u][0] = acc[u][0] + xs[c] * wp[c * SG_S];
                acc[u][1] = acc[u][1] + xs[c] * wp[c * SG_S + 1];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < SF_VPT; ++u) {
        const int vl = (ly * SF_TX + lx) * SF_TZ + lz + u * (SF_TZ / SF_VPT);
        red[grp][vl][0] = acc[u][0];
        red[grp][vl][1] = acc[u][1];
    }
    __syncthreads();
    if (threadIdx.x < SF_VOX) {
        const int vl = threadIdx.x;
        const int vz = vl % SF_TZ, vx = (vl / SF_TZ) % SF_TX, vy = vl / (SF_TZ * SF_TX);
        const float r0 = ((red[0][vl][0] + red[1][vl][0]) + red[2][vl][0]) + (bias ? bias[0] : 0.0f);
        const float r1 = ((red[0][vl][1] + red[1][vl][1]) + red[2][vl][1]) + (bias ? bias[1] : 0.0f);
        const long long v = (((long long)b * Y + (y0 + vy)) * X + (x0 + vx)) * Z + (z0 + vz);
        *reinterpret_cast<float2 *>(y + v * SG_S) = float2{r0, r1};
    }
}

// ---- input gradient: gx[v][ci] = sum_tap sum_s g[v + tap][s] * wd[tap][s][ci]   (wd: the flipped filter) --------------------------------------
constexpr int SD_THREADS = 256;

__global__ __launch_bounds__(SD_THREADS) void conv_seg_dgrad_kernel(float *__restrict__ gx, const float *__restrict__ g, const float *__restrict__ wd,
                                                                    int Y, int X, int Z, long long V)
{
    __shared__ __attribute__((aligned(16))) float s_t[SD_THREADS * SG_C];       // 36 KB output tile
    const long long v0 = (long long)blockIdx.x * SD_THREADS;
    const long long v = v0 + threadIdx.x;
    if (v < V) {
        const int z = (int)(v % Z);
        const long long t = v / Z;
        const int xx = (int)(t % X);
        const int yy = (int)((t / X) % Y);
        float acc[SG_C];
#pragma unroll
        for (int c = 0; c < SG_C; ++c) acc[c] = 0.0f;
#pragma unroll 1                // (fully unrolled, the 1 944 wave-uniform filter values were all requested up front: 1 872 spilled scalar registers)
        for (int tap = 0; tap < 27; ++tap) {
            const int dz = tap % 3 - 1, dx = (tap / 3) % 3 - 1, dy = tap / 9 - 1;
            const bool ok = (unsigned)(yy + dy) < (unsigned)Y && (unsigned)(xx + dx) < (unsigned)X && (unsigned)(z + dz) < (unsigned)Z;
            float2 gv = *reinterpret_cast<const float2 *>(g + (ok ? v + ((long long)dy * X + dx) * Z + dz : v) * SG_S);
            if (!ok) gv = float2{0.f, 0.f};
            const float *wp = wd + tap * SG_S * SG_C;                           // wave-uniform addresses: scalar loads
#pragma unroll
            for (int c = 0; c < SG_C; ++c) acc[c] = (acc[c] + gv.x * wp[c]) + gv.y * wp[SG_C + c];
        }
#pragma unroll
        for (int c = 0; c < SG_C; ++c) s_t[threadIdx.x * SG_C + c] = acc[c];
    }
    __syncthreads();
    const long long nv = min((long long)SD_THREADS, V - v0);
    const int n4 = (int)(nv * SG_C / 4);                                        // 36 % 4 == 0: no tail
    v4f *dst = reinterpret_cast<v4f *>(gx + v0 * SG_C);
    for (int e = threadIdx.x; e < n4; e += SD_THREADS) dst[e] = reinterpret_cast<const v4f *>(s_t)[e];
}

// ---- weight gradient: D[(tap, s)][ci | ones] += g[v + tap][s] * x[v][ci] ---------------------------------------------------------------------------
// A wave walks chunks of 32 consecutive voxels of one z line (Z % 32 == 0); K is permuted inside a chunk: half-wave h of MFMA j multiplies voxel 16 h + j.
constexpr int SW_THREADS = 256;
constexpr int SW_BS = SG_C + 1;       // LDS row stride of the x tile (37: conflict-free fragment reads)

__global__ __launch_bounds__(SW_THREADS) void conv_seg_wgrad_kernel(float *__restrict__ partial, const float *__restrict__ g, const float *__restrict__ x,
                                                                    int Y, int X, int Z, long long chunks, long long chunks_per_wave)
{
    __shared__ float s_x[SW_THREADS / 64][32 * SW_BS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const long long gwv = (long long)blockIdx.x * (SW_THREADS / 64) + wave;
    const long long c0 = gwv * chunks_per_wave, c1 = min(chunks, c0 + chunks_per_wave);
    float *sx = s_x[wave];
    // A rows of the two row tiles: R = mt * 32 + col -> (tap R / 2, s R % 2) for R < 54
    int rdy[2], rdx[2], rdz[2], rs[2];
    bool rok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int R = mt * 32 + col;
        rok[mt] = R < 27 * SG_S;
        const int tap = rok[mt] ? R / SG_S : 13;
        rs[mt] = R % SG_S;
        rdz[mt] = tap % 3 - 1; rdx[mt] = (tap / 3) % 3 - 1; rdy[mt] = tap / 9 - 1;
    }
    const int zpc = Z / 32;
    v16f acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    for (long long c = c0; c < c1; ++c) {
        const long long line = c / zpc;
        const int zc = (int)(c - line * zpc) * 32;
        const int xx = (int)(line % X);
        const int yy = (int)((line / X) % Y);
        const long long vbase = line * Z + zc;
        // B: the chunk's x rows, 32 x 36 floats = 288 contiguous 16-byte pieces -> LDS [voxel][36 | 1]
        const v4f *x4 = reinterpret_cast<const v4f *>(x + vbase * SG_C);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int e4 = lane + 64 * i;
            if (e4 < 288) {
                const v4f xv = x4[e4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int e = e4 * 4 + j, vx = e / SG_C, ci = e - vx * SG_C; sx[vx * SW_BS + ci] = xv[j]; }
            }
        }
        // A: this lane's (tap, s) rows at voxels zc + 16 half .. + 15 of the shifted gradient
        float a[2][16];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const bool lok = rok[mt] && (unsigned)(yy + rdy[mt]) < (unsigned)Y && (unsigned)(xx + rdx[mt]) < (unsigned)X;
            const long long src = (line + (lok ? (long long)rdy[mt] * X + rdx[mt] : 0)) * Z;
            const int z0 = zc + 16 * half + rdz[mt];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int zz = z0 + j;
                const bool ok = lok && (unsigned)zz < (unsigned)Z;
                const float v = g[(src + (ok ? zz : 0)) * SG_S + rs[mt]];
                a[mt][j] = ok ? v : 0.0f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float b0 = sx[(16 * half + j) * SW_BS + col];                              // ci = col (< 32)
            const float b1 = col < SG_C - 32 ? sx[(16 * half + j) * SW_BS + 32 + col] : (col == SG_C - 32 ? 1.0f : 0.0f);    // ci = 32 + col (< 36), then the ones column
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][j], b0, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][j], b1, acc[mt][1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // D[R][ci] -> partial[wave][R (64)][ci (64)]
    float *pw = partial + gwv * 4096;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                pw[(mt * 32 + row) * 64 + nt * 32 + col] = acc[mt][nt][r];
            }
}

// gw[s][ci][tap] = sum over waves of partial[wave][tap * 2 + s][ci]; gbias[s] = the same sum at (centre tap, s), column 36.  One block per (R, ci | 36).
__global__ __launch_bounds__(256) void conv_seg_wgrad_fold_kernel(float *__restrict__ gw, float *__restrict__ gbias, const float *__restrict__ partial, long long nw)
{
    __shared__ float s_acc[256];
    const int R = blockIdx.x / (SG_C + 1), ci = blockIdx.x - R * (SG_C + 1);
    const int tap = R / SG_S, s = R - tap * SG_S;
    if (ci == SG_C && tap != 13) return;
    float sum = 0.0f;
    for (long long j = threadIdx.x; j < nw; j += 256) sum = sum + partial[j * 4096 + R * 64 + ci];
    s_acc[threadIdx.x] = sum;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) s_acc[threadIdx.x] = s_acc[threadIdx.x] + s_acc[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (ci < SG_C) gw[(s * SG_C + ci) * 27 + (26 - tap)] = s_acc[0];      // row `tap` shifted the gradient by +tap: that is the filter's mirrored tap
        else if (gbias) gbias[s] = s_acc[0];
    }
}

long long sg_waves(long long chunks)
{
    long long w = 256LL * 8;
    if (w > chunks) w = chunks;
    if (w < 1) w = 1;
    return ((w + 3) / 4) * 4;
}

}  // namespace

extern "C" {

int mdt_conv_seg_supported(int c_big, int c_small, int Y, int X, int Z)
{
    return (c_big == SG_C && c_small == SG_S && Y > 0 && X > 0 && Z > 0 && Y % SF_TY == 0 && X % SF_TX == 0 && Z % 32 == 0) ? 1 : 0;
}

int mdt_conv_seg_forward(const float *x, const float *wt, const float *bias, float *y, int batch, int Y, int X, int Z, int c_big, int c_small, void *stream)
{
    if (!x || !wt || !y || batch < 0) return MDT_ERR_INVALID_ARGUMENT;
    if (!mdt_conv_seg_supported(c_big, c_small, Y, X, Z)) return MDT_ERR_UNSUPPORTED;
    if (batch == 0) return MDT_OK;
    if ((((uintptr_t)x) & 15) != 0 || (((uintptr_t)y) & 7) != 0) return MDT_ERR_UNSUPPORTED;
    const long long blocks = (long long)batch * (Y / SF_TY) * (X / SF_TX) * (Z / SF_TZ);
    if (blocks > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    (void)hipGetLastError();
    hipLaunchKernelGGL(conv_seg_fwd_kernel, dim3((unsigned)blocks), dim3(SF_THREADS), 0, static_cast<hipStream_t>(stream), y, x, wt, bias, Y, X, Z, Y / SF_TY, X / SF_TX,
                       Z / SF_TZ);
    return sg_check();
}

int mdt_conv_seg_input_grad(const float *g, const float *wd, float *gx, int batch, int Y, int X, int Z, int c_big, int c_small, void *stream)
{
    if (!g || !wd || !gx || batch < 0) return MDT_ERR_INVALID_ARGUMENT;
    if (!mdt_conv_seg_supported(c_big, c_small, Y, X, Z)) return MDT_ERR_UNSUPPORTED;
    const long long V = (long long)batch * Y * X * Z;
    if (V == 0) return MDT_OK;
    if ((((uintptr_t)gx) & 15) != 0 || (((uintptr_t)g) & 7) != 0 || (V + SD_THREADS - 1) / SD_THREADS > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    (void)hipGetLastError();
    hipLaunchKernelGGL(conv_seg_dgrad_kernel, dim3((unsigned)((V + SD_THREADS - 1) / SD_THREADS)), dim3(SD_THREADS), 0, static_cast<hipStream_t>(stream), gx, g, wd, Y, X, Z,
                       V);
    return sg_check();
}

size_t mdt_conv_seg_wgrad_workspace_bytes(int batch, int Y, int X, int Z)
{
    if (batch <= 0 || Y <= 0 || X <= 0 || Z <= 0) return 256;
    return (size_t)sg_waves((long long)batch * Y * X * (Z / 32)) * 4096 * sizeof(float) + 256;
}

int mdt_conv_seg_weight_grad(const float *g, const float *x, float *grad_weight, float *grad_bias, int batch, int Y, int X, int Z, int c_big, int c_small,
                             void *workspace, size_t workspace_bytes, void *stream)
{
    if (!g || !x || !grad_weight || batch < 0) return MDT_ERR_INVALID_ARGUMENT;
    if (!mdt_conv_seg_supported(c_big, c_small, Y, X, Z)) return MDT_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < mdt_conv_seg_wgrad_workspace_bytes(batch, Y, X, Z)) return MDT_ERR_WORKSPACE_TOO_SMALL;
    if ((((uintptr_t)x) & 15) != 0) return MDT_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *partial = static_cast<float *>(workspace);
    const long long chunks = (long long)batch * Y * X * (Z / 32);
    const long long nw = chunks > 0 ? sg_waves(chunks) : 0;
    (void)hipGetLastError();
    if (nw > 0) {
        hipLaunchKernelGGL(conv_seg_wgrad_kernel, dim3((unsigned)(nw / 4)), dim3(SW_THREADS), 0, s, partial, g, x, Y, X, Z, chunks, (chunks + nw - 1) / nw);
        if (sg_check() != MDT_OK) return MDT_ERR_LAUNCH_FAILED;
    }
    hipLaunchKernelGGL(conv_seg_wgrad_fold_kernel, dim3(27 * SG_S * (SG_C + 1)), dim3(256), 0, s, grad_weight, grad_bias, partial, nw);
    return sg_check();
}

}  // extern "C"
