// conv_s221.hip -- around the Retina U-Net's C1 layer (backbone.py:84: 18 -> 18 channels, 7x7x7, stride (2, 2, 1), pad 3, on the
// full-resolution C0 output): 40 % of the config-2 training step was this ONE layer on MIOpen / CK (39.6 ms forward at 23 TF/s, 58.1 ms
// input gradient, 45.9 ms weight gradient at 8 x 128^3; tools/c1_probe.py).  What this file holds:
//
//   * space-to-depth plumbing: the 2 x 2 (y, x) phases of the zero-padded input as 4x the channels turn the layer into a 72 -> 18,
//     (4, 4, 7), unit-stride problem whose FORWARD MIOpen runs in 33.2 ms and whose input gradient is again a forward convolution
//     (18 -> 72: 26.2 ms instead of 58.1).  `s2d221_input_kernel` builds that input in one pass, `s2d221_fold_kernel` turns the
//     gradient of the space-to-depth input back into the gradient of x in one pass (torch's strided copy: 4.5 ms; 2.6 GB moved);
//
//   * the WEIGHT GRADIENT as an fp32-MFMA kernel.  In channels-last storage the (kz, ci) window of an output voxel,
//         x[b, iy, ix, z - 3 .. z + 3, 0 .. 17]  =  126 CONTIGUOUS floats starting at (z - 3) * 18 of the (iy, ix) column,
//     so for one filter tap pair (ky, kx)
//         dW[ky, kx][(kz, ci)][co] = sum over (b, oy, ox, z)  x_col(2 oy + ky - 3, 2 ox + kx - 3)[(z - 3) * 18 + (kz, ci)] * gy[b, oy, ox, z, co]
//     is a [126 x V] x [V x 18] product: M = the window (4 tiles of 32 rows, 98 % useful), N = co (18 of 32), K = the voxels, 2 per
//     v_mfma_f32_32x32x2_f32.  The operand layout of that instruction is one element per lane -- lane l <-> (row / column l & 31,
//     voxel l >> 5) -- so a wave's load of 32 consecutive floats of the column IS the fragment: global -> VGPR -> MFMA, no LDS, no
//     im2col.  Window rows that fall outside the column (the z padding) read as zero through the range check of buffer loads.  Rows >= 126 and
//     columns >= 18 of the tiles are never stored.
//     Work split: 49 tap pairs x 40 waves; a wave walks (output row, 32-column segment) units of its pair and keeps the pair's four
//     accumulator tiles (64 VGPRs) for the whole launch; operands are double-buffered in registers 16 voxels (32 MFMAs) ahead.
//     Every wave writes one partial, a second small kernel adds the 40 partials of a pair in a fixed order: deterministic, no atomics.
//     411 M MFMA issues = 10.7 ms at the fp32 MFMA peak (the padded problem is 1.65 TFLOP).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mdt_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

inline int s221_check()
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

// ---- space-to-depth plumbing ----------------------------------------------------------------------------------------------------------
// x  [B, Y, X, Z, C]  (channels-last storage of [B, C, Y, X, Z])
// xs [B, Y2, X2, Zp, 4C], channel (c, py, px):  xs[b, y2, x2, zp, c*4 + py*2 + px] = xpad[b, 2 y2 + py, 2 x2 + px, zp, c],  xpad = x padded by P
// one workgroup per (b, y2, x2) column; thread <-> (zp, c): four 4-byte loads that are contiguous across the threads, one 16-byte store
__global__ __launch_bounds__(256) void s2d221_input_kernel(const float *__restrict__ x, float *__restrict__ xs, int Y, int X, int Z, int C, int P,
                                                          int Y2, int X2)
{
    const int Zp = Z + 2 * P;
    long long col = blockIdx.x;
    const int x2 = (int)(col % X2);
    col /= X2;
    const int y2 = (int)(col % Y2);
    const long long b = col / Y2;
    const float *src[4];
    bool ok[4];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        const int iy = 2 * y2 + (ph >> 1) - P, ix = 2 * x2 + (ph & 1) - P;
        ok[ph] = iy >= 0 && iy < Y && ix >= 0 && ix < X;
        src[ph] = x + ((b * Y + (ok[ph] ? iy : 0)) * X + (ok[ph] ? ix : 0)) * (long long)Z * C;
    }
    v4f *dst = reinterpret_cast<v4f *>(xs + (long long)blockIdx.x * Zp * 4 * C);
    const int n = Zp * C;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int zp = i / C;
        const int z = zp - P;
        const bool zin = z >= 0 && z < Z;
        const int j = i - P * C;              // (zp - P) * C + c
        v4f v;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) v[ph] = (zin && ok[ph]) ? src[ph][j] : 0.0f;
        dst[i] = v;
    }
}

// gx[b, y, x, z, c] = gxs[b, y2, x2, z + P, c*4 + py*2 + px]  with  y = 2 y2 + py - P,  x = 2 x2 + px - P  (the padding rows are dropped)
__global__ __launch_bounds__(256) void s2d221_fold_kernel(const float *__restrict__ gxs, float *__restrict__ gx, int Y, int X, int Z, int C, int P,
                                                         int Y2, int X2)
{
    const int Zp = Z + 2 * P;
    long long col = blockIdx.x;
    const int x2 = (int)(col % X2);
    col /= X2;
    const int y2 = (int)(col % Y2);
    const long long b = col / Y2;
    float *dst[4];
    bool ok[4];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        const int iy = 2 * y2 + (ph >> 1) - P, ix = 2 * x2 + (ph & 1) - P;
        ok[ph] = iy >= 0 && iy < Y && ix >= 0 && ix < X;
        dst[ph] = gx + ((b * Y + (ok[ph] ? iy : 0)) * X + (ok[ph] ? ix : 0)) * (long long)Z * C;
    }
    const v4f *src = reinterpret_cast<const v4f *>(gxs + ((long long)blockIdx.x * Zp + P) * 4 * C);
    const int n = Z * C;
    for (int i = threadIdx.x; i < n; i += 256) {
        const v4f v = src[i];
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
            if (ok[ph]) dst[ph][i] = v[ph];
    }
}

// ---- weight gradient --------------------------------------------------------------------------------------------------------------------
constexpr int G_THREADS = 256;      // 4 waves
constexpr int G_UNROLL = 8;         // K-steps (of 2 voxels) per trip: 16 voxels, 8 * MT MFMAs; two trips' operands are live
constexpr int G_SEG = 32;           // output columns per unit

struct S221 {
    int B, Y, X, Z, Ci, Co, K, P, OY, OX;
    int waves_per_pair, nseg, units;        // units per pair = B * OY * nseg
};

template <int MT>
__global__ __launch_bounds__(G_THREADS, 2) void conv_s221_wgrad_kernel(const float *__restrict__ gy, const float *__restrict__ x,
                                                                      float *__restrict__ partial, S221 q)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = blockIdx.y, ky = pair / q.K, kx = pair - ky * q.K;
    const int wi = blockIdx.x * (G_THREADS / 64) + wave;
    const int c = lane & 31, kk = lane >> 5;
    const int ZCi = q.Z * q.Ci, ZCo = q.Z * q.Co;
    // byte offsets inside a column for the lane's element of a 2-voxel K-step (loop-invariant VGPRs).  The loads are BUFFER loads over
    // one column: an offset outside [0, Z * C * 4) -- the z padding of the window, negative offsets included (they wrap to huge unsigned
    // values) -- returns 0 from the hardware's range check, so the mask costs no instruction and, unlike a select on the loaded value,
    // does not serialise the loads (the first form of this kernel did: load, s_waitcnt vmcnt(0), v_cndmask, 40 times per trip)
    int offa[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) offa[m] = ((kk - q.P) * q.Ci + m * 32 + c) * 4;    // + (z0 + 2 u) * Ci * 4  =  ((z - P) * Ci + window row) * 4
    const int offb = (kk * q.Co + min(c, q.Co - 1)) * 4;                            // + (z0 + 2 u) * Co * 4
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    // output columns whose input column 2 ox + kx - P exists
    const int d = q.P - kx;                                     // ix = 2 ox - d
    const int ox_min = d > 0 ? (d + 1) / 2 : 0;
    int ox_max = (q.X - 1 + d) >= 0 ? (q.X - 1 + d) / 2 : -1;   // FLOOR: C division truncates towards zero, (-1) / 2 == 0 would admit ox = 0 with ix >= X (X < K)
    if (ox_max > q.OX - 1) ox_max = q.OX - 1;
    const int tpc = q.Z / (2 * G_UNROLL);                       // trips per column

    for (int unit = wi; unit < q.units; unit += q.waves_per_pair) {
        const int row = unit / q.nseg, seg = unit - row * q.nseg;
        const int b = row / q.OY, oy = row - b * q.OY;
        const int iy = 2 * oy + ky - q.P;
        if (iy < 0 || iy >= q.Y) continue;
        const int lo = max(ox_min, seg * G_SEG), hi = min(ox_max, seg * G_SEG + G_SEG - 1);
        if (lo > hi) continue;
        const float *xbase = x + (((long long)b * q.Y + iy) * q.X + (2 * lo - d)) * ZCi;            // next column: + 2 * ZCi
        const float *gbase = gy + (((long long)b * q.OY + oy) * q.OX + lo) * ZCo;                  // next column: + ZCo
        const int total = (hi - lo + 1) * tpc;

        auto load = [&](int col, int tr, float (&a)[G_UNROLL][MT], float (&bb)[G_UNROLL]) {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)(xbase + (long long)col * 2 * ZCi), 0, ZCi * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)(gbase + (long long)col * ZCo), 0, ZCo * 4, 0x00020000);
            const int j0 = tr * (2 * G_UNROLL) * q.Ci * 4, g0 = tr * (2 * G_UNROLL) * q.Co * 4;
#pragma unroll
            for (int u = 0; u < G_UNROLL; ++u) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    a[u][m] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, offa[m] + (j0 + 2 * u * q.Ci * 4), 0, 0));
                bb[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, offb + (g0 + 2 * u * q.Co * 4), 0, 0));
            }
        };
        auto mfma = [&](const float (&a)[G_UNROLL][MT], const float (&bb)[G_UNROLL]) {
#pragma unroll
            for (int u = 0; u < G_UNROLL; ++u)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][m], bb[u], acc[m], 0, 0, 0);
        };
        float a0[G_UNROLL][MT], b0[G_UNROLL], a1[G_UNROLL][MT], b1[G_UNROLL];
        // steady state without a conditional load: a join of "loaded" and "not loaded" paths in front of the MFMAs makes the compiler wait
        // for the NEWEST loads as well (s_waitcnt counts in issue order), which defeats the double buffering; the last one or two trips
        // are peeled instead
        int lcol = 0, ltr = 0, t = 0;
        auto next = [&]() { if (++ltr == tpc) { ltr = 0; ++lcol; } };
        load(lcol, ltr, a0, b0);
        next();
        for (; t + 2 < total; t += 2) {
            load(lcol, ltr, a1, b1);
            next();
            __builtin_amdgcn_sched_barrier(0);      // keep the 40 loads of the next trip IN FRONT of this trip's 32 MFMAs (a full trip of lookahead)
            mfma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            load(lcol, ltr, a0, b0);
            next();
            __builtin_amdgcn_sched_barrier(0);
            mfma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 2 == total) {
            load(lcol, ltr, a1, b1);
            mfma(a0, b0);
            mfma(a1, b1);
        } else {
            mfma(a0, b0);
        }
    }
    // one partial per wave: [K * Ci window rows][Co]
    const int rows = q.K * q.Ci;
    float *out = partial + ((long long)pair * q.waves_per_pair + wi) * rows * q.Co;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);          // C/D map of the 32x32 MFMA
            if (row < rows && c < q.Co) out[row * q.Co + c] = acc[m][r];
        }
}

// gw memory: [Co][ky][kx][kz][ci]  (= channels_last_3d storage of the [Co, Ci, K, K, K] weight gradient)
__global__ __launch_bounds__(256) void conv_s221_wgrad_finish_kernel(const float *__restrict__ partial, float *__restrict__ gw, int K, int Ci, int Co,
                                                                    int waves_per_pair)
{
    const int pair = blockIdx.y;
    const int rows = K * Ci, n = rows * Co;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const float *p = partial + (long long)pair * waves_per_pair * n + e;
    float s = 0.0f;
    int w = 0;
    for (; w + 8 <= waves_per_pair; w += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(w + u) * n];
#pragma unroll
        for (int u = 0; u < 8; ++u) s = s + v[u];
    }
    for (; w < waves_per_pair; ++w) s = s + p[(long long)w * n];
    const int m = e / Co, co = e - m * Co;
    gw[((long long)co * K * K + pair) * rows + m] = s;
}

bool wgrad_plan(int B, int Y, int X, int Z, int Ci, int Co, int K, S221 &q)
{
    if (B <= 0 || Y <= 0 || X <= 0 || Z <= 0 || Ci <= 0 || Co <= 0 || K < 3 || (K & 1) == 0) return false;
    if (K * Ci > 128 || Co > 32 || (Y & 1) || (X & 1) || Z % (2 * G_UNROLL) != 0) return false;
    if ((long long)Z * Ci * 4 >= (1LL << 30)) return false;
    q.B = B; q.Y = Y; q.X = X; q.Z = Z; q.Ci = Ci; q.Co = Co; q.K = K; q.P = K / 2;
    q.OY = Y / 2; q.OX = X / 2;                   // (Y + 2 P - K) / 2 + 1 with P = K / 2, Y even
    q.nseg = (q.OX + G_SEG - 1) / G_SEG;
    const long long units = (long long)B * q.OY * q.nseg;
    if (units > (1LL << 30)) return false;
    q.units = (int)units;
    int n = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    (void)hipGetLastError();
    int wpp = (n * 8) / (K * K);                  // two waves per SIMD over the whole chip
    wpp -= wpp % 4;
    if (wpp > ((q.units + 3) / 4) * 4) wpp = ((q.units + 3) / 4) * 4;
    if (wpp < 4) wpp = 4;
    q.waves_per_pair = wpp;
    return true;
}

}  // namespace

extern "C" {

int mdt_s2d221_input(const float *x, float *xs, int batch, int channels, int y, int x_, int z, int k, void *stream)
{
    if (!x || !xs || batch <= 0 || channels <= 0 || y <= 0 || x_ <= 0 || z <= 0 || k < 1 || (k & 1) == 0 || (y & 1) || (x_ & 1)) return MDT_ERR_INVALID_ARGUMENT;
    const int P = k / 2, Y2 = (y + 2 * P) / 2, X2 = (x_ + 2 * P) / 2;
    const long long cols = (long long)batch * Y2 * X2;
    if (cols > 0x7fffffffLL || (long long)(z + 2 * P) * channels > 0x3fffffffLL) return MDT_ERR_UNSUPPORTED;
    (void)hipGetLastError();
    hipLaunchKernelGGL(s2d221_input_kernel, dim3((unsigned)cols), dim3(256), 0, static_cast<hipStream_t>(stream), x, xs, y, x_, z, channels, P, Y2, X2);
    return s221_check();
}

int mdt_s2d221_fold_input_grad(const float *gxs, float *gx, int batch, int channels, int y, int x_, int z, int k, void *stream)
{
    if (!gxs || !gx || batch <= 0 || channels <= 0 || y <= 0 || x_ <= 0 || z <= 0 || k < 1 || (k & 1) == 0 || (y & 1) || (x_ & 1)) return MDT_ERR_INVALID_ARGUMENT;
    const int P = k / 2, Y2 = (y + 2 * P) / 2, X2 = (x_ + 2 * P) / 2;
    const long long cols = (long long)batch * Y2 * X2;
    if (cols > 0x7fffffffLL || (long long)(z + 2 * P) * channels > 0x3fffffffLL) return MDT_ERR_UNSUPPORTED;
    (void)hipGetLastError();
    hipLaunchKernelGGL(s2d221_fold_kernel, dim3((unsigned)cols), dim3(256), 0, static_cast<hipStream_t>(stream), gxs, gx, y, x_, z, channels, P, Y2, X2);
    return s221_check();
}

int mdt_conv_s221_wgrad_supported(int batch, int y, int x_, int z, int c_in, int c_out, int k)
{
    S221 q;
    return wgrad_plan(batch, y, x_, z, c_in, c_out, k, q) ? 1 : 0;
}

size_t mdt_conv_s221_wgrad_workspace_bytes(int batch, int y, int x_, int z, int c_in, int c_out, int k)
{
    S221 q;
    if (!wgrad_plan(batch, y, x_, z, c_in, c_out, k, q)) return 0;
    return (size_t)k * k * q.waves_per_pair * k * c_in * c_out * sizeof(float) + 256;
}

int mdt_conv_s221_wgrad(const float *grad_out, const float *x, float *grad_weight, int batch, int y, int x_, int z, int c_in, int c_out, int k,
                        void *workspace, size_t workspace_bytes, void *stream)
{
    S221 q;
    if (!grad_out || !x || !grad_weight) return MDT_ERR_INVALID_ARGUMENT;
    if (!wgrad_plan(batch, y, x_, z, c_in, c_out, k, q)) return MDT_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < (size_t)k * k * q.waves_per_pair * k * c_in * c_out * sizeof(float)) return MDT_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *ws = static_cast<float *>(workspace);
    const dim3 grid((unsigned)(q.waves_per_pair / 4), (unsigned)(k * k));
    (void)hipGetLastError();
    switch ((k * c_in + 31) / 32) {
    case 1: hipLaunchKernelGGL(conv_s221_wgrad_kernel<1>, grid, dim3(G_THREADS), 0, s, grad_out, x, ws, q); break;
    case 2: hipLaunchKernelGGL(conv_s221_wgrad_kernel<2>, grid, dim3(G_THREADS), 0, s, grad_out, x, ws, q); break;
    case 3: hipLaunchKernelGGL(conv_s221_wgrad_kernel<3>, grid, dim3(G_THREADS), 0, s, grad_out, x, ws, q); break;
    default: hipLaunchKernelGGL(conv_s221_wgrad_kernel<4>, grid, dim3(G_THREADS), 0, s, grad_out, x, ws, q); break;
    }
    if (s221_check() != MDT_OK) return MDT_ERR_LAUNCH_FAILED;
    const int n = k * c_in * c_out;
    hipLaunchKernelGGL(conv_s221_wgrad_finish_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)(k * k)), dim3(256), 0, s, ws, grad_weight, k, c_in, c_out,
                       q.waves_per_pair);
    return s221_check();
}

}  // extern "C"
