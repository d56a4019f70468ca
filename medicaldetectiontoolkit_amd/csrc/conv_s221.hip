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
    int S;                                  // (y, x) stride: 2, or 1 (round 6: the size-preserving few-channel layers, mdt_conv_win_wgrad)
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
    const int d = q.P - kx;                                     // ix = S ox - d
    const int ox_min = d > 0 ? (d + q.S - 1) / q.S : 0;
    int ox_max = (q.X - 1 + d) >= 0 ? (q.X - 1 + d) / q.S : -1;   // FLOOR: C division truncates towards zero, (-1) / 2 == 0 would admit ox = 0 with ix >= X (X < K)
    if (ox_max > q.OX - 1) ox_max = q.OX - 1;
    const int tpc = q.Z / (2 * G_UNROLL);                       // trips per column

    for (int unit = wi; unit < q.units; unit += q.waves_per_pair) {
        const int row = unit / q.nseg, seg = unit - row * q.nseg;
        const int b = row / q.OY, oy = row - b * q.OY;
        const int iy = q.S * oy + ky - q.P;
        if (iy < 0 || iy >= q.Y) continue;
        const int lo = max(ox_min, seg * G_SEG), hi = min(ox_max, seg * G_SEG + G_SEG - 1);
        if (lo > hi) continue;
        const float *xbase = x + (((long long)b * q.Y + iy) * q.X + (q.S * lo - d)) * ZCi;          // next column: + S * ZCi
        const float *gbase = gy + (((long long)b * q.OY + oy) * q.OX + lo) * ZCo;                  // next column: + ZCo
        const int total = (hi - lo + 1) * tpc;

        auto load = [&](int col, int tr, float (&a)[G_UNROLL][MT], float (&bb)[G_UNROLL]) {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)(xbase + (long long)col * q.S * ZCi), 0, ZCi * 4, 0x00020000);
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

bool wgrad_plan(int B, int Y, int X, int Z, int Ci, int Co, int K, S221 &q, int S = 2)
{
    if (B <= 0 || Y <= 0 || X <= 0 || Z <= 0 || Ci <= 0 || Co <= 0 || K < 3 || (K & 1) == 0) return false;
    if (K * Ci > 128 || Co > 32 || (Y % S) || (X % S) || Z % (2 * G_UNROLL) != 0) return false;
    if ((long long)Z * Ci * 4 >= (1LL << 30)) return false;
    q.B = B; q.Y = Y; q.X = X; q.Z = Z; q.Ci = Ci; q.Co = Co; q.K = K; q.P = K / 2; q.S = S;
    q.OY = Y / S; q.OX = X / S;                   // (Y + 2 P - K) / S + 1 with P = K / 2, S | Y
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

// ---- forward (round 6) ------------------------------------------------------------------------------------------------------------------
// The same window trick with the roles turned: for one filter tap pair (ky, kx) and one output column (b, oy, ox)
//     y[z][co] += sum over the window  x_col(2 oy + ky - P, 2 ox + kx - P)[(z - P) * Ci + k] * Wt[ky][kx][k][co],   k = (kz, ci) in [0, K * Ci)
// is a [32 z x K Ci] x [K Ci x Co] product per 32 consecutive z: M = z, N = co (18 of 32), K = the window, 2 per v_mfma_f32_32x32x2_f32.
// The A operand is a TOEPLITZ view of the input column -- row m is the column shifted by m * Ci floats -- so it is read straight out of an LDS
// image of the column (lane l: element (m = l & 31) * Ci + 2 ks + (l >> 5); 18-float row stride = conflict-free over 64 banks): no im2col.
// A workgroup of 4 waves owns 4 consecutive output columns ox of one (b, oy) and 64 z: per ky it stages the 2 * 4 + K - 2 input columns its
// waves need (z halo of K - 1, zeros outside the volume) -- 65.5 KB, two workgroups per CU, so one stages while the other multiplies --
// and every wave runs its K pairs x (K Ci / 2) K-steps x 2 M tiles with the pair's B fragments (K Ci / 2 registers) loaded from the
// [ky][kx][k][co] filter image (444 KB: L2-resident).  Accumulators (2 tiles x 16 registers) live across all K x K pairs; bias and
// ReLU ride in the epilogue.  405 M MFMA issues at 8 x 128^3 = 10.5 ms at the fp32 MFMA peak (MIOpen / CK on the space-to-depth problem: 33.4 ms).
constexpr int F_WOX = 4;            // output columns (= waves) per workgroup
constexpr int F_ZH = 64;            // output z per workgroup (2 M tiles)
constexpr int F_MAXKS = 64;         // K-steps per pair (K * Ci / 2 <= 64)

struct S221F {
    int B, Y, X, Z, Ci, Co, K, P, OY, OX;
    int KC, ksteps, ZR, ncol, oxg;      // K * Ci, KC / 2, F_ZH + K - 1, S * (F_WOX - 1) + K, OX / F_WOX
    int relu;
    int S;                              // (y, x) stride of the forward kernel: 2 (the layer this file is named after) or 1 (round 6: the size-preserving few-channel layers)
};

// KSC > 0: the K-step count as a compile-time constant (63 for the 18-channel 7x7x7 layer): a straight-line block of KSC B loads and 2 KSC LDS reads /
// MFMAs per pair; KSC == 0: run-time count (other shapes; the compiler then guards every K-step with a branch)
template <int KSC, int KKC>
__global__ __launch_bounds__(F_WOX * 64, 2) void conv_s221_fwd_kernel(const float *__restrict__ x, const float *__restrict__ wt, const float *__restrict__ bias,
                                                                      float *__restrict__ y, S221F q)
{
    extern __shared__ __attribute__((aligned(16))) float sA[];          // [ncol][ZR * Ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int oxb = (blockIdx.x % q.oxg) * F_WOX, zh0 = (blockIdx.x / q.oxg) * F_ZH, oy = blockIdx.y, b = blockIdx.z;
    const int colf = q.ZR * q.Ci;                                       // floats per staged column
    const int cn = min(col, q.Co - 1);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    for (int ky = 0; ky < q.K; ++ky) {
        const int iy = q.S * oy + ky - q.P;
        if (iy < 0 || iy >= q.Y) continue;                              // uniform over the workgroup: a zero-padding row contributes nothing
        __syncthreads();                                                // everyone is done with the previous image
        {   // stage: ncol columns x ZR rows x Ci floats, float2 at a time (Ci even: a pair never straddles a z row)
            const int pairs_per_col = colf >> 1;
            const long long rowbase = ((long long)b * q.Y + iy) * q.X;
            for (int e = tid; e < q.ncol * pairs_per_col; e += F_WOX * 64) {
                const int c = e / pairs_per_col, j = (e - c * pairs_per_col) * 2;
                const int ix = q.S * oxb - q.P + c;
                const int z = zh0 - q.P + j / q.Ci;
                float2 v = make_float2(0.0f, 0.0f);
                if (ix >= 0 && ix < q.X && z >= 0 && z < q.Z)
                    v = *reinterpret_cast<const float2 *>(x + (rowbase + ix) * (long long)q.Z * q.Ci + (long long)(zh0 - q.P) * q.Ci + j);
                *reinterpret_cast<float2 *>(sA + c * colf + j) = v;
            }
        }
        __syncthreads();
        if (KSC > 0) {
            // the pair's B fragments (KSC registers) are loaded ONE PAIR AHEAD: the loads of pair kx + 1 are issued before the MFMA block of pair kx
            // (sched_barrier keeps the compiler from sinking them to their uses -- it did, with a vmcnt(0) in front of every second MFMA)
            constexpr int KS1 = KSC > 0 ? KSC : 1;
            const int co2 = 2 * q.Co, hi = 32 * q.Ci;
            float bf[2][KS1];
            {
                const float *Bp = wt + ((long long)(ky * q.K) * q.KC + half) * q.Co + cn;
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) bf[0][ks] = Bp[ks * co2];
            }
#pragma unroll
            for (int kx = 0; kx < KKC; ++kx) {
                if (kx + 1 < KKC) {
                    const float *Bn = wt + ((long long)(ky * q.K + kx + 1) * q.KC + half) * q.Co + cn;
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) bf[(kx + 1) & 1][ks] = Bn[ks * co2];
                }
                __builtin_amdgcn_sched_barrier(0);
                const float *Ac = sA + (q.S * wave + kx) * colf + col * q.Ci + half;
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    const float a0 = Ac[2 * ks];
                    const float a1 = Ac[hi + 2 * ks];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bf[kx & 1][ks], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bf[kx & 1][ks], acc1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int kx = 0; kx < q.K; ++kx) {
                const float *Ac = sA + (q.S * wave + kx) * colf + col * q.Ci + half;
                const float *Bp = wt + ((long long)(ky * q.K + kx) * q.KC + half) * q.Co + cn;
                for (int ks = 0; ks < q.ksteps; ++ks) {
                    const float bv = Bp[(long long)2 * ks * q.Co];
                    const float a0 = Ac[2 * ks];
                    const float a1 = Ac[32 * q.Ci + 2 * ks];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc1, 0, 0, 0);
                }
            }
        }
    }
    // epilogue: (+ bias)(ReLU), rows = z, columns = co; lane (col, half) holds rows (r & 3) + 8 (r >> 2) + 4 half
    if (col < q.Co) {
        const float bv = bias ? bias[col] : 0.0f;
        float *yo = y + ((((long long)b * q.OY + oy) * q.OX + oxb + wave) * q.Z + zh0) * q.Co + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            float v0 = acc0[r] + bv, v1 = acc1[r] + bv;
            if (q.relu) { v0 = v0 > 0.0f ? v0 : 0.0f; v1 = v1 > 0.0f ? v1 : 0.0f; }
            yo[(long long)row * q.Co] = v0;
            yo[(long long)(32 + row) * q.Co] = v1;
        }
    }
}

// ---- input gradient (round 6): the same machine on the output gradient --------------------------------------------------------------------
//     gx[b, iy, ix, z, ci] = sum over the taps (ky, kx) with ky = iy + P, kx = ix + P (mod 2) of
//                            sum over the window  gy_col((iy + P - ky) / 2, (ix + P - kx) / 2)[(z - P) * Co + (j, co)] * Wd[ky][kx][(j, co)][ci],   j = K - 1 - kz
// i.e. the forward kernel with gy as the input, (K + 1) / 2 or (K - 1) / 2 taps per axis depending on the parity of the output voxel, and the filter
// image Wd = w.flip(kz).permute(ky, kx, kz, co, ci).  A workgroup owns 4 consecutive ix of one (b, iy) and 64 z: its waves' taps read
// (K + 1) / 2 + 2 source columns in all (6 for K = 7).  No space-to-depth problem, no padded copy of gy, no fold: MIOpen's forward convolution on the padded
// output gradient takes 26.4 ms + 0.5 ms fold at 8 x 128^3.
template <int KSC, int KTC>
__global__ __launch_bounds__(F_WOX * 64, 2) void conv_s221_dgrad_kernel(const float *__restrict__ gy, const float *__restrict__ wd, float *__restrict__ gx, S221F q)
{
    extern __shared__ __attribute__((aligned(16))) float sA[];          // [ncol][ZR * Co]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int ixg = q.X / F_WOX;
    const int ix0 = (blockIdx.x % ixg) * F_WOX, zh0 = (blockIdx.x / ixg) * F_ZH, iy = blockIdx.y, b = blockIdx.z;
    const int CW = q.Co, CN = q.Ci;                                     // window channels (the source's), output channels
    const int colf = q.ZR * CW;
    const int cn = min(col, CN - 1);
    const int ncol = (q.K + 1) / 2 + 2;
    const int ox_base = (ix0 + q.P - (q.K - 1)) >> 1;                   // arithmetic shift = floor, also for negative values
    const int ix = ix0 + wave;
    const int kx0 = (ix + q.P) & 1, nkx = (q.K - kx0 + 1) >> 1;         // this wave's taps: kx = kx0 + 2 t
    const int ky0 = (iy + q.P) & 1, nky = (q.K - ky0 + 1) >> 1;
    const int KCW = q.K * CW, co2 = 2 * CN, hi = 32 * CW;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    for (int ty = 0; ty < nky; ++ty) {
        const int ky = ky0 + 2 * ty;
        const int oy = (iy + q.P - ky) >> 1;
        if (oy < 0 || oy >= q.OY) continue;
        __syncthreads();
        {
            const int pairs_per_col = colf >> 1;
            const long long rowbase = ((long long)b * q.OY + oy) * q.OX;
            for (int e = tid; e < ncol * pairs_per_col; e += F_WOX * 64) {
                const int c = e / pairs_per_col, j = (e - c * pairs_per_col) * 2;
                const int ox = ox_base + c;
                const int z = zh0 - q.P + j / CW;
                float2 v = make_float2(0.0f, 0.0f);
                if (ox >= 0 && ox < q.OX && z >= 0 && z < q.Z)
                    v = *reinterpret_cast<const float2 *>(gy + (rowbase + ox) * (long long)q.Z * CW + (long long)(zh0 - q.P) * CW + j);
                *reinterpret_cast<float2 *>(sA + c * colf + j) = v;
            }
        }
        __syncthreads();
        constexpr int KS1 = KSC > 0 ? KSC : 1;
        if (KSC > 0) {
            float bf[2][KS1];
            {
                const float *Bp = wd + ((long long)(ky * q.K + kx0) * KCW + half) * CN + cn;
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) bf[0][ks] = Bp[ks * co2];
            }
#pragma unroll
            for (int t = 0; t < KTC; ++t) {
                if (t < nkx) {                                          // wave-uniform
                    if (t + 1 < nkx) {
                        const float *Bn = wd + ((long long)(ky * q.K + kx0 + 2 * (t + 1)) * KCW + half) * CN + cn;
#pragma unroll
                        for (int ks = 0; ks < KS1; ++ks) bf[(t + 1) & 1][ks] = Bn[ks * co2];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const int c = ((ix + q.P - (kx0 + 2 * t)) >> 1) - ox_base;
                    const float *Ac = sA + c * colf + col * CW + half;
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) {
                        const float a0 = Ac[2 * ks];
                        const float a1 = Ac[hi + 2 * ks];
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bf[t & 1][ks], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bf[t & 1][ks], acc1, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            for (int t = 0; t < nkx; ++t) {
                const int kx = kx0 + 2 * t;
                const int c = ((ix + q.P - kx) >> 1) - ox_base;
                const float *Ac = sA + c * colf + col * CW + half;
                const float *Bp = wd + ((long long)(ky * q.K + kx) * KCW + half) * CN + cn;
                for (int ks = 0; ks < (KCW >> 1); ++ks) {
                    const float bv = Bp[(long long)ks * co2];
                    const float a0 = Ac[2 * ks];
                    const float a1 = Ac[hi + 2 * ks];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc1, 0, 0, 0);
                }
            }
        }
    }
    if (col < CN) {
        float *go = gx + ((((long long)b * q.Y + iy) * q.X + ix) * q.Z + zh0) * CN + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            go[(long long)row * CN] = acc0[r];
            go[(long long)(32 + row) * CN] = acc1[r];
        }
    }
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

static int win_wgrad(const float *grad_out, const float *x, float *grad_weight, int batch, int y, int x_, int z, int c_in, int c_out, int k, int S,
                     void *workspace, size_t workspace_bytes, void *stream);

int mdt_conv_s221_wgrad(const float *grad_out, const float *x, float *grad_weight, int batch, int y, int x_, int z, int c_in, int c_out, int k,
                        void *workspace, size_t workspace_bytes, void *stream)
{
    return win_wgrad(grad_out, x, grad_weight, batch, y, x_, z, c_in, c_out, k, 2, workspace, workspace_bytes, stream);
}

/* the weight-gradient kernel at UNIT stride (round 6): gw [c_out][ky][kx][kz][ci] of the size-preserving k x k x k, pad k / 2 convolution */
int mdt_conv_win_wgrad_supported(int batch, int y, int x_, int z, int c_in, int c_out, int k)
{
    S221 q;
    return wgrad_plan(batch, y, x_, z, c_in, c_out, k, q, 1) ? 1 : 0;
}

size_t mdt_conv_win_wgrad_workspace_bytes(int batch, int y, int x_, int z, int c_in, int c_out, int k)
{
    S221 q;
    if (!wgrad_plan(batch, y, x_, z, c_in, c_out, k, q, 1)) return 0;
    return (size_t)k * k * q.waves_per_pair * k * c_in * c_out * sizeof(float) + 256;
}

int mdt_conv_win_wgrad(const float *grad_out, const float *x, float *grad_weight, int batch, int y, int x_, int z, int c_in, int c_out, int k,
                       void *workspace, size_t workspace_bytes, void *stream)
{
    return win_wgrad(grad_out, x, grad_weight, batch, y, x_, z, c_in, c_out, k, 1, workspace, workspace_bytes, stream);
}

static int win_wgrad(const float *grad_out, const float *x, float *grad_weight, int batch, int y, int x_, int z, int c_in, int c_out, int k, int S,
                     void *workspace, size_t workspace_bytes, void *stream)
{
    S221 q;
    if (!grad_out || !x || !grad_weight) return MDT_ERR_INVALID_ARGUMENT;
    if (!wgrad_plan(batch, y, x_, z, c_in, c_out, k, q, S)) return MDT_ERR_UNSUPPORTED;
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

static int win_forward_supported(int Y, int X, int Z, int c_in, int c_out, int k, int S)
{
    if (k < 3 || (k & 1) == 0 || c_in < 1 || c_out < 1 || c_out > 32 || ((k * c_in) & 1) || (c_in & 1) || k * c_in / 2 > F_MAXKS) return 0;
    if (Y < S || X < S || (Y % S) || (X % S) || Z < F_ZH || Z % F_ZH) return 0;
    if ((X / S) % F_WOX) return 0;
    const size_t lds = (size_t)(S * (F_WOX - 1) + k) * (F_ZH + k - 1) * c_in * sizeof(float);
    return lds <= 80 * 1024 ? 1 : 0;
}

static int win_forward(const float *x, const float *wt, const float *bias, int relu, float *y, int batch, int Y, int X, int Z, int c_in, int c_out, int k, int S,
                       void *stream)
{
    if (!x || !wt || !y || batch < 0) return MDT_ERR_INVALID_ARGUMENT;
    if (!win_forward_supported(Y, X, Z, c_in, c_out, k, S)) return MDT_ERR_UNSUPPORTED;
    if (batch == 0) return MDT_OK;
    if ((((uintptr_t)x) & 7) != 0) return MDT_ERR_UNSUPPORTED;
    S221F q;
    q.B = batch; q.Y = Y; q.X = X; q.Z = Z; q.Ci = c_in; q.Co = c_out; q.K = k; q.P = k / 2; q.OY = Y / S; q.OX = X / S; q.S = S;
    q.KC = k * c_in; q.ksteps = q.KC / 2; q.ZR = F_ZH + k - 1; q.ncol = S * (F_WOX - 1) + k; q.oxg = q.OX / F_WOX; q.relu = relu ? 1 : 0;
    const size_t lds = (size_t)q.ncol * q.ZR * c_in * sizeof(float);
    static bool optin = false;
    if (!optin) {
        (void)hipFuncSetAttribute((const void *)conv_s221_fwd_kernel<63, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void *)conv_s221_fwd_kernel<27, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void *)conv_s221_fwd_kernel<54, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void *)conv_s221_fwd_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipGetLastError();
        optin = true;
    }
    (void)hipGetLastError();
    const dim3 grid(q.oxg * (Z / F_ZH), q.OY, batch), block(F_WOX * 64);
    if (q.ksteps == 63 && k == 7) hipLaunchKernelGGL((conv_s221_fwd_kernel<63, 7>), grid, block, lds, (hipStream_t)stream, x, wt, bias, y, q);
    else if (q.ksteps == 27 && k == 3) hipLaunchKernelGGL((conv_s221_fwd_kernel<27, 3>), grid, block, lds, (hipStream_t)stream, x, wt, bias, y, q);
    else if (q.ksteps == 54 && k == 3) hipLaunchKernelGGL((conv_s221_fwd_kernel<54, 3>), grid, block, lds, (hipStream_t)stream, x, wt, bias, y, q);
    else hipLaunchKernelGGL((conv_s221_fwd_kernel<0, 1>), grid, block, lds, (hipStream_t)stream, x, wt, bias, y, q);
    return s221_check();
}

int mdt_conv_s221_forward_supported(int Y, int X, int Z, int c_in, int c_out, int k)
{
    return win_forward_supported(Y, X, Z, c_in, c_out, k, 2);
}

/* y [B, Y/2, X/2, Z, c_out] (channels-last storage of [B, c_out, Y/2, X/2, Z]) = conv(x [B, Y, X, Z, c_in] channels-last, w, k x k x k, stride (2, 2, 1), pad k / 2)
 * (+ bias)(ReLU); wt = the filter as [ky][kx][kz][ci][co] (w.permute(2, 3, 4, 1, 0) contiguous). */
int mdt_conv_s221_forward(const float *x, const float *wt, const float *bias, int relu, float *y, int batch, int Y, int X, int Z, int c_in, int c_out, int k,
                          void *stream)
{
    return win_forward(x, wt, bias, relu, y, batch, Y, X, Z, c_in, c_out, k, 2, stream);
}

/* the same kernel at unit stride (round 6): the size-preserving k x k x k, pad k / 2 convolution of a channels-last activation with few channels */
int mdt_conv_win_forward_supported(int Y, int X, int Z, int c_in, int c_out, int k)
{
    return win_forward_supported(Y, X, Z, c_in, c_out, k, 1);
}

int mdt_conv_win_forward(const float *x, const float *wt, const float *bias, int relu, float *y, int batch, int Y, int X, int Z, int c_in, int c_out, int k,
                         void *stream)
{
    return win_forward(x, wt, bias, relu, y, batch, Y, X, Z, c_in, c_out, k, 1, stream);
}

int mdt_conv_s221_input_grad_supported(int Y, int X, int Z, int c_in, int c_out, int k)
{
    if (k < 3 || (k & 1) == 0 || c_in < 1 || c_in > 32 || c_out < 2 || (c_out & 1) || k * c_out / 2 > F_MAXKS) return 0;
    if (Y < 2 || X < 2 || (Y & 1) || (X & 1) || Z < F_ZH || Z % F_ZH) return 0;
    if (X % F_WOX) return 0;
    const size_t lds = (size_t)((k + 1) / 2 + 2) * (F_ZH + k - 1) * c_out * sizeof(float);
    return lds <= 80 * 1024 ? 1 : 0;
}

/* gx [B, Y, X, Z, c_in] (channels-last storage of [B, c_in, Y, X, Z]) = the input gradient of conv3d(x, w, stride (2, 2, 1), pad k / 2) for gy [B, Y/2, X/2, Z, c_out]
 * channels-last; wd = the filter as [ky][kx][K-1-kz][co][ci] (w.flip(4).permute(2, 3, 4, 0, 1) contiguous).  Every element of gx is written. */
int mdt_conv_s221_input_grad(const float *gy, const float *wd, float *gx, int batch, int Y, int X, int Z, int c_in, int c_out, int k, void *stream)
{
    if (!gy || !wd || !gx || batch < 0) return MDT_ERR_INVALID_ARGUMENT;
    if (!mdt_conv_s221_input_grad_supported(Y, X, Z, c_in, c_out, k)) return MDT_ERR_UNSUPPORTED;
    if (batch == 0) return MDT_OK;
    if ((((uintptr_t)gy) & 7) != 0) return MDT_ERR_UNSUPPORTED;
    S221F q;
    q.B = batch; q.Y = Y; q.X = X; q.Z = Z; q.Ci = c_in; q.Co = c_out; q.K = k; q.P = k / 2; q.OY = Y / 2; q.OX = X / 2;
    q.KC = k * c_out; q.ksteps = q.KC / 2; q.ZR = F_ZH + k - 1; q.ncol = (k + 1) / 2 + 2; q.oxg = X / F_WOX; q.relu = 0; q.S = 2;
    const size_t lds = (size_t)q.ncol * q.ZR * c_out * sizeof(float);
    static bool optin = false;
    if (!optin) {
        (void)hipFuncSetAttribute((const void *)conv_s221_dgrad_kernel<63, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void *)conv_s221_dgrad_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipGetLastError();
        optin = true;
    }
    (void)hipGetLastError();
    const dim3 grid((X / F_WOX) * (Z / F_ZH), Y, batch), block(F_WOX * 64);
    if (q.ksteps == 63 && k == 7) hipLaunchKernelGGL((conv_s221_dgrad_kernel<63, 4>), grid, block, lds, (hipStream_t)stream, gy, wd, gx, q);
    else hipLaunchKernelGGL((conv_s221_dgrad_kernel<0, 1>), grid, block, lds, (hipStream_t)stream, gy, wd, gx, q);
    return s221_check();
}

}  // extern "C"
