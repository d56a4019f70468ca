// pool.hip -- channels-last helpers around the MIOpen convolutions of the ResNet stem / backward pass (gfx950).
//
// (1) Max pooling of the stem (reference graph: models/backbone.py:77-79, MaxPool3d(kernel 3, stride (2, 2, 1), pad 1) in
//     front of C2).  torch's max_pool3d has no channels-last kernel: on the channels_last_3d activations of this model it
//     costs a full layout copy in, an NCDHW kernel (0.55 ms forward, 0.86 ms backward with atomics at 8 x 18 x 64 x 64 x 128)
//     and a transpose back in front of the next convolution.  Here: one thread per (voxel, channel) on the channels-last
//     storage (consecutive lanes = consecutive channels, then z: fully coalesced), the 27 window reads are served by
//     L1/L2; the arg-max tap (0..26) is kept as one byte per output; the backward is a GATHER (one thread per 2 x 2 input
//     block over the 12 windows that can contain it) -- no atomics, fixed summation order, deterministic.  Tie / NaN rule of torch
//     (first maximum in (y, x, z) scan order, NaN wins) is reproduced.
//     Algorithmic bytes: forward 4*V_in + 5*V_out, backward 5*V_out + 4*V_in  (V = B*C*voxels).  HBM-bound, no MFMA.
// (2) mdt_filter_flip_transpose: w[co][ci][taps] -> w'[ci][co][reversed taps] in one launch (either memory order); feeds the
//     "input gradient as a forward convolution" path (utils/fused_epilogue._ConvStride1).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mdt_hip.h"

namespace {

constexpr int PL_THREADS = 256;

inline int pl_check()
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

// grid: x covers one output row segment (ox, z, c) with 32-bit index math, y walks the (b, oy) rows
__global__ __launch_bounds__(PL_THREADS) void maxpool_k3s221_cl_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                                            unsigned char *__restrict__ arg, int rows,
                                                                            int Y, int X, int Z, int C, int OY, int OX)
{
    const unsigned row_len = (unsigned)OX * Z * C;
    const unsigned j = blockIdx.x * PL_THREADS + threadIdx.x;
    if (j >= row_len) return;
    const unsigned c = j % (unsigned)C;
    const unsigned t = j / (unsigned)C;
    const int z = (int)(t % (unsigned)Z);
    const int ox = (int)(t / (unsigned)Z);
    const unsigned zc = (unsigned)Z * C;
    const int x0 = 2 * ox - 1;
    const bool x_in = x0 >= 0 && x0 + 2 < X, z_in = z >= 1 && z + 1 < Z;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        const int oy = r % OY;
        const long long b = r / OY;
        const float *xb = x + b * (long long)Y * X * zc + c;
        const int y0 = 2 * oy - 1;
        float best = -__builtin_inff();
        int best_tap = -1;
        if (x_in && z_in && y0 >= 0 && y0 + 2 < Y) {          // interior: 27 unguarded loads
            const float *p = xb + ((long long)y0 * X + x0) * zc + (unsigned)(z - 1) * C;
            float v[27];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int dz = 0; dz < 3; ++dz) v[dy * 9 + dx * 3 + dz] = p[((long long)dy * X + dx) * zc + (unsigned)dz * C];
#pragma unroll
            for (int k = 0; k < 27; ++k)
                if (best_tap < 0 || v[k] > best || v[k] != v[k]) { best = v[k]; best_tap = k; }
        } else {
            for (int dy = 0; dy < 3; ++dy) {
                const int yy = y0 + dy;
                if (yy < 0 || yy >= Y) continue;
                for (int dx = 0; dx < 3; ++dx) {
                    const int xx = x0 + dx;
                    if (xx < 0 || xx >= X) continue;
                    for (int dz = 0; dz < 3; ++dz) {
                        const int zz = z - 1 + dz;
                        if (zz < 0 || zz >= Z) continue;
                        const float v = xb[((long long)yy * X + xx) * zc + (unsigned)zz * C];
                        if (best_tap < 0 || v > best || v != v) { best = v; best_tap = dy * 9 + dx * 3 + dz; }
                    }
                }
            }
        }
        const long long o = (long long)r * row_len + j;
        y[o] = best;
        arg[o] = (unsigned char)best_tap;
    }
}

// One thread per 2 x 2 (y, x) block of inputs at one (z, c): the 2 x 2 x 3 windows that can contain any of the four are read
// ONCE (12 arg bytes + 12 gradients, all independent loads) and each window's gradient goes to the one input its arg-max tap
// names -- a quarter of the window reads of a thread-per-input gather and 4 stores per thread.  Per input the windows are
// still visited in (oy, ox, oz) ascending order: same sums, bit for bit, as the thread-per-input form.
// grid: x covers one row segment (ox', z, c) of 2 x 2 blocks, y walks the (b, oy') block rows
__global__ __launch_bounds__(PL_THREADS) void maxpool_k3s221_cl_bwd_kernel(const float *__restrict__ gy, const unsigned char *__restrict__ arg,
                                                                            float *__restrict__ gx, int rows,
                                                                            int Y, int X, int Z, int C, int OY, int OX)
{
    const int BX = (X + 1) / 2, BY = (Y + 1) / 2;
    const unsigned row_len = (unsigned)BX * Z * C;
    const unsigned j = blockIdx.x * PL_THREADS + threadIdx.x;
    if (j >= row_len) return;
    const unsigned c = j % (unsigned)C;
    const unsigned t = j / (unsigned)C;
    const int z = (int)(t % (unsigned)Z);
    const int bx = (int)(t / (unsigned)Z);
    const unsigned zc = (unsigned)Z * C;
    const int oz_lo = max(z - 1, 0), oz_hi = min(z + 1, Z - 1);
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        const int by = r % BY;
        const long long b = r / BY;
        const long long ob = b * (long long)OY * OX * zc + c;
        float a00 = 0.0f, a01 = 0.0f, a10 = 0.0f, a11 = 0.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = by + a;
            if (oy >= OY) continue;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ox = bx + e;
                if (ox >= OX) continue;
                const long long base = ob + ((long long)oy * OX + ox) * zc;
                for (int oz = oz_lo; oz <= oz_hi; ++oz) {
                    const long long o = base + (unsigned)oz * C;
                    const int tap = (int)arg[o];
                    const float g = gy[o];
                    const int dy = tap / 9, rem = tap - dy * 9;
                    const int dx = rem / 3, dz = rem - dx * 3;
                    const int iy = 2 * a - 1 + dy, ix = 2 * e - 1 + dx;       // position inside the 2 x 2 block
                    if (oz - 1 + dz == z) {
                        if (iy == 0 && ix == 0) a00 = a00 + g;
                        if (iy == 0 && ix == 1) a01 = a01 + g;
                        if (iy == 1 && ix == 0) a10 = a10 + g;
                        if (iy == 1 && ix == 1) a11 = a11 + g;
                    }
                }
            }
        }
        const long long ib = b * (long long)Y * X * zc + (unsigned)z * C + c;
        const int y0 = 2 * by, x0 = 2 * bx;
        gx[ib + ((long long)y0 * X + x0) * zc] = a00;
        if (x0 + 1 < X) gx[ib + ((long long)y0 * X + x0 + 1) * zc] = a01;
        if (y0 + 1 < Y) {
            gx[ib + ((long long)(y0 + 1) * X + x0) * zc] = a10;
            if (x0 + 1 < X) gx[ib + ((long long)(y0 + 1) * X + x0 + 1) * zc] = a11;
        }
    }
}

// ---- the same two kernels on FOUR consecutive (z, c) elements per thread (round 5) --------------------------------------------------------
// A (y, x) line of a channels-last volume is F = Z * C contiguous floats; the z neighbours of element f are f -+ C.  A thread owns the quad
// f0 .. f0 + 3 (16-byte aligned when F % 4 == 0): the centre taps are one aligned 16-byte load per window position, the z-neighbour taps
// unaligned 16-byte loads at f0 -+ C (4-byte aligned: gfx950 global memory takes them) -- 27 vector loads for four outputs instead of 108
// scalar ones.  The scalar kernels above were issue / look-up bound: 158 us forward, 298 us backward on 8 x 18 x 128 x 128 x 128 -> 64 x 64
// (algorithmic 396 MB: 70 us at 5.6 TB/s).  Quads whose z neighbours would leave the line (the first / last ceil(C / 4) quads) read element by
// element.  Tap order, tie / NaN rule and the summation order of the backward are those of the scalar kernels: bit-identical results.
typedef float pl_v4 __attribute__((ext_vector_type(4)));
typedef float pl_v4u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned char pl_b4 __attribute__((ext_vector_type(4)));
typedef unsigned char pl_b4u __attribute__((ext_vector_type(4), aligned(1)));

__device__ __forceinline__ void pl_take(float v, int k, float &best, int &tap)
{
    if (tap < 0 || v > best || v != v) { best = v; tap = k; }
}

// grid: x covers one output row segment (ox, quad), y walks the (b, oy) rows
__global__ __launch_bounds__(PL_THREADS) void maxpool_k3s221_cl_fwd4_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                                             unsigned char *__restrict__ arg, int rows,
                                                                             int Y, int X, int Z, int C, int OY, int OX)
{
    const int F = Z * C, Q = F >> 2;
    const unsigned j = blockIdx.x * PL_THREADS + threadIdx.x;
    if (j >= (unsigned)(OX * Q)) return;
    const int ox = (int)(j / (unsigned)Q);
    const int f0 = 4 * (int)(j - (unsigned)ox * Q);
    const int x0 = 2 * ox - 1;
    const bool lo_ok = f0 >= C, hi_ok = f0 + 4 + C <= F;       // the whole quad has a z - 1 / z + 1 neighbour inside the line
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        const int oy = r % OY;
        const long long b = r / OY;
        const float *xb = x + b * (long long)Y * X * F;
        const int y0 = 2 * oy - 1;
        float best[4] = {0.f, 0.f, 0.f, 0.f};
        int tap[4] = {-1, -1, -1, -1};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y0 + dy;
            if (yy < 0 || yy >= Y) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = x0 + dx;
                if (xx < 0 || xx >= X) continue;
                const float *p = xb + ((long long)yy * X + xx) * F + f0;
                const int k0 = dy * 9 + dx * 3;
                const pl_v4 vc = *reinterpret_cast<const pl_v4 *>(p);
                float vl[4], vh[4];
                bool hl[4], hh[4];
                if (lo_ok) { const pl_v4u t = *reinterpret_cast<const pl_v4u *>(p - C); vl[0] = t.x; vl[1] = t.y; vl[2] = t.z; vl[3] = t.w; hl[0] = hl[1] = hl[2] = hl[3] = true; }
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hl[e] = f0 + e >= C; vl[e] = hl[e] ? p[e - C] : 0.f; }
                }
                if (hi_ok) { const pl_v4u t = *reinterpret_cast<const pl_v4u *>(p + C); vh[0] = t.x; vh[1] = t.y; vh[2] = t.z; vh[3] = t.w; hh[0] = hh[1] = hh[2] = hh[3] = true; }
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hh[e] = f0 + e + C < F; vh[e] = hh[e] ? p[e + C] : 0.f; }
                }
                const float vcs[4] = {vc.x, vc.y, vc.z, vc.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {          // tap order of the scalar kernel: dz = -1, 0, +1
                    if (hl[e]) pl_take(vl[e], k0, best[e], tap[e]);
                    pl_take(vcs[e], k0 + 1, best[e], tap[e]);
                    if (hh[e]) pl_take(vh[e], k0 + 2, best[e], tap[e]);
                }
            }
        }
        const long long o = (long long)r * OX * F + (long long)ox * F + f0;
        *reinterpret_cast<pl_v4 *>(y + o) = pl_v4{best[0], best[1], best[2], best[3]};
        *reinterpret_cast<pl_b4 *>(arg + o) = pl_b4{(unsigned char)tap[0], (unsigned char)tap[1], (unsigned char)tap[2], (unsigned char)tap[3]};
    }
}

// one thread per 2 x 2 (y, x) block of inputs at a quad of (z, c) elements; windows visited in (oy, ox, oz) ascending order per input
__global__ __launch_bounds__(PL_THREADS) void maxpool_k3s221_cl_bwd4_kernel(const float *__restrict__ gy, const unsigned char *__restrict__ arg,
                                                                             float *__restrict__ gx, int rows,
                                                                             int Y, int X, int Z, int C, int OY, int OX)
{
    const int BX = (X + 1) / 2, BY = (Y + 1) / 2;
    const int F = Z * C, Q = F >> 2;
    const unsigned j = blockIdx.x * PL_THREADS + threadIdx.x;
    if (j >= (unsigned)(BX * Q)) return;
    const int bx = (int)(j / (unsigned)Q);
    const int f0 = 4 * (int)(j - (unsigned)bx * Q);
    const bool lo_ok = f0 >= C, hi_ok = f0 + 4 + C <= F;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        const int by = r % BY;
        const long long b = r / BY;
        const long long ob = b * (long long)OY * OX * F;
        float acc[2][2][4] = {};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = by + a;
            if (oy >= OY) continue;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ox = bx + e;
                if (ox >= OX) continue;
                const long long base = ob + ((long long)oy * OX + ox) * F + f0;
#pragma unroll
                for (int d = -1; d <= 1; ++d) {         // window at oz = z + d: its tap dz must point back at z: dz = 1 - d
                    float g[4];
                    int t[4];
                    bool ok[4];
                    const bool vec = d == 0 || (d < 0 ? lo_ok : hi_ok);
                    if (vec) {
                        const pl_v4u gv = *reinterpret_cast<const pl_v4u *>(gy + base + d * C);
                        const pl_b4u tv = *reinterpret_cast<const pl_b4u *>(arg + base + d * C);
                        g[0] = gv.x; g[1] = gv.y; g[2] = gv.z; g[3] = gv.w;
                        t[0] = tv.x; t[1] = tv.y; t[2] = tv.z; t[3] = tv.w;
                        ok[0] = ok[1] = ok[2] = ok[3] = true;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int f = f0 + k + d * C;
                            ok[k] = f >= 0 && f < F;
                            g[k] = ok[k] ? gy[base + k + d * C] : 0.f;
                            t[k] = ok[k] ? (int)arg[base + k + d * C] : 0;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (!ok[k]) continue;
                        const int dy = t[k] / 9, rem = t[k] - dy * 9;
                        const int dx = rem / 3, dz = rem - dx * 3;
                        if (dz != 1 - d) continue;
                        const int iy = 2 * a - 1 + dy, ix = 2 * e - 1 + dx;       // position inside the 2 x 2 block
                        if (iy == 0 && ix == 0) acc[0][0][k] = acc[0][0][k] + g[k];
                        if (iy == 0 && ix == 1) acc[0][1][k] = acc[0][1][k] + g[k];
                        if (iy == 1 && ix == 0) acc[1][0][k] = acc[1][0][k] + g[k];
                        if (iy == 1 && ix == 1) acc[1][1][k] = acc[1][1][k] + g[k];
                    }
                }
            }
        }
        const long long ib = b * (long long)Y * X * F + f0;
        const int y0 = 2 * by, x0 = 2 * bx;
        *reinterpret_cast<pl_v4 *>(gx + ib + ((long long)y0 * X + x0) * F) = pl_v4{acc[0][0][0], acc[0][0][1], acc[0][0][2], acc[0][0][3]};
        if (x0 + 1 < X) *reinterpret_cast<pl_v4 *>(gx + ib + ((long long)y0 * X + x0 + 1) * F) = pl_v4{acc[0][1][0], acc[0][1][1], acc[0][1][2], acc[0][1][3]};
        if (y0 + 1 < Y) {
            *reinterpret_cast<pl_v4 *>(gx + ib + ((long long)(y0 + 1) * X + x0) * F) = pl_v4{acc[1][0][0], acc[1][0][1], acc[1][0][2], acc[1][0][3]};
            if (x0 + 1 < X) *reinterpret_cast<pl_v4 *>(gx + ib + ((long long)(y0 + 1) * X + x0 + 1) * F) = pl_v4{acc[1][1][0], acc[1][1][1], acc[1][1][2], acc[1][1][3]};
        }
    }
}

template <bool CL>
__global__ __launch_bounds__(PL_THREADS) void filter_flip_transpose_kernel(const float *__restrict__ w, float *__restrict__ out,
                                                                            int cout, int cin, int taps)
{
    const int n = cout * cin * taps;
    for (int i = blockIdx.x * PL_THREADS + threadIdx.x; i < n; i += gridDim.x * PL_THREADS) {
        int co, ci, t;
        if (CL) { co = i % cout; const int r = i / cout; t = r % taps; ci = r / taps; }        // out[ci][t][co]
        else { t = i % taps; const int r = i / taps; co = r % cout; ci = r / cout; }           // out[ci][co][t]
        const int ts = taps - 1 - t;
        out[i] = CL ? w[((long long)co * taps + ts) * cin + ci] : w[((long long)co * cin + ci) * taps + ts];
    }
}

// all filters of a step in ONE launch (round 6): a device table of filter records, element ranges by prefix offset; a block owns
// 1024 consecutive output elements and finds their filter by binary search over the offsets (<= a few hundred entries)
struct FlipRec { const float *w; float *out; int cout, cin, taps, cl; long long first; };      // 40 bytes, mirrored by mdt_flip_record in mdt_hip.h

__global__ __launch_bounds__(PL_THREADS) void filter_flip_transpose_batched_kernel(const FlipRec *__restrict__ recs, int n_recs, long long total)
{
    const long long base = (long long)blockIdx.x * (PL_THREADS * 4);
    for (int u = 0; u < 4; ++u) {
        const long long e = base + u * PL_THREADS + threadIdx.x;
        if (e >= total) return;
        int lo = 0, hi = n_recs - 1;
        while (lo < hi) {                                   // last record with first <= e
            const int mid = (lo + hi + 1) >> 1;
            if (recs[mid].first <= e) lo = mid; else hi = mid - 1;
        }
        const FlipRec r = recs[lo];
        const int i = (int)(e - r.first);
        int co, ci, t;
        if (r.cl) { co = i % r.cout; const int q = i / r.cout; t = q % r.taps; ci = q / r.taps; }
        else { t = i % r.taps; const int q = i / r.taps; co = q % r.cout; ci = q / r.cout; }
        const int ts = r.taps - 1 - t;
        r.out[i] = r.cl ? r.w[((long long)co * r.taps + ts) * r.cin + ci] : r.w[((long long)co * r.cin + ci) * r.taps + ts];
    }
}

inline unsigned grid_for(long long n)
{
    long long blocks = (n + PL_THREADS - 1) / PL_THREADS;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

extern "C" {

int mdt_maxpool3d_k3s221_cl_forward(const float *x, float *y, unsigned char *argmax, int batch, int Y, int X, int Z, int channels,
                                    void *stream)
{
    if (batch < 0 || Y <= 0 || X <= 0 || Z <= 0 || channels <= 0) return MDT_ERR_INVALID_ARGUMENT;
    const int OY = (Y - 1) / 2 + 1, OX = (X - 1) / 2 + 1;
    const long long row_len = (long long)OX * Z * channels, rows = (long long)batch * OY;
    if (rows == 0) return MDT_OK;
    if (row_len > 0x3fffffffLL || rows > 0x7fffffffLL || (long long)X * Z * channels > 0x3fffffffLL) return MDT_ERR_UNSUPPORTED;
    (void)hipGetLastError();
    const long long F = (long long)Z * channels;
    if (F % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 && (reinterpret_cast<uintptr_t>(argmax) & 3) == 0) {
        const long long quads = (long long)OX * (F / 4);
        hipLaunchKernelGGL(maxpool_k3s221_cl_fwd4_kernel, dim3((unsigned)((quads + PL_THREADS - 1) / PL_THREADS), (unsigned)(rows < 65535 ? rows : 65535)),
                           dim3(PL_THREADS), 0, (hipStream_t)stream, x, y, argmax, (int)rows, Y, X, Z, channels, OY, OX);
        return pl_check();
    }
    hipLaunchKernelGGL(maxpool_k3s221_cl_fwd_kernel, dim3((unsigned)((row_len + PL_THREADS - 1) / PL_THREADS), (unsigned)(rows < 65535 ? rows : 65535)),
                       dim3(PL_THREADS), 0, (hipStream_t)stream, x, y, argmax, (int)rows, Y, X, Z, channels, OY, OX);
    return pl_check();
}

int mdt_maxpool3d_k3s221_cl_backward(const float *gy, const unsigned char *argmax, float *gx, int batch, int Y, int X, int Z,
                                     int channels, void *stream)
{
    if (batch < 0 || Y <= 0 || X <= 0 || Z <= 0 || channels <= 0) return MDT_ERR_INVALID_ARGUMENT;
    const int OY = (Y - 1) / 2 + 1, OX = (X - 1) / 2 + 1;
    const long long row_len = (long long)((X + 1) / 2) * Z * channels, rows = (long long)batch * ((Y + 1) / 2);
    if (rows == 0) return MDT_OK;
    if (row_len > 0x3fffffffLL || rows > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    (void)hipGetLastError();
    const long long F = (long long)Z * channels;
    if (F % 4 == 0 && (reinterpret_cast<uintptr_t>(gx) & 15) == 0) {
        const long long quads = (long long)((X + 1) / 2) * (F / 4);
        hipLaunchKernelGGL(maxpool_k3s221_cl_bwd4_kernel, dim3((unsigned)((quads + PL_THREADS - 1) / PL_THREADS), (unsigned)(rows < 65535 ? rows : 65535)),
                           dim3(PL_THREADS), 0, (hipStream_t)stream, gy, argmax, gx, (int)rows, Y, X, Z, channels, OY, OX);
        return pl_check();
    }
    hipLaunchKernelGGL(maxpool_k3s221_cl_bwd_kernel, dim3((unsigned)((row_len + PL_THREADS - 1) / PL_THREADS), (unsigned)(rows < 65535 ? rows : 65535)),
                       dim3(PL_THREADS), 0, (hipStream_t)stream, gy, argmax, gx, (int)rows, Y, X, Z, channels, OY, OX);
    return pl_check();
}

int mdt_filter_flip_transpose(const float *w, float *out, int cout, int cin, int taps, int channels_last, void *stream)
{
    if (cout <= 0 || cin <= 0 || taps <= 0 || (long long)cout * cin * taps > 0x7fffffffLL) return MDT_ERR_INVALID_ARGUMENT;
    const long long n = (long long)cout * cin * taps;
    (void)hipGetLastError();
    if (channels_last) hipLaunchKernelGGL(filter_flip_transpose_kernel<true>, dim3(grid_for(n)), dim3(PL_THREADS), 0, (hipStream_t)stream, w, out, cout, cin, taps);
    else hipLaunchKernelGGL(filter_flip_transpose_kernel<false>, dim3(grid_for(n)), dim3(PL_THREADS), 0, (hipStream_t)stream, w, out, cout, cin, taps);
    return pl_check();
}

int mdt_filter_flip_transpose_batched(const void *records_dev, int n_records, long long total_elements, void *stream)
{
    if (n_records < 0 || total_elements < 0 || (n_records > 0 && records_dev == nullptr)) return MDT_ERR_INVALID_ARGUMENT;
    if (n_records == 0 || total_elements == 0) return MDT_OK;
    const long long blocks = (total_elements + PL_THREADS * 4 - 1) / (PL_THREADS * 4);
    if (blocks > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    (void)hipGetLastError();
    hipLaunchKernelGGL(filter_flip_transpose_batched_kernel, dim3((unsigned)blocks), dim3(PL_THREADS), 0, (hipStream_t)stream,
                       reinterpret_cast<const FlipRec *>(records_dev), n_records, total_elements);
    return pl_check();
}

}  // extern "C"
