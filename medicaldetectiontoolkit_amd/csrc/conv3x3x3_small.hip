// conv3x3x3_small.hip -- 3x3x3, stride 1, pad 1 convolution with FEW channels (C_in even <= 32, C_out <= 32) on
// channels-last fp32 activations, for gfx950: the 18 -> 18 bottleneck convolutions on the 8 x 32x32x128 maps of the ResNet
// stage C2 (models/backbone.py:186-190 ResBlock.conv2).  MIOpen's solvers run them at 17-21 TF/s (862 us forward-type, six
// such calls per training step: profiles/r03_op_profile.txt); padded to the 32 x 32 fp32 MFMA tile the same sums take 207 us
// of MFMA time.  Used for the forward pass and -- with the flipped, transposed filter -- for the input gradient
// (utils/fused_epilogue._ConvStride1).
//
//   out[v][co] = sum_{tap, ci} in[v + off(tap)][ci] * Wt[tap][ci][co]            (zero padding)
//
// v_mfma_f32_32x32x2_f32 with M = 32 consecutive z voxels of one (b, y, x) column, N = output channels, K = (tap, ci pairs):
// 27 taps x C_in / 2 K-steps per tile.  The A operand (lane <-> (voxel l & 31, channel pair element l >> 5)) is read from an
// LDS image of the 3 x 6 input rows a workgroup's four x positions need ([row][34 voxels][C_in] floats: voxel stride C_in
// words -- bank-conflict-free for C_in = 18), the B operand from the LDS copy of the filter ([tap][ci][co], loaded once per
// workgroup).  A workgroup (4 waves = 4 x positions) walks the z axis in tiles of 32, reloading only the A image (LDS-DMA).
// Exact fp32 (the MFMA is an fmaf chain); sums are ordered (tap, ci) ascending per output.
// Compute-bound on the padded tile (C/32 x C/32 of it useful); HBM traffic = in + out once (+ L2-served re-reads of rows).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mdt_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C3_THREADS = 256;
constexpr int C3_XT = 4;              // x positions per workgroup (one per wave)
constexpr int C3_ZT = 32;             // z voxels per tile (the MFMA's M)

struct C3Params {
    const float *in;                  // [B, Y, X, Z, Cin]
    const float *wt;                  // [27][Cin][Cout]
    float *out;                       // [B, Y, X, Z, Cout]
    const float *bias;                // [Cout] or null: added in the epilogue (round 4)
    int relu;                         // 1: max(., 0) in the epilogue
    int B, Y, X, Z, Cin, Cout;
    int xg;                           // x groups per row: ceil(X / 4)
    int row_floats;                   // (ZT + 2) * Cin
};

// KS = C_in / 2 K-steps per tap as a compile-time constant (operand arrays in registers), or 0 for the generic loop
// EPI: bias add + ReLU in the epilogue -- a compile-time switch: as a run-time branch the extra code raised the plain kernel from 235 to 250
// VGPRs and halved its occupancy (383 -> 640 us on the 18 -> 18 layer, round 4)
template <int KS, bool EPI>
__global__ __launch_bounds__(C3_THREADS) void conv3x3x3_small_kernel(C3Params p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Wl = lds;                                        // [27 * Cin * Cout]
    float *Al = lds + ((27 * p.Cin * p.Cout + 3) & ~3);     // [3][XT + 2][ZT + 2][Cin]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int g = blockIdx.x;
    const int xq = g % p.xg; g /= p.xg;
    const int y = g % p.Y;
    const int b = g / p.Y;
    const int x0 = xq * C3_XT;
    const int nW = 27 * p.Cin * p.Cout;
    for (int c = wave; c * 64 < nW; c += C3_THREADS / 64) {       // filter -> LDS by LDS-DMA as well (34 dependent load/store trips otherwise)
        const int t = c * 64 + lane;
        if (t < nW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p.wt + t),
                                             (__attribute__((address_space(3))) void *)(Wl + c * 64), 4, 0, 0);
    }
    const int rf = p.row_floats;
    const int i = lane & 31, kk = lane >> 5;
    const int x = x0 + wave;
    const int ksteps = p.Cin >> 1;
    const int cj = min(i, p.Cout - 1);                      // lanes past C_out read a valid column; their results are not stored
    for (int z0 = 0; z0 < p.Z; z0 += C3_ZT) {
        __syncthreads();                                    // previous tile's readers are done (and, first trip, Wl is being filled)
        // ---- A image: rows (y - 1 .. y + 1) x (x0 - 1 .. x0 + XT), voxels z0 - 1 .. z0 + ZT, zero outside the volume
        const bool first_z = (z0 == 0), last_z = (z0 + C3_ZT >= p.Z);
        // global -> LDS by LDS-DMA (no staging registers, every wave's ~45 transfers in flight at once: with ordinary loads the
        // fill was 54 dependent round trips per tile and took 6x the tile's MFMA time); zeros are written with ds_write
        const int chunks = (rf + 63) >> 6;
        for (int row = 0; row < 3 * (C3_XT + 2); ++row) {            // uniform: row -> (ry, rx) in scalar registers
            const int ry = row / (C3_XT + 2), rx = row - ry * (C3_XT + 2);
            const int yy = y + ry - 1, xx = x0 + rx - 1;
            const bool row_ok = yy >= 0 && yy < p.Y && xx >= 0 && xx < p.X;
            const float *src = p.in + ((((long long)b * p.Y + yy) * p.X + xx) * p.Z + z0 - 1) * p.Cin;
            float *dst = Al + row * rf;
            for (int c = wave; c < chunks; c += C3_THREADS / 64) {
                const int e = c * 64 + lane;
                if (e < rf) {
                    // only the first / last voxel of the row can fall outside [0, Z)
                    const bool ok = row_ok && !(first_z && e < p.Cin) && !(last_z && e >= (C3_ZT + 1) * p.Cin);
                    if (ok)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + e),
                                                         (__attribute__((address_space(3))) void *)(dst + c * 64), 4, 0, 0);
                    else
                        dst[e] = 0.0f;
                }
            }
        }
        __syncthreads();
        if (x < p.X) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            auto tap_rows = [&](int tap, const float *&arow, const float *&brow) {
                const int dy = tap / 9, rem = tap - dy * 9;
                const int dx = rem / 3, dz = rem - dx * 3;
                arow = Al + (dy * (C3_XT + 2) + wave + dx) * rf + (i + dz) * p.Cin + kk;
                brow = Wl + (tap * p.Cin + kk) * p.Cout + cj;
            };
            if (KS > 0) {
                // two operand register sets: the LDS reads of tap t + 1 are issued before the MFMAs of tap t (the compiler's own
                // schedule was read -> wait -> MFMA for every single MFMA: 170 cycles each instead of 64)
                constexpr int K = KS > 0 ? KS : 1;
                float a0[K], b0[K], a1[K], b1[K];
                auto load_tap = [&](int tap, float (&a)[K], float (&b)[K]) {
                    const float *arow, *brow;
                    tap_rows(tap, arow, brow);
#pragma unroll
                    for (int k = 0; k < K; ++k) { a[k] = arow[2 * k]; b[k] = brow[(2 * k) * p.Cout]; }
                };
                auto mfma_tap = [&](const float (&a)[K], const float (&b)[K]) {
#pragma unroll
                    for (int k = 0; k < K; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc, 0, 0, 0);
                };
                load_tap(0, a0, b0);
                for (int tap = 0; tap < 27; tap += 2) {
                    if (tap + 1 < 27) load_tap(tap + 1, a1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_tap(a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap + 1 >= 27) break;
                    if (tap + 2 < 27) load_tap(tap + 2, a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_tap(a1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                for (int tap = 0; tap < 27; ++tap) {
                    const float *arow, *brow;
                    tap_rows(tap, arow, brow);
                    for (int ks = 0; ks < ksteps; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * ks], brow[(2 * ks) * p.Cout], acc, 0, 0, 0);
                }
            }
            // C/D map of the 32x32 MFMA: column = lane & 31 (output channel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (voxel)
            if (i < p.Cout) {
                float *o = p.out + ((((long long)b * p.Y + y) * p.X + x) * p.Z + z0) * p.Cout + i;
                if (EPI) {       // bias add + ReLU here instead of a separate pass over the output (epilogue.hip)
                    const float bv = p.bias ? p.bias[i] : 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int vz = (r & 3) + 8 * (r >> 2) + 4 * kk;
                        float v = acc[r] + bv;
                        if (p.relu) v = v > 0.0f ? v : 0.0f;
                        o[(long long)vz * p.Cout] = v;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int vz = (r & 3) + 8 * (r >> 2) + 4 * kk;
                        o[(long long)vz * p.Cout] = acc[r];
                    }
                }
            }
        }
    }
}

// ---- weight gradient of the same layer:  dW[tap][ci][co] = sum_v X[v + off(tap)][ci] * dY[v][co]
// M = input channels (A = the shifted input voxel: lanes run over ci, conflict-free), N = output channels (B = dY), K = voxels.
// The workgroup tile and the LDS image of the input rows are those of the forward kernel, plus the tile's 4 x 32 dY rows; the 27
// taps are dealt to the four waves (7, 7, 7, 6 accumulator tiles), every wave walking all 64 K-steps of the tile.  A workgroup
// accumulates over its z tiles and writes one [27][C_in][C_out] partial; a second kernel adds the partials in a fixed order.
constexpr int C3_WTAPS = 7;           // taps per wave (4 x 7 >= 27)

__global__ __launch_bounds__(C3_THREADS) void conv3x3x3_small_wgrad_kernel(C3Params p, const float *__restrict__ dy, float *__restrict__ ws)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Al = lds;                                        // [3][XT + 2][ZT + 2][Cin]
    float *Dl = lds + 3 * (C3_XT + 2) * p.row_floats;       // [XT][ZT][Cout]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int g = blockIdx.x;
    const int xq = g % p.xg; g /= p.xg;
    const int y = g % p.Y;
    const int b = g / p.Y;
    const int x0 = xq * C3_XT;
    const int rf = p.row_floats;
    const int i = lane & 31, kk = lane >> 5;
    const int ci = min(i, p.Cin - 1), cj = min(i, p.Cout - 1);
    const int drow = C3_ZT * p.Cout;                        // floats of one dY row in the tile
    f32x16 acc[C3_WTAPS];
#pragma unroll
    for (int t = 0; t < C3_WTAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    // per-lane LDS offsets of this wave's taps inside the input image (relative to x position 0, voxel 0)
    int aoff[C3_WTAPS];
#pragma unroll
    for (int t = 0; t < C3_WTAPS; ++t) {
        const int tap = min(wave * C3_WTAPS + t, 26);
        const int dy_ = tap / 9, rem = tap - dy_ * 9;
        const int dx = rem / 3, dz = rem - dx * 3;
        aoff[t] = (dy_ * (C3_XT + 2) + dx) * rf + (dz + kk) * p.Cin + ci;
    }
    for (int z0 = 0; z0 < p.Z; z0 += C3_ZT) {
        __syncthreads();
        const bool first_z = (z0 == 0), last_z = (z0 + C3_ZT >= p.Z);
        const int chunks = (rf + 63) >> 6;
        for (int row = 0; row < 3 * (C3_XT + 2); ++row) {
            const int ry = row / (C3_XT + 2), rx = row - ry * (C3_XT + 2);
            const int yy = y + ry - 1, xx = x0 + rx - 1;
            const bool row_ok = yy >= 0 && yy < p.Y && xx >= 0 && xx < p.X;
            const float *src = p.in + ((((long long)b * p.Y + yy) * p.X + xx) * p.Z + z0 - 1) * p.Cin;
            float *dst = Al + row * rf;
            for (int c = wave; c < chunks; c += C3_THREADS / 64) {
                const int e = c * 64 + lane;
                if (e < rf) {
                    const bool ok = row_ok && !(first_z && e < p.Cin) && !(last_z && e >= (C3_ZT + 1) * p.Cin);
                    if (ok)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + e),
                                                         (__attribute__((address_space(3))) void *)(dst + c * 64), 4, 0, 0);
                    else
                        dst[e] = 0.0f;
                }
            }
        }
        const int dchunks = (drow + 63) >> 6;
        for (int xp = 0; xp < C3_XT; ++xp) {                 // dY rows of the tile (zeros past the last x)
            const bool row_ok = x0 + xp < p.X;
            const float *src = dy + ((((long long)b * p.Y + y) * p.X + x0 + xp) * p.Z + z0) * p.Cout;
            float *dst = Dl + xp * drow;
            for (int c = wave; c < dchunks; c += C3_THREADS / 64) {
                const int e = c * 64 + lane;
                if (e < drow) {
                    if (row_ok)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + e),
                                                         (__attribute__((address_space(3))) void *)(dst + c * 64), 4, 0, 0);
                    else
                        dst[e] = 0.0f;
                }
            }
        }
        __syncthreads();
        // 4 x positions x 16 K-steps; per K-step one B read and this wave's 7 A reads, issued two K-steps ahead of their MFMAs
        auto load_step = [&](int s, float (&a)[C3_WTAPS], float &bv) {
            const int xp = s >> 4, zk = (s & 15) * 2;
            bv = Dl[xp * drow + (zk + kk) * p.Cout + cj];
            const float *base = Al + xp * rf + zk * p.Cin;
#pragma unroll
            for (int t = 0; t < C3_WTAPS; ++t) a[t] = base[aoff[t]];
        };
        auto mfma_step = [&](const float (&a)[C3_WTAPS], float bv) {
#pragma unroll
            for (int t = 0; t < C3_WTAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bv, acc[t], 0, 0, 0);
        };
        float a0[C3_WTAPS], a1[C3_WTAPS], b0, b1;
        load_step(0, a0, b0);
        constexpr int NS = C3_XT * (C3_ZT / 2);
        for (int s = 0; s < NS; s += 2) {
            load_step(s + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < NS) load_step(s + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // C/D map: column = lane & 31 (co), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (ci)
    float *out = ws + (size_t)blockIdx.x * 27 * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < C3_WTAPS; ++t) {
        const int tap = wave * C3_WTAPS + t;
        if (tap < 27 && i < p.Cout) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c_in = (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (c_in < p.Cin) out[((size_t)tap * p.Cin + c_in) * p.Cout + i] = acc[t][r];
            }
        }
    }
}

__global__ __launch_bounds__(256) void conv3x3x3_small_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dW, int n_elem, int n_part)
{
    __shared__ float part[16][17];
    const int sub = threadIdx.x & 15, el = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    float s = 0.0f;
    if (e < n_elem) {
        int q = sub;
        for (; q + 16 * 7 < n_part; q += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(q + 16 * u) * n_elem + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s = s + v[u];
        }
        for (; q < n_part; q += 16) s = s + ws[(size_t)q * n_elem + e];
    }
    part[el][sub] = s;
    __syncthreads();
    if (sub == 0 && e < n_elem) {
        float t = part[el][0];
#pragma unroll
        for (int k = 1; k < 16; ++k) t = t + part[el][k];
        dW[e] = t;
    }
}

inline int c3_check()
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

}  // namespace

extern "C" {

int mdt_conv3x3x3_small_supported(int Y, int X, int Z, int c_in, int c_out)
{
    if (Y <= 0 || X <= 0 || Z <= 0 || Z % C3_ZT != 0) return 0;
    if (c_in < 2 || c_in > 32 || (c_in & 1) || c_out < 1 || c_out > 32) return 0;
    const size_t lds = ((size_t)((27 * c_in * c_out + 3) & ~3) + (size_t)3 * (C3_XT + 2) * (C3_ZT + 2) * c_in) * sizeof(float);
    return lds <= 80 * 1024 ? 1 : 0;
}

int mdt_conv3x3x3_small_forward(const float *in, const float *w_tap_ci_co, float *out, int batch, int Y, int X, int Z,
                                int c_in, int c_out, void *stream)
{
    return mdt_conv3x3x3_small_forward_bias_act(in, w_tap_ci_co, nullptr, 0, out, batch, Y, X, Z, c_in, c_out, stream);
}

int mdt_conv3x3x3_small_forward_bias_act(const float *in, const float *w_tap_ci_co, const float *bias, int relu, float *out, int batch, int Y, int X, int Z,
                                         int c_in, int c_out, void *stream)
{
    if (!in || !w_tap_ci_co || !out || batch <= 0) return MDT_ERR_INVALID_ARGUMENT;
    if (!mdt_conv3x3x3_small_supported(Y, X, Z, c_in, c_out)) return MDT_ERR_UNSUPPORTED;
    C3Params p;
    p.in = in; p.wt = w_tap_ci_co; p.out = out; p.bias = bias; p.relu = relu ? 1 : 0;
    p.B = batch; p.Y = Y; p.X = X; p.Z = Z; p.Cin = c_in; p.Cout = c_out;
    p.xg = (X + C3_XT - 1) / C3_XT;
    p.row_floats = (C3_ZT + 2) * c_in;
    const long long grid = (long long)batch * Y * p.xg;
    if (grid > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    const size_t lds = ((size_t)((27 * c_in * c_out + 3) & ~3) + (size_t)3 * (C3_XT + 2) * p.row_floats) * sizeof(float);
    static bool optin = false;
    if (!optin) {
        (void)hipFuncSetAttribute((const void *)conv3x3x3_small_kernel<9, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void *)conv3x3x3_small_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void *)conv3x3x3_small_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void *)conv3x3x3_small_kernel<9, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void *)conv3x3x3_small_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void *)conv3x3x3_small_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipGetLastError();
        optin = true;
    }
    (void)hipGetLastError();
    const bool epi = bias != nullptr || relu;
#define C3_LAUNCH(KS) do { if (epi) hipLaunchKernelGGL((conv3x3x3_small_kernel<KS, true>), dim3((unsigned)grid), dim3(C3_THREADS), lds, static_cast<hipStream_t>(stream), p); \
                           else hipLaunchKernelGGL((conv3x3x3_small_kernel<KS, false>), dim3((unsigned)grid), dim3(C3_THREADS), lds, static_cast<hipStream_t>(stream), p); } while (0)
    if (c_in == 18) C3_LAUNCH(9);
    else if (c_in == 6) C3_LAUNCH(3);
    else C3_LAUNCH(0);
#undef C3_LAUNCH
    return c3_check();
}

size_t mdt_conv3x3x3_small_wgrad_workspace_bytes(int batch, int Y, int X, int c_in, int c_out)
{
    if (batch <= 0 || Y <= 0 || X <= 0 || c_in <= 0 || c_out <= 0) return 0;
    return (size_t)batch * Y * ((X + C3_XT - 1) / C3_XT) * 27 * c_in * c_out * sizeof(float) + 256;
}

int mdt_conv3x3x3_small_wgrad(const float *in, const float *grad_out, float *grad_w_tap_ci_co, int batch, int Y, int X, int Z,
                              int c_in, int c_out, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!in || !grad_out || !grad_w_tap_ci_co || batch <= 0) return MDT_ERR_INVALID_ARGUMENT;
    if (!mdt_conv3x3x3_small_supported(Y, X, Z, c_in, c_out)) return MDT_ERR_UNSUPPORTED;
    C3Params p;
    p.in = in; p.wt = nullptr; p.out = nullptr;
    p.B = batch; p.Y = Y; p.X = X; p.Z = Z; p.Cin = c_in; p.Cout = c_out;
    p.xg = (X + C3_XT - 1) / C3_XT;
    p.row_floats = (C3_ZT + 2) * c_in;
    const long long grid = (long long)batch * Y * p.xg;
    if (grid > 0x7fffffLL) return MDT_ERR_UNSUPPORTED;
    const size_t need = (size_t)grid * 27 * c_in * c_out * sizeof(float);
    if (!workspace || workspace_bytes < need) return MDT_ERR_WORKSPACE_TOO_SMALL;
    const size_t lds = ((size_t)3 * (C3_XT + 2) * p.row_floats + (size_t)C3_XT * C3_ZT * c_out) * sizeof(float);
    if (lds > 80 * 1024) return MDT_ERR_UNSUPPORTED;
    static bool optin = false;
    if (!optin) {
        (void)hipFuncSetAttribute((const void *)conv3x3x3_small_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipGetLastError();
        optin = true;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    hipLaunchKernelGGL(conv3x3x3_small_wgrad_kernel, dim3((unsigned)grid), dim3(C3_THREADS), lds, s, p, grad_out, static_cast<float *>(workspace));
    const int n_elem = 27 * c_in * c_out;
    hipLaunchKernelGGL(conv3x3x3_small_reduce_kernel, dim3((unsigned)((n_elem + 15) / 16)), dim3(256), 0, s, static_cast<const float *>(workspace),
                       grad_w_tap_ci_co, n_elem, (int)grid);
    return c3_check();
}

}  // extern "C"
