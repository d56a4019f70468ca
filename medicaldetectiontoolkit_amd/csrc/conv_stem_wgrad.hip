// conv_stem_wgrad.hip -- weight gradient of the ResNet stem on gfx950: a k x k x k (k = 7) convolution of a ONE-channel volume
// with stride (sy, sx, 1) (models/backbone.py:66-68: C1 = conv(1 -> 18, ks 7, stride (2, 2, 1), pad 3)):
//     dW[co][ky, kx, kz] = sum_{b, oy, ox, oz} dY[b, oy, ox, oz][co] * Xpad[b, sy*oy + ky, sx*ox + kx, oz + kz]
// a [C_out x V] x [V x 343] contraction over the V = 4.2 M output voxels: 52 GFLOP, 3.7 ms in MIOpen's backward-weights
// (the largest single weight gradient of the training step, profiles/r03_op_profile.txt).
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32), M = output channels (A = dY rows, channels-last: lane <-> (channel l & 31, voxel l >> 5),
// the wave's load IS the fragment, as in conv1x1_wgrad.hip), N = filter taps in tiles of 32 (B = the input voxel each tap
// reads for that output voxel: a gather inside a 7^3 neighbourhood of the zero-padded volume, served by L1/L2), K = voxels.
// The caller pads the volume by k / 2 on every face, so no lane ever tests a border.  The four waves of a workgroup take the
// same output rows (b, oy, ox, :) but different tap tiles (3 each: 48 accumulator registers); a workgroup sweeps a contiguous run
// of rows (input windows of neighbouring rows overlap: L1 reuse); operands are double-buffered in registers (4 K-steps per trip).  Every workgroup writes one [C_out][taps]
// partial and a second kernel adds the partials in a fixed order: deterministic, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mdt_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SW_THREADS = 256;
constexpr int SW_NTW = 3;            // tap tiles per wave
constexpr int SW_UNROLL = 4;         // K-steps (of 2 voxels) per trip

struct SWParams {
    const float *dy;                 // [B, OY, OX, OZ, Co]
    const float *xp;                 // [B, YP, XP, ZP] zero-padded input
    float *ws;                       // [n_wg][Co][T]
    int B, OY, OX, OZ, Co;
    int YP, XP, ZP, sy, sx;
    int k, T;                        // kernel extent, taps = k^3
    long long rows;                  // B * OY * OX
};

__global__ __launch_bounds__(SW_THREADS) void conv_stem_wgrad_partial_kernel(SWParams p)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = lane & 31, kk = lane >> 5;
    // per-lane constants: the tap each tile column reads (as an offset inside the padded volume) and the dY column
    unsigned tapoff[SW_NTW];
#pragma unroll
    for (int n = 0; n < SW_NTW; ++n) {
        int t = (wave * SW_NTW + n) * 32 + ch;
        if (t > p.T - 1) t = p.T - 1;                       // columns past the last tap read a valid address; never stored
        const int ky = t / (p.k * p.k), rem = t - ky * p.k * p.k;
        const int kx = rem / p.k, kz = rem - kx * p.k;
        tapoff[n] = (unsigned)((ky * p.XP + kx) * p.ZP + kz + kk);      // + kk: the K-step's second voxel is one z further
    }
    const unsigned offa = (unsigned)(kk * p.Co + min(ch, p.Co - 1));
    f32x16 acc[SW_NTW];
#pragma unroll
    for (int n = 0; n < SW_NTW; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

    const int trips = p.OZ / (2 * SW_UNROLL);               // OZ is a multiple of 8 (checked by the launcher)
    // a workgroup sweeps a CONTIGUOUS run of output rows (consecutive ox of one (b, oy) line): neighbouring rows read input
    // windows that overlap by (k - sx) / k, so the gathers of the B operand hit in L1 instead of L2
    const long long chunk = (p.rows + gridDim.x - 1) / gridDim.x;
    const long long row_end = min(p.rows, (long long)(blockIdx.x + 1) * chunk);
    for (long long row = (long long)blockIdx.x * chunk; row < row_end; ++row) {
        const int ox = (int)(row % p.OX);
        const long long r2 = row / p.OX;
        const int oy = (int)(r2 % p.OY);
        const long long b = r2 / p.OY;
        const float *arow = p.dy + row * p.OZ * p.Co;                                             // wave-uniform bases
        const float *brow = p.xp + ((b * p.YP + (long long)oy * p.sy) * p.XP + (long long)ox * p.sx) * p.ZP;
        auto load_trip = [&](int trip, float (&a)[SW_UNROLL], float (&bv)[SW_UNROLL][SW_NTW]) {
            const int z = trip * 2 * SW_UNROLL;
#pragma unroll
            for (int u = 0; u < SW_UNROLL; ++u) {
                a[u] = arow[(z + 2 * u) * p.Co + offa];
#pragma unroll
                for (int n = 0; n < SW_NTW; ++n) bv[u][n] = brow[(z + 2 * u) + tapoff[n]];
            }
        };
        auto mfma_trip = [&](const float (&a)[SW_UNROLL], const float (&bv)[SW_UNROLL][SW_NTW]) {
#pragma unroll
            for (int u = 0; u < SW_UNROLL; ++u)
#pragma unroll
                for (int n = 0; n < SW_NTW; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bv[u][n], acc[n], 0, 0, 0);
        };
        float a0[SW_UNROLL], b0[SW_UNROLL][SW_NTW], a1[SW_UNROLL], b1[SW_UNROLL][SW_NTW];
        load_trip(0, a0, b0);
        for (int trip = 0; trip < trips; trip += 2) {
            if (trip + 1 < trips) load_trip(trip + 1, a1, b1);
            mfma_trip(a0, b0);
            if (trip + 1 >= trips) break;
            if (trip + 2 < trips) load_trip(trip + 2, a0, b0);
            mfma_trip(a1, b1);
        }
    }
    // C/D map: column = lane & 31 (tap within the tile), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (output channel)
    float *out = p.ws + (size_t)blockIdx.x * p.Co * p.T;
#pragma unroll
    for (int n = 0; n < SW_NTW; ++n) {
        const int t = (wave * SW_NTW + n) * 32 + ch;
        if (t < p.T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (co < p.Co) out[(size_t)co * p.T + t] = acc[n][r];
            }
        }
    }
}

// ---- second form (k = 7, stride (2, 2), OZ in {32, 64, 128}, OX % 4 == 0): the B operand from an LDS image ------------------------
// The gathers of the first form go through the vector memory pipe (one 64-lane gather per MFMA: 6-10 cache lines each) and hold
// it at 36 % of the MFMA bound.  Here a workgroup stages the 7 x 13 input lines of four neighbouring output columns in LDS
// (as csrc/conv_stem_fwd.hip does: 7 contiguous runs of the padded volume, by LDS-DMA), every wave takes ONE output column and
// ALL 11 tap tiles (176 accumulator registers), so one dY fragment load and 11 ds_read_b32 (address = per-lane tap offset +
// compile-time voxel offset) feed 11 MFMAs.  The four waves' sums are added through LDS in a fixed order at the end.
constexpr int SL_K = 7, SL_T = 343, SL_NT = 11, SL_NX = 4, SL_COLS = 2 * (SL_NX - 1) + SL_K, SL_GS = 4;

template <int OZ>
__global__ __launch_bounds__(SW_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_stem_wgrad_lds_kernel(SWParams p)
{
    constexpr int ZP = OZ + SL_K - 1;
    constexpr int RS = SL_COLS * ZP;
    constexpr int STEPS = OZ / 2, GROUPS = STEPS / SL_GS;
    static_assert(GROUPS % 2 == 0 && SL_GS % 2 == 0, "buffer parities must be compile-time");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *img = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = lane & 31, kk = lane >> 5;
    // per-lane B bases: the tap of this lane in tile n (as an offset inside the image of this wave's column) + the K-step's voxel parity
    const float *bbase[SL_NT];
#pragma unroll
    for (int n = 0; n < SL_NT; ++n) {
        int t = n * 32 + ch;
        if (t > SL_T - 1) t = SL_T - 1;                     // columns past the last tap read a valid address; never stored
        const int ky = t / (SL_K * SL_K), rem = t - ky * SL_K * SL_K;
        const int kx = rem / SL_K, kz = rem - kx * SL_K;
        bbase[n] = img + ky * RS + (2 * wave + kx) * ZP + kz + kk;
    }
    const unsigned offa = (unsigned)(kk * p.Co + min(ch, p.Co - 1));
    f32x16 acc[SL_NT];
#pragma unroll
    for (int n = 0; n < SL_NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

    const int oxg_n = p.OX / SL_NX;
    const int passes = (int)(p.rows / SL_NX);
    const int per = (passes + gridDim.x - 1) / gridDim.x;
    const int pass_end = min(passes, (int)(blockIdx.x + 1) * per);
    for (int pass = blockIdx.x * per; pass < pass_end; ++pass) {
        const int oxg = pass % oxg_n;
        const int r2 = pass / oxg_n;
        const int oy = r2 % p.OY, b = r2 / p.OY;
        __syncthreads();
        {
            constexpr int CH = (RS + 63) / 64;
            const float *src0 = p.xp + (((long long)b * p.YP + (long long)p.sy * oy) * p.XP + (long long)p.sx * SL_NX * oxg) * ZP;
            const long long ystride = (long long)p.XP * ZP;
            for (int c = wave; c < SL_K * CH; c += SW_THREADS / 64) {
                const int ky = c / CH, cc = c - ky * CH;
                if (cc * 64 + lane < RS)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src0 + ky * ystride + cc * 64 + lane),
                                                     (__attribute__((address_space(3))) void *)(img + ky * RS + cc * 64), 4, 0, 0);
            }
        }
        __syncthreads();
        const float *arow = p.dy + ((((long long)b * p.OY + oy) * p.OX + oxg * SL_NX + wave) * OZ) * p.Co + offa;
        float a[2][SL_GS], bv[2][SL_NT];
        auto load_a = [&](int grp, float (&d)[SL_GS]) {
#pragma unroll
            for (int u = 0; u < SL_GS; ++u) d[u] = arow[(grp * SL_GS + u) * 2 * p.Co];
        };
        auto load_b = [&](int step, float (&d)[SL_NT]) {
#pragma unroll
            for (int n = 0; n < SL_NT; ++n) d[n] = bbase[n][2 * step];
        };
        load_a(0, a[0]);
        load_b(0, bv[0]);
        // two groups (8 K-steps, 88 MFMAs) per trip: buffer parities are compile-time, the code stays well inside the I-cache
#pragma unroll 1
        for (int g2 = 0; g2 < GROUPS; g2 += 2) {
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int grp = g2 + gg;
                if (grp + 1 < GROUPS) load_a(grp + 1, a[(gg + 1) & 1]);
#pragma unroll
                for (int u = 0; u < SL_GS; ++u) {
                    const int step = grp * SL_GS + u;
                    __builtin_amdgcn_sched_barrier(0);
                    if (step + 1 < STEPS) load_b(step + 1, bv[(u + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int n = 0; n < SL_NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gg][u], bv[u & 1][n], acc[n], 0, 0, 0);
                }
            }
        }
    }
    // add the four waves' sums in a fixed order (wave 0 + 1 + 2 + 3) through LDS, then wave 0 writes the workgroup's partial
    float *red = lds;                                       // [SL_NT * 16][64]
    for (int w = 1; w < SW_THREADS / 64; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int n = 0; n < SL_NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(n * 16 + r) * 64 + lane] = acc[n][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int n = 0; n < SL_NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[n][r] = acc[n][r] + red[(n * 16 + r) * 64 + lane];
        }
    }
    if (wave == 0) {
        float *out = p.ws + (size_t)blockIdx.x * p.Co * p.T;
#pragma unroll
        for (int n = 0; n < SL_NT; ++n) {
            const int t = n * 32 + ch;
            if (t < p.T) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = (r & 3) + 8 * (r >> 2) + 4 * kk;
                    if (co < p.Co) out[(size_t)co * p.T + t] = acc[n][r];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void conv_stem_wgrad_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dW, int n_elem, int n_part)
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

inline int sw_wgs()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        (void)hipGetLastError();
        n = 2 * cus;
    }
    return n;
}

// LDS-image form: one workgroup per resident slot (the register count decides how many fit a CU), one partial per workgroup
template <int OZ>
long long launch_lds(const SWParams &p, hipStream_t s)
{
    constexpr size_t img = (size_t)SL_K * SL_COLS * (OZ + SL_K - 1) * sizeof(float), red = (size_t)SL_NT * 16 * 64 * sizeof(float);
    constexpr size_t lds = img > red ? img : red;
    static int per_cu = 0;
    if (per_cu == 0) {
        (void)hipFuncSetAttribute((const void *)conv_stem_wgrad_lds_kernel<OZ>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_stem_wgrad_lds_kernel<OZ>, SW_THREADS, lds) != hipSuccess || n < 1) n = 1;
        (void)hipGetLastError();
        per_cu = n > 2 ? 2 : n;
    }
    long long wgs = (long long)per_cu * (sw_wgs() / 2);
    if (wgs > p.rows / SL_NX) wgs = p.rows / SL_NX;
    hipLaunchKernelGGL(conv_stem_wgrad_lds_kernel<OZ>, dim3((unsigned)wgs), dim3(SW_THREADS), lds, s, p);
    return wgs;
}

}  // namespace

extern "C" {

size_t mdt_conv_stem_wgrad_workspace_bytes(int c_out, int k)
{
    if (c_out <= 0 || k <= 0) return 0;
    return (size_t)sw_wgs() * c_out * k * k * k * sizeof(float) + 256;
}

int mdt_conv_stem_wgrad(const float *grad_out, const float *x_padded, float *grad_weight, int batch, int OY, int OX, int OZ,
                        int c_out, int k, int sy, int sx, int YP, int XP, int ZP, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!grad_out || !x_padded || !grad_weight || batch <= 0 || OY <= 0 || OX <= 0 || OZ <= 0) return MDT_ERR_INVALID_ARGUMENT;
    const int T = k * k * k;
    if (c_out < 1 || c_out > 32 || k < 1 || (k & 1) == 0 || T > 4 * SW_NTW * 32 || OZ % (2 * SW_UNROLL) != 0 || sy < 1 || sx < 1)
        return MDT_ERR_UNSUPPORTED;
    // the padded volume must hold every tap of every output voxel
    if (YP < (OY - 1) * sy + k || XP < (OX - 1) * sx + k || ZP < OZ - 1 + k) return MDT_ERR_INVALID_ARGUMENT;
    if ((long long)YP * XP * ZP > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    const long long rows = (long long)batch * OY * OX;
    long long n_wg = sw_wgs();
    if (n_wg > rows) n_wg = rows;
    if (!workspace || workspace_bytes < (size_t)n_wg * c_out * T * sizeof(float)) return MDT_ERR_WORKSPACE_TOO_SMALL;
    SWParams p;
    p.dy = grad_out; p.xp = x_padded; p.ws = static_cast<float *>(workspace);
    p.B = batch; p.OY = OY; p.OX = OX; p.OZ = OZ; p.Co = c_out;
    p.YP = YP; p.XP = XP; p.ZP = ZP; p.sy = sy; p.sx = sx; p.k = k; p.T = T; p.rows = rows;
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (k == SL_K && sy == 2 && sx == 2 && OX % SL_NX == 0 && ZP == OZ + SL_K - 1 && (OZ == 128 || OZ == 64 || OZ == 32)
        && XP >= 2 * OX + SL_K - 2 && rows / SL_NX <= 0x7fffffffLL) {
        n_wg = OZ == 128 ? launch_lds<128>(p, s) : OZ == 64 ? launch_lds<64>(p, s) : launch_lds<32>(p, s);
    } else {
        hipLaunchKernelGGL(conv_stem_wgrad_partial_kernel, dim3((unsigned)n_wg), dim3(SW_THREADS), 0, s, p);
    }
    const int n_elem = c_out * T;
    hipLaunchKernelGGL(conv_stem_wgrad_reduce_kernel, dim3((unsigned)((n_elem + 15) / 16)), dim3(256), 0, s, p.ws, grad_weight, n_elem, (int)n_wg);
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

}  // extern "C"
