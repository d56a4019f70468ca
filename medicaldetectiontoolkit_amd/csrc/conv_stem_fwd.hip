// conv_stem_fwd.hip -- forward of the ResNet stem on gfx950: a 7 x 7 x 7 convolution of a ONE-channel volume with stride
// (2, 2, 1) and padding 3 (models/backbone.py:66-68: C1 = conv(1 -> 18, ks 7, stride (2, 2, 1), pad 3)):
//     out[b, oy, ox, oz][co] = sum_{ky, kx, kz} w[co][ky, kx, kz] * Xpad[b, 2 oy + ky, 2 ox + kx, oz + kz]
// 52 GFLOP on 8 x 128^3; MIOpen takes 1.96 ms for it even in space-to-depth form (profiles/r03_op_profile.txt), 3.1 ms as is.
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32): M = 32 consecutive oz of one output column (b, oy, ox), N = output channels (<= 32),
// K = the 343 taps in pairs (172 MFMAs per tile; the 344th tap has a zero weight).
//   * B (weights) never changes: every lane keeps its 172 B values in REGISTERS for the lifetime of the (persistent) workgroup.
//   * A (input): a workgroup stages the 7 x 13 input lines (all z) that its four output columns (4 consecutive ox, one per
//     wave) read -- one contiguous run of 13 lines per ky in the caller-padded volume -- in LDS; an A fragment is then 32
//     consecutive floats per half-wave (conflict-free ds_read_b32), at a compile-time offset per tap.
//   * two z tiles per wave are in flight (independent accumulators), the LDS reads of the next group of taps are issued before
//     the MFMAs of the current one (two operand register sets).
//   * the 32 x C_out result of a tile is a CONTIGUOUS run of the channels-last output: it goes through a per-wave LDS tile and
//     leaves as full 256-byte stores.
// Optional epilogue: + bias[co], ReLU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mdt_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SF_THREADS = 256;
constexpr int SF_K = 7;
constexpr int SF_T = SF_K * SF_K * SF_K;        // 343 taps
constexpr int SF_PAIRS = (SF_T + 1) / 2;        // 172 MFMAs per tile
constexpr int SF_NX = 4;                        // output columns (consecutive ox) per workgroup = waves
constexpr int SF_G = 4;                         // MFMAs per software-pipeline group
constexpr int SF_GROUPS = SF_PAIRS / SF_G;      // 43
static_assert(SF_GROUPS * SF_G == SF_PAIRS, "tap pairs must split into whole groups");
constexpr int SF_COLS = 2 * (SF_NX - 1) + SF_K; // 13 input lines in x per ky

struct SFParams {
    const float *xp;      // [B, YP, XP, OZ + 6] zero-padded one-channel input
    const float *w;       // [Co][343]
    const float *bias;    // [Co] or null
    float *out;           // [B, OY, OX, OZ, Co]
    int B, OY, OX, Co, YP, XP, relu;
    int passes;           // B * OY * OX / 4
};

// K slots.  A half-wave h = 1 must read the SECOND tap of its pair; a per-lane select of the address would cost a register per
// pair (the compiler hoists it), so the 343 taps are paired such that the second tap lies at one of three CONSTANT distances
// from the first -- +1 float (next kz), +ZP (next kx), +RS (next ky) -- and each distance has its own per-lane base pointer:
//   slots   0..146: (ky, kx, kz = 2 j) + (ky, kx, 2 j + 1)                     j = 0..2, all 49 (ky, kx)
//   slots 147..167: (ky, kx = 2 j, kz = 6) + (ky, 2 j + 1, 6)                  j = 0..2, all ky
//   slots 168..170: (ky = 2 j, 6, 6) + (2 j + 1, 6, 6)                         j = 0..2
//   slot       171: (6, 6, 6) alone (the h = 1 half has a zero weight and reads the same address)
__host__ __device__ constexpr int slot_tap0(int s)
{
    if (s < 147) return (s / 3) * SF_K + 2 * (s % 3);
    if (s < 168) return (((s - 147) / 3) * SF_K + 2 * ((s - 147) % 3)) * SF_K + 6;
    if (s < 171) return ((2 * (s - 168)) * SF_K + 6) * SF_K + 6;
    return SF_T - 1;
}
__host__ __device__ constexpr int slot_tap1(int s)      // -1: none
{
    if (s < 147) return slot_tap0(s) + 1;
    if (s < 168) return slot_tap0(s) + SF_K;
    if (s < 171) return slot_tap0(s) + SF_K * SF_K;
    return -1;
}
__host__ __device__ constexpr int slot_kind(int s) { return s < 147 ? 0 : (s < 168 ? 1 : (s < 171 ? 2 : 3)); }

template <int ZP>
__host__ __device__ constexpr int tap_offset(int t)
{
    return (t / (SF_K * SF_K)) * (SF_COLS * ZP) + ((t % (SF_K * SF_K)) / SF_K) * ZP + t % SF_K;
}

template <int OZ>
__global__ __launch_bounds__(SF_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_stem_fwd_kernel(SFParams p)
{
    constexpr int ZP = OZ + SF_K - 1;
    constexpr int RS = SF_COLS * ZP;            // floats per ky run (even: OZ is even)
    constexpr int TZ = 1;                       // z tiles in flight per wave (2 would not fit the 256 registers of two waves per SIMD next to the 172 B registers)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *img = lds;                           // [7][13][ZP]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci = lane & 31, h = lane >> 5;
    float *stage = lds + SF_K * RS + wave * 32 * p.Co;       // [32][Co] per wave

    // B operand: lane (k = h, j = ci) of slot s holds w[ci][tap h of the slot]
    float wreg[SF_PAIRS];
#pragma unroll
    for (int s = 0; s < SF_PAIRS; ++s) {
        const int t = h ? slot_tap1(s) : slot_tap0(s);
        wreg[s] = (t >= 0 && ci < p.Co) ? p.w[ci * SF_T + t] : 0.0f;
    }
    const float bias = (p.bias && ci < p.Co) ? p.bias[ci] : 0.0f;
    const float *lane_img = img + 2 * wave * ZP + ci;        // + z0 + tap offset
    const float *lane_base[4] = {lane_img + h, lane_img + h * ZP, lane_img + h * RS, lane_img};   // by slot kind

    const int oxg_n = p.OX / SF_NX;
    const int per = (p.passes + gridDim.x - 1) / gridDim.x;
    const int pass_end = min(p.passes, (int)(blockIdx.x + 1) * per);
    for (int pass = blockIdx.x * per; pass < pass_end; ++pass) {
        const int oxg = pass % oxg_n;
        const int r2 = pass / oxg_n;
        const int oy = r2 % p.OY, b = r2 / p.OY;
        __syncthreads();                                     // the previous image is no longer read
        {   // LDS-DMA (no registers, every chunk in flight at once): 7 contiguous runs of 13 lines
            constexpr int CH = (RS + 63) / 64;               // 64-float chunks per run
            const float *src0 = p.xp + (((long long)b * p.YP + 2 * oy) * p.XP + 2 * SF_NX * oxg) * ZP;
            const long long ystride = (long long)p.XP * ZP;
            for (int c = wave; c < SF_K * CH; c += SF_THREADS / 64) {
                const int ky = c / CH, cc = c - ky * CH;
                if (cc * 64 + lane < RS)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src0 + ky * ystride + cc * 64 + lane),
                                                     (__attribute__((address_space(3))) void *)(img + ky * RS + cc * 64), 4, 0, 0);
            }
        }
        __syncthreads();
        const int ox = oxg * SF_NX + wave;
        float *orow = p.out + (((long long)b * p.OY + oy) * p.OX + ox) * (long long)OZ * p.Co;
        for (int zt = 0; zt < OZ / 32; zt += TZ) {
            f32x16 acc[TZ];
#pragma unroll
            for (int z = 0; z < TZ; ++z)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[z][r] = 0.0f;
            float a[2][SF_G][TZ];
            auto load_group = [&](int g, float (&d)[SF_G][TZ]) {
#pragma unroll
                for (int u = 0; u < SF_G; ++u) {
                    const int s = g * SF_G + u;
                    const float *q = lane_base[slot_kind(s)] + zt * 32 + tap_offset<ZP>(slot_tap0(s));
#pragma unroll
                    for (int z = 0; z < TZ; ++z) d[u][z] = q[32 * z];
                }
            };
            auto mfma_group = [&](int g, const float (&d)[SF_G][TZ]) {
#pragma unroll
                for (int u = 0; u < SF_G; ++u)
#pragma unroll
                    for (int z = 0; z < TZ; ++z) acc[z] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[u][z], wreg[g * SF_G + u], acc[z], 0, 0, 0);
            };
            load_group(0, a[0]);
#pragma unroll
            for (int g = 0; g < SF_GROUPS; ++g) {
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < SF_GROUPS) load_group(g + 1, a[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(g, a[g & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // C/D map: column = lane & 31 (channel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (voxel within the tile)
#pragma unroll
            for (int z = 0; z < TZ; ++z) {
                if (ci < p.Co) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[z][r] + bias;
                        if (p.relu) v = fmaxf(v, 0.0f);
                        stage[((r & 3) + 8 * (r >> 2) + 4 * h) * p.Co + ci] = v;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                float *o = orow + (long long)(zt + z) * 32 * p.Co;
                for (int q = lane; q < 32 * p.Co; q += 64) o[q] = stage[q];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
}

inline int sf_cus()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        (void)hipGetLastError();
        n = cus;
    }
    return n;
}

template <int OZ>
int launch_stem_fwd(const SFParams &p, hipStream_t s)
{
    constexpr int ZP = OZ + SF_K - 1;
    const size_t lds = ((size_t)SF_K * SF_COLS * ZP + (size_t)4 * 32 * p.Co) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)conv_stem_fwd_kernel<OZ>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr = true;
    }
    if (lds > 80 * 1024) return MDT_ERR_UNSUPPORTED;
    int n_wg = 2 * sf_cus();
    if (n_wg > p.passes) n_wg = p.passes;
    (void)hipGetLastError();
    hipLaunchKernelGGL(conv_stem_fwd_kernel<OZ>, dim3((unsigned)n_wg), dim3(SF_THREADS), lds, s, p);
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

}  // namespace

extern "C" {

int mdt_conv_stem_forward_supported(int OY, int OX, int OZ, int c_out, int k, int sy, int sx)
{
    return (k == SF_K && sy == 2 && sx == 2 && c_out >= 1 && c_out <= 32 && OY > 0 && OX > 0 && OX % SF_NX == 0 && (OZ == 128 || OZ == 64 || OZ == 32)) ? 1 : 0;
}

int mdt_conv_stem_forward(const float *x_padded, const float *weight, const float *bias, float *out, int batch, int OY, int OX, int OZ,
                          int c_out, int k, int sy, int sx, int YP, int XP, int ZP, int relu, void *stream)
{
    if (!x_padded || !weight || !out || batch <= 0) return MDT_ERR_INVALID_ARGUMENT;
    if (!mdt_conv_stem_forward_supported(OY, OX, OZ, c_out, k, sy, sx)) return MDT_ERR_UNSUPPORTED;
    // the padded volume must hold every tap of every output voxel, and the 13-line run of the last column group
    if (YP < (OY - 1) * sy + k || XP < (OX - 1) * sx + k || ZP != OZ + k - 1) return MDT_ERR_INVALID_ARGUMENT;
    if ((reinterpret_cast<uintptr_t>(x_padded) & 7) != 0) return MDT_ERR_INVALID_ARGUMENT;
    if ((long long)batch * OY * (OX / SF_NX) > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    SFParams p;
    p.xp = x_padded; p.w = weight; p.bias = bias; p.out = out;
    p.B = batch; p.OY = OY; p.OX = OX; p.Co = c_out; p.YP = YP; p.XP = XP; p.relu = relu ? 1 : 0;
    p.passes = batch * OY * (OX / SF_NX);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (OZ == 128) return launch_stem_fwd<128>(p, s);
    if (OZ == 64) return launch_stem_fwd<64>(p, s);
    return launch_stem_fwd<32>(p, s);
}

}  // extern "C"
