// conv1x1_wgrad.hip -- weight gradient of the 1x1(x1) convolutions of the ResNet/FPN bottlenecks on gfx950:
//     dW[co][ci] = sum_v dY[v][co] * X[v][ci]        v = every voxel of the batch (channels-last storage: rows of C floats)
// i.e. a [Cout x V] x [V x Cin] product with V ~ 10^5..10^6 and 18..288 channels: 2.7 GFLOP against 377 MB for the
// 18 -> 72 layer on the 8 x 32x32x128 maps -- HBM-bound (47 us at 8 TB/s), and MIOpen's backward-weights solvers take
// 310-434 us for it (profiles/r02_conv_reformulation_probes.txt; 19 ms of the 59 ms training step are backward-weights).
//
// Design: split V over all waves of the chip; every wave streams its voxels ONCE and accumulates its tiles of dW with the
// fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32, 157 TF/s -- the contraction is 30x below that bound).  The operand layout of
// that instruction is one element per lane, lane l <-> (channel l & 31, voxel l >> 5): with channels-last rows a wave's load
// IS the fragment (two 128-byte runs), so operands go global -> VGPR -> MFMA with no LDS staging and no transposes.  Lanes
// beyond the channel count read a clamped (valid) address: they only feed rows / columns of the tile that are never stored.
// 8 voxels (4 K-steps) of loads form a trip; two operand register sets keep the next trip's loads in flight under the MFMAs.
// Reduction: the 4 waves of a workgroup add their tiles through LDS in wave order, every workgroup writes one [Cout][Cin]
// partial, and a second small kernel adds the partials in a fixed order -- deterministic, no atomics.
//
// HBM-bound; algorithmic bytes 4 * V * (Cout + Cin) (+ the partials: n_wg * Cout * Cin * 8).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mdt_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int W_THREADS = 256;       // 4 waves per workgroup
constexpr int W_UNROLL = 4;          // K-steps (of 2 voxels) per trip; two trips' operands are live (double buffering)
constexpr int W_MAX_TILES = 4;       // 32 x 32 accumulator tiles per wave (64 VGPRs)

// MT x NT tiles per wave: rows = output channels (dY), columns = input channels (X)
template <int MT, int NT>
__global__ __launch_bounds__(W_THREADS) void conv1x1_wgrad_partial_kernel(const float *__restrict__ dY, const float *__restrict__ X,
                                                                        float *__restrict__ ws, long long V, int Cout, int Cin,
                                                                        int groups_n)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [4 waves][MT * NT][1024]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: block addresses stay in SGPRs
    const int gm = blockIdx.y / groups_n, gn = blockIdx.y - gm * groups_n;
    const int co0 = gm * MT * 32, ci0 = gn * NT * 32;
    // block-cyclic split of the voxels: trip t of wave w takes the 8 voxels [(t * n_waves + w) * 8, +8), so at any moment all
    // waves of the chip read ONE contiguous window (a private contiguous chunk per wave makes 2048 waves advance in lockstep
    // at a fixed 147 KB stride: 1.4 TB/s, HBM channel camping)
    const long long wave_id = (long long)blockIdx.x * (W_THREADS / 64) + wave;
    const long long n_waves = (long long)gridDim.x * (W_THREADS / 64);
    const long long n_blocks = V / (2 * W_UNROLL);                 // full blocks of 2 * W_UNROLL voxels
    const int ch = lane & 31, kk = lane >> 5;
    // per-lane 32-bit offsets inside a 2-voxel K-step (channel clamped: see the header); the K-step's base address is wave-uniform
    unsigned offa[MT], offb[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) offa[m] = (unsigned)(kk * Cout + min(co0 + 32 * m + ch, Cout - 1));
#pragma unroll
    for (int n = 0; n < NT; ++n) offb[n] = (unsigned)(kk * Cin + min(ci0 + 32 * n + ch, Cin - 1));
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    // two operand register sets: the loads of the NEXT block are issued before the MFMAs of the current one, so the memory
    // pipe works while the wave computes (a single set leaves it idle for the ~1500 MFMA cycles of every trip)
    auto load_block = [&](long long blk, float (&a)[W_UNROLL][MT], float (&b)[W_UNROLL][NT]) {
        const long long v = blk * (2 * W_UNROLL);
#pragma unroll
        for (int u = 0; u < W_UNROLL; ++u) {
            const float *ra = dY + (v + 2 * u) * Cout;       // scalar base + per-lane 32-bit offset
            const float *rb = X + (v + 2 * u) * Cin;
#pragma unroll
            for (int m = 0; m < MT; ++m) a[u][m] = ra[offa[m]];
#pragma unroll
            for (int n = 0; n < NT; ++n) b[u][n] = rb[offb[n]];
        }
    };
    auto mfma_block = [&](const float (&a)[W_UNROLL][MT], const float (&b)[W_UNROLL][NT]) {
#pragma unroll
        for (int u = 0; u < W_UNROLL; ++u)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][m], b[u][n], acc[m][n], 0, 0, 0);
    };
    {
        float a0[W_UNROLL][MT], b0[W_UNROLL][NT], a1[W_UNROLL][MT], b1[W_UNROLL][NT];
        long long blk = wave_id;
        if (blk < n_blocks) load_block(blk, a0, b0);
        while (blk < n_blocks) {
            long long nb = blk + n_waves;
            if (nb < n_blocks) load_block(nb, a1, b1);
            mfma_block(a0, b0);
            blk = nb;
            if (blk >= n_blocks) break;
            nb = blk + n_waves;
            if (nb < n_blocks) load_block(nb, a0, b0);
            mfma_block(a1, b1);
            blk = nb;
        }
    }
    if (wave_id == 0) {               // the voxels past the last full block: rows past the end contribute zeros
        for (long long v = n_blocks * (2 * W_UNROLL); v < V; v += 2) {
            const bool ok = v + kk < V;
            const float *ra = dY + (ok ? v : v - 1) * Cout;      // v - 1 + kk == v: a valid row for the lanes past the end (zeroed below)
            const float *rb = X + (ok ? v : v - 1) * Cin;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float av = ok ? ra[offa[m]] : 0.0f;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float bv = ok ? rb[offb[n]] : 0.0f;
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m][n], 0, 0, 0);
                }
            }
        }
    }
    // the 4 waves of the workgroup: tiles -> LDS, summed in wave order, one [Cout][Cin] partial per workgroup
    float *mine = lds + (size_t)wave * (MT * NT * 1024);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(m * NT + n) * 1024 + r * 64 + lane] = acc[m][n][r];
    __syncthreads();
    float *out = ws + (size_t)blockIdx.x * Cout * Cin;
    for (int idx = tid; idx < MT * NT * 1024; idx += W_THREADS) {
        const int t = idx >> 10, rem = idx & 1023;
        const int r = rem >> 6, l = rem & 63;
        const int m = t / NT, n = t - m * NT;
        const int co = co0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);       // C/D map of the 32x32 MFMA
        const int ci = ci0 + 32 * n + (l & 31);
        if (co < Cout && ci < Cin) {
            float s = lds[idx];
#pragma unroll
            for (int w = 1; w < W_THREADS / 64; ++w) s = s + lds[(size_t)w * (MT * NT * 1024) + idx];
            out[(size_t)co * Cin + ci] = s;
        }
    }
}

// dW[e] = sum over the partials, in a FIXED order: 16 lanes per element each add every 16th partial (8 loads in flight per
// trip: a thread summing all partials one dependent load after the other is latency-bound -- 512 partials took 250 us), then
// the 16 lane sums are added lane 0 .. 15 by one lane
__global__ __launch_bounds__(256) void conv1x1_wgrad_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dW, int n_elem, int n_part)
{
    __shared__ float part[16][17];
    const int sub = threadIdx.x & 15, el = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    float s = 0.0f;
    if (e < n_elem) {
        int p = sub;
        for (; p + 16 * 7 < n_part; p += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(p + 16 * u) * n_elem + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s = s + v[u];
        }
        for (; p < n_part; p += 16) s = s + ws[(size_t)p * n_elem + e];
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

struct Plan {
    int mt, nt, groups_m, groups_n, n_wg;
    long long vox_per_wave;
};

inline int cu_count()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        (void)hipGetLastError();
    }
    return n;
}

bool make_plan(long long V, int Cout, int Cin, Plan &p)
{
    if (V <= 0 || Cout <= 0 || Cin <= 0 || Cout > 4096 || Cin > 4096) return false;
    const int tm = (Cout + 31) / 32, tn = (Cin + 31) / 32;
    // tiles per wave: as many of the smaller operand's tiles as fit, so that the larger operand is streamed once
    int mt, nt;
    if (tm * tn <= W_MAX_TILES) { mt = tm; nt = tn; }
    else if (tn <= 2) { nt = tn; mt = W_MAX_TILES / nt; }
    else if (tm <= 2) { mt = tm; nt = W_MAX_TILES / mt; }
    else { mt = 2; nt = 2; }
    p.mt = mt; p.nt = nt;
    p.groups_m = (tm + mt - 1) / mt;
    p.groups_n = (tn + nt - 1) / nt;
    // workgroups per CU, never more than one block per wave would justify
    const int per_cu = (mt * nt == 1 ? 4 : 2);      // measured: 18 -> 18 59 vs 87 us; 36 -> 144 65 vs 81 us
    long long n_wg = (long long)cu_count() * per_cu;
    const long long blocks = (V + 2 * W_UNROLL - 1) / (2 * W_UNROLL);
    const long long max_wg = (blocks + (W_THREADS / 64) - 1) / (W_THREADS / 64);
    if (n_wg > max_wg) n_wg = max_wg;
    if (n_wg < 1) n_wg = 1;
    p.n_wg = (int)n_wg;
    p.vox_per_wave = 0;
    return true;
}

template <int MT, int NT>
void launch_partial(const Plan &p, const float *dY, const float *X, float *ws, long long V, int Cout, int Cin, hipStream_t s)
{
    const size_t lds = (size_t)(W_THREADS / 64) * MT * NT * 1024 * sizeof(float);
    auto kernel = conv1x1_wgrad_partial_kernel<MT, NT>;
    static bool optin = false;
    if (!optin && lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipGetLastError();
        optin = true;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)p.n_wg, (unsigned)(p.groups_m * p.groups_n)), dim3(W_THREADS), lds, s,
                       dY, X, ws, V, Cout, Cin, p.groups_n);
}

}  // namespace

extern "C" {

size_t mdt_conv1x1_wgrad_workspace_bytes(long long n_voxels, int c_out, int c_in)
{
    Plan p;
    if (!make_plan(n_voxels, c_out, c_in, p)) return 0;
    return (size_t)p.n_wg * c_out * c_in * sizeof(float) + 256;
}

int mdt_conv1x1_wgrad(const float *grad_out, const float *x, float *grad_weight, long long n_voxels, int c_out, int c_in,
                      void *workspace, size_t workspace_bytes, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    Plan p;
    if (!grad_out || !x || !grad_weight) return MDT_ERR_INVALID_ARGUMENT;
    if (!make_plan(n_voxels, c_out, c_in, p)) return MDT_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < (size_t)p.n_wg * c_out * c_in * sizeof(float)) return MDT_ERR_WORKSPACE_TOO_SMALL;
    float *ws = static_cast<float *>(workspace);
    (void)hipGetLastError();
    const int key = p.mt * 10 + p.nt;
    switch (key) {
    case 11: launch_partial<1, 1>(p, grad_out, x, ws, n_voxels, c_out, c_in, s); break;
    case 21: launch_partial<2, 1>(p, grad_out, x, ws, n_voxels, c_out, c_in, s); break;
    case 31: launch_partial<3, 1>(p, grad_out, x, ws, n_voxels, c_out, c_in, s); break;
    case 41: launch_partial<4, 1>(p, grad_out, x, ws, n_voxels, c_out, c_in, s); break;
    case 12: launch_partial<1, 2>(p, grad_out, x, ws, n_voxels, c_out, c_in, s); break;
    case 13: launch_partial<1, 3>(p, grad_out, x, ws, n_voxels, c_out, c_in, s); break;
    case 14: launch_partial<1, 4>(p, grad_out, x, ws, n_voxels, c_out, c_in, s); break;
    case 22: launch_partial<2, 2>(p, grad_out, x, ws, n_voxels, c_out, c_in, s); break;
    default: return MDT_ERR_UNSUPPORTED;
    }
    const int n_elem = c_out * c_in;
    hipLaunchKernelGGL(conv1x1_wgrad_reduce_kernel, dim3((unsigned)((n_elem + 15) / 16)), dim3(256), 0, s, ws, grad_weight, n_elem, p.n_wg);
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

}  // extern "C"
