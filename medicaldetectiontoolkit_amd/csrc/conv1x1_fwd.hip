// conv1x1_fwd.hip -- 1x1(x1) convolution forward WITH its epilogue, for the bottleneck layers of the ResNet backbone on the large maps (gfx950).
//
//   out[v][n] = act( (sum_k x[v][k] * w[n][k] + bias[n]) (+ res[v][n]) )          channels-last rows, k ascending
//
// The reference's ResBlock (models/backbone.py:197-206) runs conv1 (1x1, C -> C/4) + ReLU, conv2 (3x3) + ReLU, conv3 (1x1, C/4 -> C) + residual + ReLU.
// With the 1x1 layers on MIOpen / CK and the bias / residual / ReLU in mdt_bias_act_forward, conv3 of a C2 block at the benchmark patch (8 x 128^3:
// 131072 voxels x 8, 18 -> 72 channels) is a 111 us convolution that writes 302 MB and a 157 us epilogue that reads it back with the 302 MB residual
// and writes it again: 1.28 GB of traffic for a layer whose operands are 75 + 302 MB in and 302 MB out.  Here the product runs on the matrix cores
// inside the pass that streams the operands once: the input rows of 32 voxels go through the wave's LDS slot (one contiguous run, 16-byte loads),
// the filter lives in registers for the wave's lifetime, bias / residual / ReLU are applied to the accumulators on their way out.
// HBM-bound (2 * K MACs per output float at K = 18 .. 72 is 2-9 % of the fp32 MFMA peak at the HBM rate); fp32 MFMA: exact products, fixed order.
//
// The accumulator layout of v_mfma_f32_32x32x2_f32 (column = lane & 31 = output channel, 16 rows per lane) makes every wave-level load of the residual
// and store of the result two 128-byte row segments -- the same data path as conv1x1_dgrad_add_mfma_kernel (epilogue.hip), whose shape this is.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mdt_hip.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

inline int c1_check()
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

// KS: K-steps of two input channels (K = 2 KS), NT: 32-column tiles of output channels (N <= 32 NT), WPB: waves per block (LDS: WPB slots of 32 x (K + 1)),
// BL: the filter lives in LDS instead of registers (36 -> 144: 90 filter registers beside 80 residual registers left ONE wave per SIMD, 66 us; LDS bandwidth is
// nowhere near a limit at 23 KB of fragment reads per tile)
template <int KS, int NT, int WPB, bool BL, bool RES, bool RELU>
__global__ __launch_bounds__(64 * WPB) void conv1x1_fwd_mfma_kernel(float *__restrict__ out, const float *__restrict__ x, const float *__restrict__ w,
                                                                    const float *__restrict__ bias, const float *res, long long V, int N)
{
    constexpr int K = 2 * KS;
    constexpr int ASTR = K + 1;                          // odd row stride: the 32 rows of a fragment read hit 32 different banks
    __shared__ float s_a[WPB][32 * ASTR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    constexpr int NB = NT * 32;                          // (NB % 64 == 32 at NT = 5: the two half-waves of a fragment read hit disjoint banks)
    __shared__ float s_b[BL ? K * NB : 1];
    float bfrag[BL ? 1 : KS][BL ? 1 : NT], bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 32 + col;
        bv[nt] = n < N ? bias[n] : 0.0f;
        if (!BL) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bfrag[ks][nt] = n < N ? w[(long long)n * K + 2 * ks + half] : 0.0f;
        }
    }
    if (BL) {
        for (int e = threadIdx.x; e < K * NB; e += 64 * WPB) {
            const int k = e / NB, n = e - k * NB;
            s_b[e] = n < N ? w[(long long)n * K + k] : 0.0f;
        }
        __syncthreads();
    }
    const long long tiles = (V + 31) / 32;
    float *sa = s_a[wave];
    for (long long tile = (long long)blockIdx.x * WPB + wave; tile < tiles; tile += (long long)gridDim.x * WPB) {
        const long long v0 = tile * 32;
        const int nv = (int)min((long long)32, V - v0);
        // A: the tile's input rows, one contiguous run of nv * K floats -> LDS (row stride ASTR); 16-byte loads (32 * K floats per tile: the run starts
        // 16-byte aligned).  Full tiles: the input run AND the residual rows are requested together, before anything waits -- a wave has its whole
        // tile (2.3 + 9.2 KB at 18 -> 72) in flight at once; requested one after the other the layer was latency-bound (237 us, 2.9 TB/s).
        const v4f *src = reinterpret_cast<const v4f *>(x + v0 * K);
        float rv[NT][16];
        if (nv == 32) {
            constexpr int NA = (8 * K + 63) / 64;        // 16-byte loads per lane
            v4f areg[NA];
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int e4 = lane + 64 * i;
                areg[i] = (e4 < 8 * K) ? src[e4] : v4f{0.f, 0.f, 0.f, 0.f};
            }
            if (RES) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = nt * 32 + col;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                        rv[nt][r] = (n < N) ? res[(v0 + row) * N + n] : 0.0f;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int e4 = lane + 64 * i;
                if (e4 < 8 * K) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = e4 * 4 + j, m = e / K, k = e - m * K;
                        sa[m * ASTR + k] = areg[i][j];
                    }
                }
            }
        } else {
            const int n4 = (nv * K) >> 2;                // K is even: a 2-float tail is handled below
            for (int e4 = lane; e4 < n4; e4 += 64) {
                const v4f v = src[e4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = e4 * 4 + j, m = e / K, k = e - m * K;
                    sa[m * ASTR + k] = v[j];
                }
            }
            for (int e = n4 * 4 + lane; e < 32 * K; e += 64) {
                const int m = e / K, k = e - m * K;
                sa[m * ASTR + k] = e < nv * K ? x[v0 * K + e] : 0.0f;
            }
            if (RES) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = nt * 32 + col;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                        rv[nt][r] = (n < N && row < nv) ? res[(v0 + row) * N + n] : 0.0f;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float afrag[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) afrag[ks] = sa[col * ASTR + 2 * ks + half];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 32 + col;
            v16f acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float bf = BL ? s_b[(2 * ks + half) * NB + nt * 32 + col] : bfrag[BL ? 0 : ks][BL ? 0 : nt];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[ks], bf, acc, 0, 0, 0);
            }
            if (n < N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    float v = acc[r] + bv[nt];           // (conv + bias) + residual: the order of the reference's graph
                    if (RES) v = v + rv[nt][r];
                    if (RELU) v = v > 0.0f ? v : 0.0f;
                    if (row < nv) out[(v0 + row) * N + n] = v;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                 // the slot is rewritten by the next tile
    }
}

// ---- backward of the same layer's epilogue AND its input gradient in one pass (round 6) ------------------------------------------------------------
//   g[v][n] = gy[v][n] * (y[v][n] > 0)   (y == null: g = gy, nothing stored),   gx[v][k] = sum_n g[v][n] * w[n][k],   gbias[n] = sum_v g[v][n]
// conv3 of a C2 block (18 -> 72, + residual + ReLU): the ReLU-mask / bias-gradient pass (mdt_bias_act_backward: reads gy and y, writes g, 906 MB) was
// followed by the input-gradient convolution reading g again (CK, 302 + 75 MB).  Here the masked rows of 32 voxels go to memory AND through the wave's
// LDS slot into the matrix cores (A = g rows, B = the filter as it is stored, [c_out][c_in]); the bias gradient is the column sum of the same LDS tile,
// kept per wave in registers and written as one partial row per wave (folded in a fixed order by the second stage: deterministic).
// R = 2 KS output channels of the layer (the reduction of this product), C <= 32 input channels.
template <int KS, int WPB, bool RELU>
__global__ __launch_bounds__(64 * WPB) void conv1x1_bwd_mfma_kernel(float *__restrict__ g, float *__restrict__ gx, float *__restrict__ partial,
                                                                    const float *__restrict__ gy, const float *__restrict__ y, const float *__restrict__ w,
                                                                    long long V, int C)
{
    constexpr int R = 2 * KS;
    constexpr int ASTR = R + 1;
    constexpr int NA = (8 * R + 63) / 64;
    __shared__ float s_a[WPB][32 * ASTR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    float bfrag[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bfrag[ks] = col < C ? w[(long long)(2 * ks + half) * C + col] : 0.0f;
    float bsum0 = 0.0f, bsum1 = 0.0f;                    // channels lane and lane + 64
    const long long tiles = (V + 31) / 32;
    float *sa = s_a[wave];
    for (long long tile = (long long)blockIdx.x * WPB + wave; tile < tiles; tile += (long long)gridDim.x * WPB) {
        const long long v0 = tile * 32;
        const int nv = (int)min((long long)32, V - v0);
        const int n4 = (nv * R) >> 2;                    // R is even; nv * R % 4 != 0 only when R % 4 == 2 and nv odd: scalar tail below
        const v4f *sg = reinterpret_cast<const v4f *>(gy + v0 * R);
        const v4f *sy = reinterpret_cast<const v4f *>(y + v0 * R);
        v4f *dg = reinterpret_cast<v4f *>(g + v0 * R);
        v4f ga[NA], ya[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e4 = lane + 64 * i;
            ga[i] = (e4 < n4) ? sg[e4] : v4f{0.f, 0.f, 0.f, 0.f};
            if (RELU) ya[i] = (e4 < n4) ? sy[e4] : v4f{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e4 = lane + 64 * i;
            if (RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ga[i][j] = ya[i][j] > 0.0f ? ga[i][j] : 0.0f;
                if (e4 < n4) dg[e4] = ga[i];
            }
            if (e4 < 8 * R) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = e4 * 4 + j, m = e / R, k = e - m * R;
                    sa[m * ASTR + k] = ga[i][j];             // (rows past nv: zeros)
                }
            }
        }
        for (int e = n4 * 4 + lane; e < nv * R; e += 64) {   // at most a 2-float tail of the last tile
            float v = gy[v0 * R + e];
            if (RELU) { v = y[v0 * R + e] > 0.0f ? v : 0.0f; g[v0 * R + e] = v; }
            const int m = e / R, k = e - m * R;
            sa[m * ASTR + k] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float afrag[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) afrag[ks] = sa[col * ASTR + 2 * ks + half];
        // bias gradient: column sums of the tile, rows ascending
        {
            float s0 = 0.0f, s1 = 0.0f;
            const int c1 = (lane + 64 < R) ? lane + 64 : lane;
#pragma unroll 8
            for (int m = 0; m < 32; ++m) { s0 = s0 + sa[m * ASTR + lane]; s1 = s1 + sa[m * ASTR + c1]; }
            bsum0 = bsum0 + s0;
            bsum1 = bsum1 + s1;
        }
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[ks], bfrag[ks], acc, 0, 0, 0);
        if (col < C) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < nv) gx[(v0 + row) * C + col] = acc[r];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // the block's waves are folded in wave order (LDS slot rows are free now), one value per (channel, block): partial[c][block] -- contiguous per channel
    __syncthreads();
    float *fold = &s_a[0][0];
    if (lane < R) fold[wave * R + lane] = bsum0;
    if (lane + 64 < R) fold[wave * R + lane + 64] = bsum1;
    __syncthreads();
    for (int c = threadIdx.x; c < R; c += 64 * WPB) {
        float sum = 0.0f;
#pragma unroll
        for (int wv = 0; wv < WPB; ++wv) sum = sum + fold[wv * R + c];
        partial[(long long)c * gridDim.x + blockIdx.x] = sum;
    }
}

// gbias[n] = sum over the blocks' partials of channel n (contiguous), ascending within a thread's strided subset, then a fixed LDS tree (one block per channel)
__global__ __launch_bounds__(256) void conv1x1_bwd_bias_finish_kernel(float *__restrict__ gbias, const float *__restrict__ partial, int rows, int R)
{
    __shared__ float s_acc[256];
    const int n = blockIdx.x;
    float s = 0.0f;
    for (int j = threadIdx.x; j < rows; j += 256) s = s + partial[(long long)n * rows + j];
    s_acc[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) s_acc[threadIdx.x] = s_acc[threadIdx.x] + s_acc[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) gbias[n] = s_acc[0];
}

// ---- the RPN's two 1x1 heads on the raw output of conv_shared (round 6) ----------------------------------------------------------------------------
//   y[v][n] = sum_k relu(h[v][k] + bs[k]) * w[n][k] + b[n],   n < n_class (class logits) | n >= n_class (box deltas)
// written STRAIGHT into the level's slice of the concatenated [B][anchors of all levels][2] / [..][2 * dim] tensors (mrcnn.py:70-86 + the torch.cat of
// :1030): the reference's conv_shared epilogue (bias + ReLU: a 537 MB read and a 537 MB write on P2 at the benchmark patch), the head convolution, its bias
// pass, the two slicing copies and the two concatenations become ONE pass that reads the hidden map once.  Forward only (the training step differentiates
// the RPN losses through the sampled anchors' patches, models/mrcnn.rpn_at_anchors).  K = 2 KS hidden channels, N <= 32 head channels.
template <int KS, int WPB>
__global__ __launch_bounds__(64 * WPB) void rpn_heads_mfma_kernel(float *__restrict__ logits, float *__restrict__ deltas, const float *__restrict__ h,
                                                                  const float *__restrict__ bs, const float *__restrict__ w, const float *__restrict__ b,
                                                                  unsigned V, unsigned Vl, int ncl, int nbox, long long lstride, long long loff, long long dstride,
                                                                  long long doff)
{
    constexpr int K = 2 * KS;
    constexpr int ASTR = K + 1;
    constexpr int NA = (8 * K + 63) / 64;
    __shared__ float s_a[WPB][32 * ASTR];
    __shared__ float s_bs[K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int N = ncl + nbox;
    __shared__ float s_b[K * 32];                        // the filter, [k][n]: 64 filter registers beside the 64 row registers left one wave per SIMD
    for (int k = threadIdx.x; k < K; k += 64 * WPB) s_bs[k] = bs[k];
    for (int e = threadIdx.x; e < K * 32; e += 64 * WPB) {       // (rows of w are read as they lie: coalesced)
        const int n = e / K, k = e - n * K;
        s_b[k * 32 + n] = n < N ? w[e] : 0.0f;
    }
    const float bv = col < N ? b[col] : 0.0f;
    __syncthreads();
    const unsigned tiles = (V + 31) / 32;
    float *sa = s_a[wave];
    v4f areg[NA];
    bool first = true;
    for (unsigned tile = blockIdx.x * WPB + wave; tile < tiles; tile += gridDim.x * WPB) {
        const unsigned v0 = tile * 32;
        const int nv = (int)min(32u, V - v0);
        const int n4 = (nv * K) >> 2;                    // K % 4 == 0
        if (first) {                                     // (later tiles: requested during the previous tile's product, below)
            const v4f *src = reinterpret_cast<const v4f *>(h + (unsigned long long)v0 * K);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int e4 = lane + 64 * i;
                areg[i] = (e4 < n4) ? src[e4] : v4f{0.f, 0.f, 0.f, 0.f};
            }
            first = false;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e4 = lane + 64 * i;
            if (e4 < 8 * K) {
                const int e = e4 * 4, m = e / K, k = e - m * K;          // K % 4 == 0: the four share a row
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = areg[i][j] + s_bs[k + j];
                    sa[m * ASTR + k + j] = (e4 < n4 && v > 0.0f) ? v : 0.0f;
                }
            }
        }
        {   // the rows are in LDS, their registers are free: the NEXT tile's rows start travelling now and arrive during this tile's 64 MFMAs and
            // stores (one tile per wave in flight and nothing behind it measured 219 us on P2, 2.9 TB/s)
            const unsigned nt = tile + gridDim.x * WPB;
            if (nt < tiles) {
                const unsigned w0 = nt * 32;
                const int m4 = ((int)min(32u, V - w0) * K) >> 2;
                const v4f *src = reinterpret_cast<const v4f *>(h + (unsigned long long)w0 * K);
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int e4 = lane + 64 * i;
                    areg[i] = (e4 < m4) ? src[e4] : v4f{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[col * ASTR + 2 * ks + half], s_b[(2 * ks + half) * 32 + col], acc, 0, 0, 0);
        if (col < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < nv) {
                    const unsigned v = v0 + row, be = v / Vl, vl = v - be * Vl;
                    const float val = acc[r] + bv;
                    if (col < ncl) logits[be * lstride + loff + (long long)vl * ncl + col] = val;
                    else deltas[be * dstride + doff + (long long)vl * nbox + (col - ncl)] = val;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int KS, int NT, int WPB, bool BL>
int launch(float *out, const float *x, const float *w, const float *bias, const float *res, long long V, int N, int relu, hipStream_t s)
{
    const long long tiles = (V + 31) / 32;
    long long blocks = (tiles + WPB - 1) / WPB;
    if (blocks > 2048) blocks = 2048;
    const dim3 g((unsigned)blocks), b(64 * WPB);
    (void)hipGetLastError();
    if (res) {
        if (relu) hipLaunchKernelGGL((conv1x1_fwd_mfma_kernel<KS, NT, WPB, BL, true, true>), g, b, 0, s, out, x, w, bias, res, V, N);
        else hipLaunchKernelGGL((conv1x1_fwd_mfma_kernel<KS, NT, WPB, BL, true, false>), g, b, 0, s, out, x, w, bias, res, V, N);
    } else {
        if (relu) hipLaunchKernelGGL((conv1x1_fwd_mfma_kernel<KS, NT, WPB, BL, false, true>), g, b, 0, s, out, x, w, bias, res, V, N);
        else hipLaunchKernelGGL((conv1x1_fwd_mfma_kernel<KS, NT, WPB, BL, false, false>), g, b, 0, s, out, x, w, bias, res, V, N);
    }
    return c1_check();
}

}  // namespace

extern "C" {

// the bottleneck shapes of the LIDC backbone whose maps are large: C2 (18 <-> 72) and C3's conv3 (36 -> 144)
int mdt_conv1x1_forward_supported(int c_in, int c_out)
{
    return ((c_in == 18 && c_out == 72) || (c_in == 72 && c_out == 18) || (c_in == 36 && c_out == 144)) ? 1 : 0;
}

int mdt_conv1x1_forward(const float *x, const float *w, const float *bias, const float *res, float *out, long long n_voxels, int c_in, int c_out, int relu,
                        void *stream)
{
    if (!x || !w || !bias || !out || n_voxels < 0) return MDT_ERR_INVALID_ARGUMENT;
    if (!mdt_conv1x1_forward_supported(c_in, c_out)) return MDT_ERR_UNSUPPORTED;
    if (n_voxels == 0) return MDT_OK;
    if (((uintptr_t)x & 15) != 0 || (((uintptr_t)out | (uintptr_t)res | (uintptr_t)w | (uintptr_t)bias) & 3) != 0) return MDT_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (c_in == 18) return launch<9, 3, 4, false>(out, x, w, bias, res, n_voxels, c_out, relu, s);
    if (c_in == 36) return launch<18, 5, 4, true>(out, x, w, bias, res, n_voxels, c_out, relu, s);
    // (144 -> 36, conv1 of the C3 blocks, measured SLOWER here than CK + the epilogue kernel: 85 us against 27 + 8 us -- 18 KB of input rows per
    // wave tile, 290 registers; not served)
    return launch<36, 1, 4, false>(out, x, w, bias, res, n_voxels, c_out, relu, s);
}

// conv3 of the C2 blocks (72 output channels back to 18 inputs); the C3 shape (144 -> 36) would need 18 KB of rows per wave tile (see the forward's note)
int mdt_conv1x1_backward_supported(int c_in, int c_out) { return (c_in == 18 && c_out == 72) ? 1 : 0; }

static long long c1_bwd_blocks(long long n_voxels)
{
    const long long tiles = (n_voxels + 31) / 32;
    long long blocks = (tiles + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return blocks;
}

size_t mdt_conv1x1_backward_workspace_bytes(long long n_voxels, int c_out)
{
    if (n_voxels < 0 || c_out <= 0) return 256;
    return (size_t)c1_bwd_blocks(n_voxels) * 4 * c_out * sizeof(float) + 256;
}

int mdt_conv1x1_backward(const float *gy, const float *y, const float *w, float *g, float *gx, float *gbias, long long n_voxels, int c_in, int c_out,
                         void *workspace, size_t workspace_bytes, void *stream)
{
    if (!gy || !w || !gx || !gbias || n_voxels < 0 || (y != nullptr && g == nullptr)) return MDT_ERR_INVALID_ARGUMENT;
    if (!mdt_conv1x1_backward_supported(c_in, c_out)) return MDT_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < mdt_conv1x1_backward_workspace_bytes(n_voxels, c_out)) return MDT_ERR_WORKSPACE_TOO_SMALL;
    if ((((uintptr_t)gy | (uintptr_t)y | (uintptr_t)g) & 15) != 0) return MDT_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *partial = static_cast<float *>(workspace);
    const long long blocks = n_voxels > 0 ? c1_bwd_blocks(n_voxels) : 0;
    (void)hipGetLastError();
    if (blocks > 0) {
        if (y) hipLaunchKernelGGL((conv1x1_bwd_mfma_kernel<36, 4, true>), dim3((unsigned)blocks), dim3(256), 0, s, g, gx, partial, gy, y, w, n_voxels, c_in);
        else hipLaunchKernelGGL((conv1x1_bwd_mfma_kernel<36, 4, false>), dim3((unsigned)blocks), dim3(256), 0, s, g, gx, partial, gy, gy, w, n_voxels, c_in);
        if (c1_check() != MDT_OK) return MDT_ERR_LAUNCH_FAILED;
    }
    hipLaunchKernelGGL(conv1x1_bwd_bias_finish_kernel, dim3((unsigned)c_out), dim3(256), 0, s, gbias, partial, (int)blocks, c_out);
    return c1_check();
}

int mdt_rpn_heads_forward_supported(int hidden, int n_class, int n_box)
{
    return (hidden == 128 && n_class >= 1 && n_box >= 1 && n_class + n_box <= 32) ? 1 : 0;
}

int mdt_rpn_heads_forward(const float *h, const float *bias_shared, const float *w, const float *bias, float *logits, float *deltas, int batch,
                          long long voxels_per_element, int hidden, int n_class, int n_box, long long anchors_total, long long anchor_offset, void *stream)
{
    if (!h || !bias_shared || !w || !bias || !logits || !deltas || batch < 0 || voxels_per_element < 0 || anchors_total < 0 || anchor_offset < 0)
        return MDT_ERR_INVALID_ARGUMENT;
    if (!mdt_rpn_heads_forward_supported(hidden, n_class, n_box)) return MDT_ERR_UNSUPPORTED;
    const long long V = (long long)batch * voxels_per_element;
    if (V == 0) return MDT_OK;
    if (V >= 0x7fffffffLL / 4 || ((uintptr_t)h & 15) != 0) return MDT_ERR_UNSUPPORTED;
    // anchors per voxel A: n_class = 2 A, n_box = 2 dim A; the level's slice of element b starts at anchor b * anchors_total + anchor_offset
    const int A = n_class / 2;
    if (A < 1 || n_class != 2 * A || n_box % A != 0) return MDT_ERR_INVALID_ARGUMENT;
    const int d2 = n_box / A;
    if (anchor_offset + voxels_per_element * A > anchors_total) return MDT_ERR_INVALID_ARGUMENT;
    const long long tiles = (V + 31) / 32;
    long long blocks = (tiles + 1) / 2;
    if (blocks > 768) blocks = 768;          // three resident blocks per CU (50 KB of LDS each): one generation, every wave amortises the filter's trip into LDS over ~20 tiles
    (void)hipGetLastError();
    hipLaunchKernelGGL((rpn_heads_mfma_kernel<64, 2>), dim3((unsigned)blocks), dim3(128), 0, static_cast<hipStream_t>(stream), logits, deltas, h, bias_shared, w, bias,
                       (unsigned)V, (unsigned)voxels_per_element, n_class, n_box, anchors_total * 2, anchor_offset * 2, anchors_total * d2, anchor_offset * d2);
    return c1_check();
}

}  // extern "C"
