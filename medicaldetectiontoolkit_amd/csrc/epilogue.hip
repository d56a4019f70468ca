// epilogue.hip -- fused convolution epilogues for the FPN / ResNet conv path (gfx950).
//
// The backbone convolutions themselves stay on MIOpen / CK (north star).  What torch wraps around every one of them
// is not: a broadcast bias add, the residual / top-down add and the ReLU are three separate full passes over the
// activation (reference graph: models/backbone.py:147-153,197-206, utils/model_utils.py:751-781), and in backward the
// ReLU mask and the bias-gradient reduction are two more.  At fp32 these passes are pure HBM traffic
// (profiles/r01_bench_train_step_steady_state_kernels.csv: ~10 % of the step).  Here:
//
//   forward   y = act(x + bias[c] (+ residual))            one pass, 16-byte accesses, in place on the conv output
//   backward  gx = gy * (y > 0)   and   gbias[c] = sum gx  one pass + a tiny deterministic second stage
//
// Both memory orders of torch are handled through `inner`: channel(i) = (i / inner) % C with inner = 1 for
// channels_last(_3d) storage and inner = prod(spatial) for contiguous NC(D)HW.  No atomics: the bias gradient is
// reduced in a fixed order (per-thread running sums -> per-block partials -> one thread per channel), so training
// stays run-to-run deterministic.  HBM-bound elementwise work, no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mdt_hip.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int EP_THREADS = 256;
constexpr int EP_MAX_BLOCKS = 1024;

inline int ep_check()
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

// ---- forward ---------------------------------------------------------------------------------------------------
// CL: channels-last storage (channel = i % C); else contiguous (channel = (i / inner) % C, inner % 4 == 0 on the vector path)
template <bool CL, bool RES, bool RELU>
// y and x (and res) may alias: the Python side runs the epilogue IN PLACE (y == x); each thread reads its elements before it
// writes the same ones, so no restrict qualifier on them
__global__ __launch_bounds__(EP_THREADS) void bias_act_fwd_kernel(float *y, const float *x,
                                                                  const float *__restrict__ bias, const float *res,
                                                                  long long n4, long long n, int C, long long inner)
{
    const long long stride = (long long)gridDim.x * EP_THREADS;
    for (long long i4 = (long long)blockIdx.x * EP_THREADS + threadIdx.x; i4 < n4; i4 += stride) {
        const long long i = i4 * 4;
        v4f v = reinterpret_cast<const v4f *>(x)[i4];
        float b[4];
        if (CL) {
            int c = (int)(i % C);
#pragma unroll
            for (int k = 0; k < 4; ++k) { b[k] = bias[c]; c = (c + 1 == C) ? 0 : c + 1; }
        } else {
            const float bb = bias[(int)((i / inner) % C)];      // inner % 4 == 0: the four share one channel
            b[0] = b[1] = b[2] = b[3] = bb;
        }
        if (RES) {
            const v4f r = reinterpret_cast<const v4f *>(res)[i4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] + b[k] + r[k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] + b[k];
        }
        if (RELU) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.0f ? v[k] : 0.0f;
        }
        reinterpret_cast<v4f *>(y)[i4] = v;
    }
    // scalar tail (n % 4 elements) and the fully scalar fallback are handled by the launcher with n4 = 0 .. see below
    if (blockIdx.x == 0) {
        for (long long i = n4 * 4 + threadIdx.x; i < n; i += EP_THREADS) {
            const int c = CL ? (int)(i % C) : (int)((i / inner) % C);
            float v = x[i] + bias[c];
            if (RES) v = v + res[i];
            if (RELU) v = v > 0.0f ? v : 0.0f;
            y[i] = v;
        }
    }
}

// partial rows of the in-launch second stage: write-through (sc1) stores / sc1 loads = relaxed agent-scope atomics on gfx950
__device__ __forceinline__ void store_partial(float *p, float v, bool through)
{
    if (through) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ float load_partial(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- backward, channels-last -----------------------------------------------------------------------------------
// Thread t of a block owns channel slots; a block walks a contiguous range of pixels.  C < 256: ppb = 256 / C pixels are
// processed side by side (thread t -> pixel lane t / C, channel t % C) and the lanes are folded in LDS in lane order;
// C >= 256: thread t owns channels t, t + 256, ... of every pixel.  Per-block partials go to `partial[C][blocks]` (one contiguous run per channel: the second
// stage's loads coalesce -- as `[block][C]` its 256 threads read 256 different rows, ~6 us per layer, ~55 layers per step); with a ticket: `partial[block][C]`.
// ticket != null (round 6): the second stage runs INSIDE this launch -- every block publishes its partial row with write-through (sc1) stores,
// draws a ticket, and the block that draws the last one folds all rows into gbias with sc1 loads (per-XCD L2s are not coherent: the sc1 / sc1
// form of cdna_hip_programming.md 6 G16 / MI355X_MICROARCH.md "inter-workgroup visibility") and resets the ticket to 0 for the next launch.
// One launch instead of two per convolution layer (~60 per training step); still a fixed summation order (deterministic).
template <bool RELU>
__global__ __launch_bounds__(EP_THREADS) void bias_act_bwd_cl_kernel(float *__restrict__ gx, const float *__restrict__ gy,
                                                                     const float *__restrict__ y, float *__restrict__ partial,
                                                                     long long npix, int C, int ppb, int ktiles, long long pix_per_block,
                                                                     int *__restrict__ ticket, float *__restrict__ gbias)
{
    __shared__ float s_acc[EP_THREADS + 1];        // [EP_THREADS]: "this block drew the last ticket" (one LDS object)
    const int t = threadIdx.x;
    const long long p0 = (long long)blockIdx.x * pix_per_block;
    long long p1 = p0 + pix_per_block;
    if (p1 > npix) p1 = npix;
    if (ppb >= 1 && ktiles == 1) {
        const int lanes = ppb * C;                 // active threads
        const int pl = t / C, c = t - pl * C;
        float acc = 0.0f;
        if (t < lanes) {
            long long p = p0 + pl;
            for (; p + 3LL * ppb < p1; p += 4LL * ppb) {          // four independent loads in flight
                float g[4], v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const long long i = (p + (long long)u * ppb) * C + c; g[u] = gy[i]; v[u] = RELU ? y[i] : 1.0f; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (RELU) g[u] = (v[u] > 0.0f) ? g[u] : 0.0f;
                    if (RELU || gx != nullptr) gx[(p + (long long)u * ppb) * C + c] = g[u];
                    acc = acc + g[u];
                }
            }
            for (; p < p1; p += ppb) {
                const long long i = p * C + c;
                float g = gy[i];
                if (RELU) g = (y[i] > 0.0f) ? g : 0.0f;
                if (RELU || gx != nullptr) gx[i] = g;
                acc = acc + g;
            }
        }
        s_acc[t] = acc;
        __syncthreads();
        if (t < C) {
            float s = 0.0f;
            for (int l = 0; l < ppb; ++l) s = s + s_acc[l * C + t];     // fixed lane order
            store_partial(ticket ? partial + (long long)blockIdx.x * C + t : partial + (long long)t * gridDim.x + blockIdx.x, s, ticket != nullptr);
        }
    } else {
        for (int k = 0; k < ktiles; ++k) {
            const int c = t + k * EP_THREADS;
            float acc = 0.0f;
            if (c < C) {
                long long p = p0;
                for (; p + 3 < p1; p += 4) {
                    float g[4], v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const long long i = (p + u) * C + c; g[u] = gy[i]; v[u] = RELU ? y[i] : 1.0f; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (RELU) g[u] = (v[u] > 0.0f) ? g[u] : 0.0f;
                        if (RELU || gx != nullptr) gx[(p + u) * C + c] = g[u];
                        acc = acc + g[u];
                    }
                }
                for (; p < p1; ++p) {
                    const long long i = p * C + c;
                    float g = gy[i];
                    if (RELU) g = (y[i] > 0.0f) ? g : 0.0f;
                    if (RELU || gx != nullptr) gx[i] = g;
                    acc = acc + g;
                }
                store_partial(ticket ? partial + (long long)blockIdx.x * C + c : partial + (long long)c * gridDim.x + blockIdx.x, acc, ticket != nullptr);
            }
        }
    }
    if (ticket == nullptr) return;
    // ---- publish this block's row, draw a ticket.  The row was written with write-through (sc1) stores: once they have drained
    // (vmcnt(0)) they are in memory, and the relaxed agent-scope ticket add cannot overtake them.  NO release fence: an agent-scope release
    // is an L2 write-back of everything the XCD has dirtied (buffer_wbl2) -- this kernel has just written the whole gx tensor, and 2048 blocks
    // each flushing it made the launch 5x slower (827 us instead of 170 us on the C2 maps, profiles/r06).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        const int drawn = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_acc[EP_THREADS] = (drawn == (int)gridDim.x - 1) ? 1.0f : 0.0f;
    }
    __syncthreads();
    if (s_acc[EP_THREADS] == 0.0f) return;
    // ---- last arriver: gbias[c] = sum over blocks of partial[block][c]; the rows are read with sc1 loads (served by memory / the fabric, never
    // by this CU's L1 or a stale line of this XCD's L2), whole rows at a time, lanes folded in lane order
    if (t == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch on this stream
    const long long nb = gridDim.x;
    if (ktiles == 1) {
        const int lanes = ppb * C;
        const int pl = t / C, c = t - pl * C;
        float acc = 0.0f;
        if (t < lanes) {
            // eight independent sc1 loads in flight per lane (each is a ~2 us trip to memory; one at a time the 2048 rows cost 700 us)
            long long b = pl;
            for (; b + 7LL * ppb < nb; b += 8LL * ppb) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = load_partial(partial + (b + (long long)u * ppb) * C + c);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = acc + v[u];
            }
            for (; b < nb; b += ppb) acc = acc + load_partial(partial + b * C + c);
        }
        __syncthreads();
        s_acc[t] = acc;
        __syncthreads();
        if (t < C) {
            float sum = 0.0f;
            for (int l = 0; l < ppb; ++l) sum = sum + s_acc[l * C + t];
            gbias[t] = sum;
        }
    } else {
        for (int c = t; c < C; c += EP_THREADS) {
            float sum = 0.0f;
            long long b = 0;
            for (; b + 7 < nb; b += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = load_partial(partial + (b + u) * C + c);
#pragma unroll
                for (int u = 0; u < 8; ++u) sum = sum + v[u];
            }
            for (; b < nb; ++b) sum = sum + load_partial(partial + b * C + c);
            gbias[c] = sum;
        }
    }
}

// ---- forward with the residual UP-SAMPLED on the fly (round 6) ---------------------------------------------------------------------
// y = x + bias[c] + coarse[b][y / sy][x / sx][z / sz][c] on channels-last storage: the FPN's top-down step (models/backbone.py:147-153:
// P_conv1(c) + F.interpolate(p_coarser, scale_factor=2)) without materialising the up-sampled map (a 151 MB write + read on P2 at the
// benchmark patch).  Thread = 4 consecutive channels of one voxel (C % 4 == 0); index math is 32-bit.
__global__ __launch_bounds__(EP_THREADS) void bias_act_fwd_up_kernel(float *y, const float *x, const float *__restrict__ bias, const float *__restrict__ coarse,
                                                                     unsigned n4, unsigned q, unsigned Y, unsigned X, unsigned Z, unsigned sy, unsigned sx, unsigned sz)
{
    const unsigned Yc = Y / sy, Xc = X / sx, Zc = Z / sz;
    const unsigned stride = gridDim.x * EP_THREADS;
    for (unsigned i4 = blockIdx.x * EP_THREADS + threadIdx.x; i4 < n4; i4 += stride) {
        const unsigned vox = i4 / q, cq = i4 - vox * q;
        const unsigned r = vox / Z, zz = vox - r * Z;
        const unsigned r2 = r / X, xx = r - r2 * X;
        const unsigned b = r2 / Y, yy = r2 - b * Y;
        const unsigned cv = ((b * Yc + yy / sy) * Xc + xx / sx) * Zc + zz / sz;
        v4f v = reinterpret_cast<const v4f *>(x)[i4];
        float bb[4];                                                  // (scalar loads: a bias inside a flat parameter buffer is only 4-byte aligned)
#pragma unroll
        for (int k = 0; k < 4; ++k) bb[k] = bias[cq * 4 + k];
        const v4f rr = reinterpret_cast<const v4f *>(coarse)[(unsigned long long)cv * q + cq];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] + bb[k] + rr[k];      // the order of mdt_bias_act_forward with a residual
        reinterpret_cast<v4f *>(y)[i4] = v;
    }
}

// ---- backward of a bias-only epilogue whose output gradient arrives ROW-MAJOR while the layer runs channels-last (round 6) ----------
// gx[b][v][c] = gy[b][c][v], gbias[c] = sum of gy over channel c, one pass: the P_conv2 layers of the FPN receive their output gradient
// from the RoIAlign backward / the RPN patch scatter (row-major maps, cuda_functions/_roi_align_impl.PyramidGradAccumulator) and hand it
// to a channels-last convolution backward -- a layout copy (302 MB of traffic on P2) followed by the bias reduction (151 MB) before.
// Block = 256 voxels of one batch element: thread = voxel reads its C values (a wave reads 256 contiguous bytes per channel) into an
// LDS tile [voxel][C | 1]; then thread = (voxel lane, channel) streams the tile out as one contiguous run of 256 * C floats, summing its
// channel on the way; lanes are folded in lane order, per-block partials go to bias_grad_finish_kernel (fixed order, deterministic).
constexpr int TR_TV = 256;
__global__ __launch_bounds__(EP_THREADS) void bias_grad_to_cl_kernel(float *__restrict__ gx, const float *__restrict__ gy, float *__restrict__ partial,
                                                                     long long V, int C, int ppb, int tiles_per_elem)
{
    extern __shared__ __attribute__((aligned(16))) float s_t[];      // [TR_TV][C | 1], then [EP_THREADS]
    const int CS = C | 1;
    float *s_acc = s_t + TR_TV * CS;
    const int t = threadIdx.x;
    const int b = blockIdx.x / tiles_per_elem, tile = blockIdx.x - b * tiles_per_elem;
    const long long v0 = (long long)tile * TR_TV;
    const int nv = (int)min((long long)TR_TV, V - v0);
    if (t < nv) {
        const float *src = gy + (long long)b * C * V + v0 + t;
#pragma unroll 6
        for (int c = 0; c < C; ++c) s_t[t * CS + c] = src[(long long)c * V];
    }
    __syncthreads();
    const int pl = t / C, c = t - pl * C;
    float acc = 0.0f;
    if (pl < ppb) {
        float *dst = gx + ((long long)b * V + v0) * C + c;
        for (int p = pl; p < nv; p += ppb) {
            const float g = s_t[p * CS + c];
            dst[(long long)p * C] = g;
            acc = acc + g;
        }
    }
    s_acc[t] = acc;
    __syncthreads();
    if (t < C) {
        float sum = 0.0f;
        for (int l = 0; l < ppb; ++l) sum = sum + s_acc[l * C + t];
        partial[(long long)t * gridDim.x + blockIdx.x] = sum;
    }
}

// ---- backward, contiguous NC(D)HW: one block per chunk of one (n, c) row ---------------------------------------
template <bool RELU>
__global__ __launch_bounds__(EP_THREADS) void bias_act_bwd_nc_kernel(float *__restrict__ gx, const float *__restrict__ gy,
                                                                     const float *__restrict__ y, float *__restrict__ partial,
                                                                     long long inner, int chunks, long long chunk_len)
{
    __shared__ float s_acc[EP_THREADS];
    const long long row = blockIdx.x / chunks;         // n * C + c
    const int chunk = blockIdx.x % chunks;
    const long long e0 = (long long)chunk * chunk_len;
    long long e1 = e0 + chunk_len;
    if (e1 > inner) e1 = inner;
    float acc = 0.0f;
    for (long long e = e0 + threadIdx.x; e < e1; e += EP_THREADS) {
        const long long i = row * inner + e;
        float g = gy[i];
        if (RELU) g = (y[i] > 0.0f) ? g : 0.0f;
        if (RELU || gx != nullptr) gx[i] = g;
        acc = acc + g;
    }
    s_acc[threadIdx.x] = acc;
    __syncthreads();
    for (int s = EP_THREADS / 2; s > 0; s >>= 1) {      // fixed tree
        if ((int)threadIdx.x < s) s_acc[threadIdx.x] = s_acc[threadIdx.x] + s_acc[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s_acc[0];
}

// second stage: gbias[c] = sum of the partials of channel c.  One block per channel: every thread adds a fixed strided
// subset in index order, then a fixed LDS tree -- deterministic, and no serial chain of dependent global loads
__global__ __launch_bounds__(EP_THREADS) void bias_grad_finish_kernel(float *__restrict__ gbias, const float *__restrict__ partial,
                                                                      int C, long long count, long long stride_c, long long stride_k,
                                                                      long long groups, long long stride_g)
{
    __shared__ float s_acc[EP_THREADS];
    const int c = blockIdx.x;
    const long long total = groups * count;
    float s = 0.0f;
    if (groups == 1) {
        // one group (channels-last layers: ~55 launches per training step): no index division, four independent loads in flight per thread; the
        // order of a thread's additions is the order of the general loop below (j ascending)
        const float *pc = partial + (long long)c * stride_c;
        long long j = threadIdx.x;
        for (; j + 3 * EP_THREADS < count; j += 4 * EP_THREADS) {
            const float v0 = pc[j * stride_k], v1 = pc[(j + EP_THREADS) * stride_k], v2 = pc[(j + 2 * EP_THREADS) * stride_k], v3 = pc[(j + 3 * EP_THREADS) * stride_k];
            s = s + v0; s = s + v1; s = s + v2; s = s + v3;
        }
        for (; j < count; j += EP_THREADS) s = s + pc[j * stride_k];
    } else {
        for (long long j = threadIdx.x; j < total; j += EP_THREADS) {
            const long long g = j / count, k = j - g * count;
            s = s + partial[g * stride_g + (long long)c * stride_c + k * stride_k];
        }
    }
    s_acc[threadIdx.x] = s;
    __syncthreads();
    for (int h = EP_THREADS / 2; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) s_acc[threadIdx.x] = s_acc[threadIdx.x] + s_acc[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) gbias[c] = s_acc[0];
}

// ---- input gradient of a 1x1(x1) convolution, ADDED to another gradient of the same tensor (round 4) -----------------------------
//   out[v][ci] = res[v][ci] + sum_co gy[v][co] * w[co][ci]          channels-last rows, co ascending (deterministic)
// A ResBlock input x feeds conv1 (1x1x1) AND the residual add (models/backbone.py:197-205): autograd computes conv1's input gradient
// (a 302 MB tensor on the C2 maps) and then adds it to the residual path's gradient in a separate pass (read 2 x 302 MB, write 302 MB:
// 150 us, four times per step).  Here the input gradient is produced already added to the other gradient: one pass that reads gy
// (75 MB), the residual gradient (302 MB) and writes the sum (302 MB).  HBM-bound (K = 18 / 36 MACs per output on the VALU), no MFMA.
// Thread = (voxel of the block's group, 4 consecutive input channels); w lives in LDS, a voxel's gy row is read by its threads as float2.
__global__ __launch_bounds__(EP_THREADS) void conv1x1_dgrad_add_kernel(float *__restrict__ out, const float *__restrict__ gy, const float *__restrict__ w,
                                                                       const float *__restrict__ res, long long V, int cout, int cin, int vb, long long groups_per_block)
{
    extern __shared__ __attribute__((aligned(16))) float s_w[];      // [cout][cin]
    for (int t = threadIdx.x; t < cout * cin; t += EP_THREADS) s_w[t] = w[t];
    __syncthreads();
    const int Q = cin >> 2;
    const int t = threadIdx.x;
    if (t >= vb * Q) return;
    const int vl = t / Q, q = t - vl * Q;
    const long long g0 = (long long)blockIdx.x * groups_per_block;
    for (long long gi = g0; gi < g0 + groups_per_block; ++gi) {
        const long long v = gi * vb + vl;
        if (v >= V) return;
        v4f acc = res ? *reinterpret_cast<const v4f *>(res + v * cin + 4 * q) : v4f{0.f, 0.f, 0.f, 0.f};
        const float2 *grow = reinterpret_cast<const float2 *>(gy + v * cout);
        for (int c2 = 0; c2 < (cout >> 1); ++c2) {
            const float2 gv = grow[c2];
            const v4f w0 = *reinterpret_cast<const v4f *>(s_w + (2 * c2) * cin + 4 * q);
            const v4f w1 = *reinterpret_cast<const v4f *>(s_w + (2 * c2 + 1) * cin + 4 * q);
            acc.x = acc.x + gv.x * w0.x; acc.y = acc.y + gv.x * w0.y; acc.z = acc.z + gv.x * w0.z; acc.w = acc.w + gv.x * w0.w;
            acc.x = acc.x + gv.y * w1.x; acc.y = acc.y + gv.y * w1.y; acc.z = acc.z + gv.y * w1.z; acc.w = acc.w + gv.y * w1.w;
        }
        *reinterpret_cast<v4f *>(out + v * cin + 4 * q) = acc;
    }
}

// The same product on the matrix cores (the VALU form above is issue-bound: 249 us for the 18 -> 72 layer on the C2 maps, 2.7 TB/s):
// a wave owns tiles of 32 voxels; A = the tile's gy rows (32 x c_out, one contiguous 32 * c_out * 4 byte run, staged through the wave's
// LDS slot with 16-byte loads), B = the weight, kept in registers for the wave's lifetime (KS K-steps x NT column tiles), C = the residual
// gradient, loaded straight into the accumulator layout of v_mfma_f32_32x32x2_f32 (column = lane & 31 = channel, 16 rows per lane: every
// wave-level load / store is two 128-byte row segments), D = the sum, stored the same way.  fp32 MFMA: exact products, fixed order.
template <int KS, int NT>
__global__ __launch_bounds__(256) void conv1x1_dgrad_add_mfma_kernel(float *__restrict__ out, const float *__restrict__ gy, const float *__restrict__ w,
                                                                     const float *__restrict__ res, long long V, int cin)
{
    typedef float v16f __attribute__((ext_vector_type(16)));
    constexpr int COUT = 2 * KS;
    constexpr int ASTR = COUT + 1;                         // odd row stride: the 32 rows of a fragment read hit 32 different banks
    __shared__ float s_a[4][32 * ASTR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    float bfrag[KS][NT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ci = nt * 32 + col;
            bfrag[ks][nt] = ci < cin ? w[(long long)(2 * ks + half) * cin + ci] : 0.0f;
        }
    const long long tiles = (V + 31) / 32;
    float *sa = s_a[wave];
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < tiles; tile += (long long)gridDim.x * 4) {
        const long long v0 = tile * 32;
        const int nv = (int)min((long long)32, V - v0);
        // A: the tile's gy rows, contiguous in memory -> LDS (row stride ASTR)
        for (int e = lane; e < nv * COUT; e += 64) {
            const int m = e / COUT, k = e - m * COUT;
            sa[m * ASTR + k] = gy[v0 * COUT + e];
        }
        for (int e = nv * COUT + lane; e < 32 * COUT; e += 64) { const int m = e / COUT, k = e - m * COUT; sa[m * ASTR + k] = 0.0f; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float afrag[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) afrag[ks] = sa[col * ASTR + 2 * ks + half];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ci = nt * 32 + col;
            v16f acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                acc[r] = (res != nullptr && ci < cin && row < nv) ? res[(v0 + row) * cin + ci] : 0.0f;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[ks], bfrag[ks][nt], acc, 0, 0, 0);
            if (ci < cin) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row < nv) out[(v0 + row) * cin + ci] = acc[r];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();       // the slot is rewritten by the next tile
    }
}

}  // namespace

extern "C" {

int mdt_bias_act_forward(float *y, const float *x, const float *bias, const float *residual,
                         long long n, int channels, long long inner, int relu, void *stream)
{
    if (n < 0 || channels <= 0 || inner <= 0 || (n % ((long long)channels * inner)) != 0) return MDT_ERR_INVALID_ARGUMENT;
    if (n == 0) return MDT_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool cl = inner == 1;
    const bool aligned = ((((uintptr_t)y) | ((uintptr_t)x) | ((uintptr_t)residual)) & 15) == 0;
    const long long n4 = (aligned && (cl || inner % 4 == 0)) ? n / 4 : 0;
    long long blocks = (n4 + EP_THREADS - 1) / EP_THREADS;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    (void)hipGetLastError();
#define LAUNCH_FWD(CL, RES, RELU) hipLaunchKernelGGL((bias_act_fwd_kernel<CL, RES, RELU>), dim3((unsigned)blocks), dim3(EP_THREADS), 0, s, \
                                                     y, x, bias, residual, n4, n, channels, inner)
    const bool res = residual != nullptr;
    if (cl) {
        if (res) { if (relu) LAUNCH_FWD(true, true, true); else LAUNCH_FWD(true, true, false); }
        else { if (relu) LAUNCH_FWD(true, false, true); else LAUNCH_FWD(true, false, false); }
    } else {
        if (res) { if (relu) LAUNCH_FWD(false, true, true); else LAUNCH_FWD(false, true, false); }
        else { if (relu) LAUNCH_FWD(false, false, true); else LAUNCH_FWD(false, false, false); }
    }
#undef LAUNCH_FWD
    return ep_check();
}

size_t mdt_bias_act_backward_workspace_bytes(long long n, int channels, long long inner)
{
    if (n <= 0 || channels <= 0 || inner <= 0) return 256;
    if (inner == 1) return (size_t)4 * EP_MAX_BLOCKS * channels * sizeof(float) + 256;
    const long long rows = n / inner;
    long long chunks = (EP_MAX_BLOCKS * 2 + rows - 1) / rows;
    if (chunks < 1) chunks = 1;
    return (size_t)(rows * chunks) * sizeof(float) + 256;
}

static int bias_act_backward_impl(float *gx, const float *gy, const float *y, float *gbias, long long n, int channels, long long inner, int relu,
                                  void *workspace, size_t workspace_bytes, int *ticket, void *stream);

int mdt_bias_act_backward(float *gx, const float *gy, const float *y, float *gbias,
                          long long n, int channels, long long inner, int relu,
                          void *workspace, size_t workspace_bytes, void *stream)
{
    return bias_act_backward_impl(gx, gy, y, gbias, n, channels, inner, relu, workspace, workspace_bytes, nullptr, stream);
}

int mdt_bias_act_backward_ticket(float *gx, const float *gy, const float *y, float *gbias,
                                 long long n, int channels, long long inner, int relu,
                                 void *workspace, size_t workspace_bytes, int *ticket, void *stream)
{
    if (ticket == nullptr) return MDT_ERR_INVALID_ARGUMENT;
    return bias_act_backward_impl(gx, gy, y, gbias, n, channels, inner, relu, workspace, workspace_bytes, ticket, stream);
}

static int bias_act_backward_impl(float *gx, const float *gy, const float *y, float *gbias, long long n, int channels, long long inner, int relu,
                                  void *workspace, size_t workspace_bytes, int *ticket, void *stream)
{
    if (n < 0 || channels <= 0 || inner <= 0 || (n % ((long long)channels * inner)) != 0) return MDT_ERR_INVALID_ARGUMENT;
    if (relu && (y == nullptr || gx == nullptr)) return MDT_ERR_INVALID_ARGUMENT;      // (relu == 0: gx may be null -- the input gradient IS gy, nothing is stored)
    hipStream_t s = (hipStream_t)stream;
    if (workspace == nullptr || workspace_bytes < mdt_bias_act_backward_workspace_bytes(n, channels, inner)) return MDT_ERR_WORKSPACE_TOO_SMALL;
    float *partial = reinterpret_cast<float *>(workspace);
    (void)hipGetLastError();
    const int fin_blocks = channels;
    if (n == 0) {
        hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(fin_blocks), dim3(EP_THREADS), 0, s, gbias, partial, channels, 0LL, 0LL, 0LL, 0LL, 0LL);
        return ep_check();
    }
    if (inner == 1) {
        const long long npix = n / channels;
        const int ktiles = (channels + EP_THREADS - 1) / EP_THREADS;
        const int ppb = (ktiles == 1) ? EP_THREADS / channels : 1;
        // enough blocks to fill the chip, each with a contiguous pixel range (a multiple of ppb)
        long long blocks = (npix + 8LL * ppb - 1) / (8LL * ppb);      // ~8 pixel iterations per thread
        // with the in-launch second stage the last block reads every partial row: one resident wave of blocks (256 CUs x 8), not more
        const long long cap = ticket ? 2 * EP_MAX_BLOCKS : 4 * EP_MAX_BLOCKS;
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        long long ppblock = (npix + blocks - 1) / blocks;
        ppblock = ((ppblock + ppb - 1) / ppb) * ppb;
        blocks = (npix + ppblock - 1) / ppblock;
        if (relu) hipLaunchKernelGGL(bias_act_bwd_cl_kernel<true>, dim3((unsigned)blocks), dim3(EP_THREADS), 0, s, gx, gy, y, partial, npix, channels, ppb, ktiles, ppblock, ticket, gbias);
        else hipLaunchKernelGGL(bias_act_bwd_cl_kernel<false>, dim3((unsigned)blocks), dim3(EP_THREADS), 0, s, gx, gy, y, partial, npix, channels, ppb, ktiles, ppblock, ticket, gbias);
        if (ep_check() != MDT_OK) return MDT_ERR_LAUNCH_FAILED;
        if (ticket) return MDT_OK;            // the last block of the launch has written gbias
        // partial[c][block]: channel stride blocks, block stride 1
        hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(fin_blocks), dim3(EP_THREADS), 0, s, gbias, partial, channels, blocks, blocks, 1LL, 1LL, 0LL);
        return ep_check();
    }
    const long long rows = n / inner;                       // batch * C
    long long chunks = (EP_MAX_BLOCKS * 2 + rows - 1) / rows;
    if (chunks < 1) chunks = 1;
    if (chunks > (inner + 1023) / 1024) chunks = (inner + 1023) / 1024;
    if (chunks < 1) chunks = 1;
    const long long chunk_len = (inner + chunks - 1) / chunks;
    const long long blocks = rows * chunks;
    if (blocks > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    if (relu) hipLaunchKernelGGL(bias_act_bwd_nc_kernel<true>, dim3((unsigned)blocks), dim3(EP_THREADS), 0, s, gx, gy, y, partial, inner, (int)chunks, chunk_len);
    else hipLaunchKernelGGL(bias_act_bwd_nc_kernel<false>, dim3((unsigned)blocks), dim3(EP_THREADS), 0, s, gx, gy, y, partial, inner, (int)chunks, chunk_len);
    if (ep_check() != MDT_OK) return MDT_ERR_LAUNCH_FAILED;
    // partial[(n * C + c) * chunks + k]: channel stride chunks, chunk stride 1, batch groups stride C * chunks
    hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(fin_blocks), dim3(EP_THREADS), 0, s, gbias, partial, channels, chunks, chunks, 1LL,
                       rows / channels, (long long)channels * chunks);
    return ep_check();
}

int mdt_bias_act_forward_upsampled_supported(int channels, long long n)
{
    return (channels >= 4 && (channels & 3) == 0 && n > 0 && n / 4 < 0x7fffffffLL) ? 1 : 0;
}

int mdt_bias_act_forward_upsampled(float *y, const float *x, const float *bias, const float *coarse, int batch, int Y, int X, int Z, int channels,
                                   int sy, int sx, int sz, void *stream)
{
    if (!y || !x || !bias || !coarse || batch < 0 || Y <= 0 || X <= 0 || Z <= 0 || sy <= 0 || sx <= 0 || sz <= 0) return MDT_ERR_INVALID_ARGUMENT;
    if ((Y % sy) || (X % sx) || (Z % sz)) return MDT_ERR_INVALID_ARGUMENT;
    const long long n = (long long)batch * Y * X * Z * channels;
    if (n == 0) return MDT_OK;
    if (!mdt_bias_act_forward_upsampled_supported(channels, n)) return MDT_ERR_UNSUPPORTED;
    if (((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)coarse)) & 15) != 0) return MDT_ERR_UNSUPPORTED;
    const long long n4 = n / 4;
    long long blocks = (n4 + EP_THREADS - 1) / EP_THREADS;
    if (blocks > 8192) blocks = 8192;
    (void)hipGetLastError();
    hipLaunchKernelGGL(bias_act_fwd_up_kernel, dim3((unsigned)blocks), dim3(EP_THREADS), 0, (hipStream_t)stream, y, x, bias, coarse, (unsigned)n4,
                       (unsigned)(channels / 4), (unsigned)Y, (unsigned)X, (unsigned)Z, (unsigned)sy, (unsigned)sx, (unsigned)sz);
    return ep_check();
}

int mdt_bias_grad_to_channels_last_supported(int channels) { return (channels >= 1 && channels <= 48) ? 1 : 0; }

size_t mdt_bias_grad_to_channels_last_workspace_bytes(int batch, int channels, long long inner)
{
    if (batch <= 0 || channels <= 0 || inner <= 0) return 256;
    return (size_t)batch * (size_t)((inner + TR_TV - 1) / TR_TV) * channels * sizeof(float) + 256;
}

int mdt_bias_grad_to_channels_last(float *gx, const float *gy, float *gbias, int batch, int channels, long long inner,
                                   void *workspace, size_t workspace_bytes, void *stream)
{
    if (!gx || !gy || !gbias || batch < 0 || channels <= 0 || inner < 0) return MDT_ERR_INVALID_ARGUMENT;
    if (!mdt_bias_grad_to_channels_last_supported(channels)) return MDT_ERR_UNSUPPORTED;
    if (workspace == nullptr || workspace_bytes < mdt_bias_grad_to_channels_last_workspace_bytes(batch, channels, inner)) return MDT_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t s = (hipStream_t)stream;
    float *partial = reinterpret_cast<float *>(workspace);
    (void)hipGetLastError();
    const long long tiles = (inner + TR_TV - 1) / TR_TV;
    const long long blocks = (long long)batch * tiles;
    if (blocks > 0x7fffffffLL || tiles > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    if (blocks > 0) {
        const size_t lds = ((size_t)TR_TV * (channels | 1) + EP_THREADS) * sizeof(float);
        hipLaunchKernelGGL(bias_grad_to_cl_kernel, dim3((unsigned)blocks), dim3(EP_THREADS), lds, s, gx, gy, partial, inner, channels, EP_THREADS / channels,
                           (int)tiles);
        if (ep_check() != MDT_OK) return MDT_ERR_LAUNCH_FAILED;
    }
    hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(channels), dim3(EP_THREADS), 0, s, gbias, partial, channels, blocks, blocks, 1LL, 1LL, 0LL);
    return ep_check();
}

int mdt_conv1x1_dgrad_add_supported(int c_out, int c_in)
{
    return (c_out >= 2 && (c_out & 1) == 0 && c_in >= 4 && (c_in & 3) == 0 && c_in / 4 <= EP_THREADS && (long long)c_out * c_in <= 12288) ? 1 : 0;
}

int mdt_conv1x1_dgrad_add(const float *gy, const float *w, const float *res, float *out, long long n_voxels, int c_out, int c_in, void *stream)
{
    if (!gy || !w || !out || n_voxels < 0) return MDT_ERR_INVALID_ARGUMENT;
    if (!mdt_conv1x1_dgrad_add_supported(c_out, c_in)) return MDT_ERR_UNSUPPORTED;
    if (n_voxels == 0) return MDT_OK;
    if (((((uintptr_t)gy) & 7) | (((uintptr_t)out) & 15) | (((uintptr_t)res) & 15)) != 0) return MDT_ERR_UNSUPPORTED;
    // the two layer shapes of the LIDC backbone whose maps are large (C2: 18 -> 72 wide, C3: 36 -> 144): matrix cores (the VALU form below
    // measured 249 / 127 us against 207 / 44 us: profiles/r04/r04_res_tap_probe.jsonl)
    if ((c_out == 18 && c_in == 72) || (c_out == 36 && c_in == 144)) {
        const long long tiles = (n_voxels + 31) / 32;
        long long blocks = (tiles + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        (void)hipGetLastError();
        if (c_out == 18) hipLaunchKernelGGL((conv1x1_dgrad_add_mfma_kernel<9, 3>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), out, gy, w, res, n_voxels, c_in);
        else hipLaunchKernelGGL((conv1x1_dgrad_add_mfma_kernel<18, 5>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), out, gy, w, res, n_voxels, c_in);
        return ep_check();
    }
    const int Q = c_in / 4;
    const int vb = EP_THREADS / Q;                                   // voxels per block and pass
    const long long groups = (n_voxels + vb - 1) / vb;
    long long blocks = groups < 8192 ? groups : 8192;
    const long long gpb = (groups + blocks - 1) / blocks;
    blocks = (groups + gpb - 1) / gpb;
    (void)hipGetLastError();
    hipLaunchKernelGGL(conv1x1_dgrad_add_kernel, dim3((unsigned)blocks), dim3(EP_THREADS), (size_t)c_out * c_in * sizeof(float), static_cast<hipStream_t>(stream),
                       out, gy, w, res, n_voxels, c_out, c_in, vb, gpb);
    return ep_check();
}

}  // extern "C"
