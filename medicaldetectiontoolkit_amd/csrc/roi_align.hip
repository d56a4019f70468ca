// roi_align.hip -- 2D/3D RoIAlign ("crop and resize") forward / backward for gfx950.
//
// Behaviour follows the reference CUDA kernels (paths relative to the reference):
//   3D: cuda_functions/roi_align_3D/roi_align/src/cuda/crop_and_resize_kernel.cu:12-151 (fwd), :154-304 (bwd)
//   2D: cuda_functions/roi_align_2D/roi_align/src/cuda/crop_and_resize_kernel.cu:11-99 (fwd), :102-194 (bwd)
// but the design is not theirs:
//   * forward: one workgroup per (RoI, slab of outputs); the per-axis sample
//     tables (floor index + lerp) are computed once per workgroup in LDS, so no
//     thread re-reads the box or redoes the coordinate arithmetic; outputs are
//     written coalesced; rows with an out-of-range box_ind are written as zeros
//     (fuses the zero-fill of crop_and_resize_gpu.c:26-27).
//   * backward: gather form, write-once, no atomics.  A workgroup owns a tile of
//     the gradient feature map, finds the RoIs whose footprint reaches the tile,
//     stages their sample tables and gradient slabs in LDS, accumulates in
//     registers and streams the tile out with 16-byte stores.  Tiles no RoI
//     reaches are a pure zero stream.  This fuses both zero-fills of the
//     reference (crop_and_resize.py:40, crop_and_resize_gpu.c:61) and makes the
//     result deterministic: per voxel the terms are added in exactly the order
//     a sequential out_idx loop would add them (RoI, y, x, z sample order; corner
//     order of kernel.cu:256-301), so it equals the CPU oracle bit for bit.
// This translation unit is compiled with -ffp-contract=off: sample coordinates
// and lerps must round like the uncontracted oracle.
//
// HBM-bound gather/scatter work: no MFMA.  Algorithmic bytes:
//   bwd: 4*B*C*V (grad_image written once) + 4*N*C*P (grads read once) + 28*N.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mdt_hip.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

struct AxisEntry {
    int lo;      // floorf(in)
    float lerp;  // in - lo;  ceilf(in) == lo + (lerp > 0)
};

// crop_and_resize_kernel.cu:51-75 -- see oracle/mdt_oracle.c sample_coord for the
// type analysis (the 0.5 literals are double).
__device__ __forceinline__ float sample_coord(float a1, float a2, int L, int P, int p)
{
    float in;
    if (P > 1) {
        const float scale = (a2 - a1) * (float)L / (float)P;
        const float t = a1 * (float)L + (float)p * scale + scale / 2.0f;
        in = (float)((double)t - 0.5);
    } else {
        in = (float)(0.5 * (double)(a1 + a2) * (double)L);
    }
    if (in > (float)(L - 1)) in = (float)(L - 1);
    if (in < 0.0f) in = 0.0f;
    return in;
}

__device__ __forceinline__ AxisEntry axis_entry(float a1, float a2, int L, int P, int p)
{
    const float in = sample_coord(a1, a2, L, P, p);
    AxisEntry e;
    e.lo = (int)floorf(in);
    e.lerp = in - (float)e.lo;
    return e;
}

__device__ __forceinline__ int entry_hi(const AxisEntry &e) { return e.lo + (e.lerp > 0.0f ? 1 : 0); }

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
constexpr int FWD_THREADS = 256;
constexpr int FWD_PER_THREAD = 4;
constexpr int FWD_SLAB = FWD_THREADS * FWD_PER_THREAD;

// DIM == 3: image [B,C,H,W,D], boxes [N,6], crops [N,C,ch,cw,cd]
// DIM == 2: image [B,C,H,W],   boxes [N,4], crops [N,C,ch,cw]      (D = cd = 1)
template <int DIM>
__global__ __launch_bounds__(FWD_THREADS) void crop_fwd_kernel(
    const float *__restrict__ image, const float *__restrict__ boxes,
    const int *__restrict__ box_ind, int B, int H, int W, int D,
    int ch, int cw, int cd, int C, float *__restrict__ crops)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    AxisEntry *tab = reinterpret_cast<AxisEntry *>(smem_raw);  // [ch + cw + cd]

    const int n = blockIdx.x;
    const int P = ch * cw * cd;
    const long long per_roi = (long long)C * P;
    const long long base = (long long)blockIdx.y * FWD_SLAB;
    const int b_in = box_ind[n];
    float *out = crops + (long long)n * per_roi;

    if (b_in < 0 || b_in >= B) {  // skipped RoI: reference leaves the zero-fill
#pragma unroll
        for (int k = 0; k < FWD_PER_THREAD; ++k) {
            const long long e = base + k * FWD_THREADS + threadIdx.x;
            if (e < per_roi) out[e] = 0.0f;
        }
        return;
    }

    const float *bx = boxes + (long long)n * (2 * DIM);
    for (int t = threadIdx.x; t < ch + cw + cd; t += FWD_THREADS) {
        AxisEntry e;
        if (t < ch) {
            e = axis_entry(bx[0], bx[2], H, ch, t);
        } else if (t < ch + cw) {
            e = axis_entry(bx[1], bx[3], W, cw, t - ch);
        } else {
            if (DIM == 3) e = axis_entry(bx[4], bx[5], D, cd, t - ch - cw);
            else { e.lo = 0; e.lerp = 0.0f; }
        }
        tab[t] = e;
    }
    __syncthreads();

    const long long vol = (long long)H * W * D;
#pragma unroll
    for (int k = 0; k < FWD_PER_THREAD; ++k) {
        const long long e = base + k * FWD_THREADS + threadIdx.x;
        if (e >= per_roi) continue;
        int idx = (int)e;
        int z = 0;
        if (DIM == 3) { z = idx % cd; idx /= cd; }
        const int x = idx % cw; idx /= cw;
        const int y = idx % ch;
        const int c = idx / ch;

        const AxisEntry ey = tab[y];
        const AxisEntry ex = tab[ch + x];
        const int top = ey.lo, bottom = entry_hi(ey);
        const int left = ex.lo, right = entry_hi(ex);
        const float *pimage = image + ((long long)b_in * C + c) * vol;

        if (DIM == 3) {
            const AxisEntry ez = tab[ch + cw + z];
            const int front = ez.lo, back = entry_hi(ez);
            const long long rt_l = (long long)D * (left + (long long)W * top);
            const long long rt_r = (long long)D * (right + (long long)W * top);
            const long long rb_l = (long long)D * (left + (long long)W * bottom);
            const long long rb_r = (long long)D * (right + (long long)W * bottom);
            const float tlf = pimage[front + rt_l], trf = pimage[front + rt_r];
            const float blf = pimage[front + rb_l], brf = pimage[front + rb_r];
            const float tlb = pimage[back + rt_l], trb = pimage[back + rt_r];
            const float blb = pimage[back + rb_l], brb = pimage[back + rb_r];
            const float top_front = tlf + (trf - tlf) * ex.lerp;
            const float bottom_front = blf + (brf - blf) * ex.lerp;
            const float top_back = tlb + (trb - tlb) * ex.lerp;
            const float bottom_back = blb + (brb - blb) * ex.lerp;
            const float frontv = top_front + (bottom_front - top_front) * ey.lerp;
            const float backv = top_back + (bottom_back - top_back) * ey.lerp;
            out[e] = frontv + (backv - frontv) * ez.lerp;
        } else {
            const float tl = pimage[(long long)top * W + left];
            const float tr = pimage[(long long)top * W + right];
            const float bl = pimage[(long long)bottom * W + left];
            const float br = pimage[(long long)bottom * W + right];
            const float topv = tl + (tr - tl) * ex.lerp;
            const float bottomv = bl + (br - bl) * ex.lerp;
            out[e] = topv + (bottomv - topv) * ey.lerp;
        }
    }
}

// ---------------------------------------------------------------------------
// backward, gather form
// ---------------------------------------------------------------------------
// Internal axes (slow -> fast): 3D (y, x, z);  2D (y, x) with z extent 1.
// The contiguous axis is vectorised: VEC = 4 when its extent is a multiple of 4
// (16-byte stores), else 1.
constexpr int BWD_THREADS = 256;
constexpr int BWD_K = 4;                         // vector units per thread per tile
constexpr int BWD_TILE_UNITS = BWD_THREADS * BWD_K;
constexpr int BWD_TB = 4;                        // RoIs staged per pass
constexpr int BWD_SLAB_FLOATS = 2048;            // LDS floats per staged RoI gradient slab

struct BwdParams {
    const float *grads;
    const float *boxes;
    const int *box_ind;
    float *out;
    int N, B, C;
    int H, W, D;        // D == 1 for 2D
    int ph, pw, pd;     // pd == 1 for 2D
    int units_per_vol;  // H*W*D / VEC
    int tiles_per_vol;
    long long tiles_total;
};

// LDS layout (dynamic): tab [BWD_TB][psum] AxisEntry | slab [BWD_TB][BWD_SLAB_FLOATS] float
//                       | list [BWD_THREADS] int | small control words
template <int DIM, int VEC>
__global__ __launch_bounds__(BWD_THREADS) void crop_bwd_gather_kernel(BwdParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int psum = p.ph + p.pw + p.pd;
    AxisEntry *tab = reinterpret_cast<AxisEntry *>(smem_raw);
    float *slab = reinterpret_cast<float *>(tab + BWD_TB * psum);
    int *list = reinterpret_cast<int *>(slab + BWD_TB * BWD_SLAB_FLOATS);
    int *wave_cnt = list + BWD_THREADS;         // [4]
    int *roi_meta = wave_cnt + 4;               // [BWD_TB][8]: r, py0, npy, staged, xlo, xhi, zlo, zhi

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int Dv = p.D / VEC;                   // vector units along the contiguous axis (3D: z, 2D: x)
    const int P = p.ph * p.pw * p.pd;
    // 2D: contiguous axis is x (extent W); treat as (y, x) with row length Wv
    const int row_units = (DIM == 3) ? p.W * Dv : (p.W / VEC);  // units per y row

    // contiguous chunk of tiles per workgroup (keeps b, c stable across iterations)
    const long long per_wg = (p.tiles_total + gridDim.x - 1) / gridDim.x;
    long long t0 = (long long)blockIdx.x * per_wg;
    long long t1 = t0 + per_wg;
    if (t1 > p.tiles_total) t1 = p.tiles_total;

    for (long long tile = t0; tile < t1; ++tile) {
        const int vol = (int)(tile / p.tiles_per_vol);
        const int chunk = (int)(tile % p.tiles_per_vol);
        const int b = vol / p.C;
        const int c = vol % p.C;
        const int u_base = chunk * BWD_TILE_UNITS;
        int u_end = u_base + BWD_TILE_UNITS;
        if (u_end > p.units_per_vol) u_end = p.units_per_vol;
        const int y_lo = u_base / row_units;
        const int y_hi = (u_end - 1) / row_units;

        float acc[BWD_K][VEC];
        int uy[BWD_K], ux[BWD_K], uz[BWD_K];   // uz/ux: first index on the contiguous axis
#pragma unroll
        for (int k = 0; k < BWD_K; ++k) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[k][v] = 0.0f;
            const int u = u_base + k * BWD_THREADS + tid;
            if (DIM == 3) {
                uz[k] = (u % Dv) * VEC;
                const int r = u / Dv;
                ux[k] = r % p.W;
                uy[k] = r / p.W;
            } else {
                uz[k] = 0;
                ux[k] = (u % row_units) * VEC;
                uy[k] = u / row_units;
            }
        }

        // ---- scan RoIs in chunks of 256, ordered compaction of those reaching this tile ----
        for (int rb = 0; rb < p.N; rb += BWD_THREADS) {
            const int r = rb + tid;
            bool hit = false;
            if (r < p.N && p.box_ind[r] == b) {
                const float *bx = p.boxes + (long long)r * (2 * DIM);
                const AxisEntry e0 = axis_entry(bx[0], bx[2], p.H, p.ph, 0);
                const AxisEntry e1 = axis_entry(bx[0], bx[2], p.H, p.ph, p.ph - 1);
                const int lo = min(e0.lo, e1.lo);
                const int hi = max(entry_hi(e0), entry_hi(e1));
                hit = (lo <= y_hi) && (hi >= y_lo);
            }
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) wave_cnt[wave] = __popcll(bal);
            __syncthreads();
            int off = 0, total = 0;
#pragma unroll
            for (int w = 0; w < BWD_THREADS / 64; ++w) {
                const int cnt = wave_cnt[w];
                if (w < wave) off += cnt;
                total += cnt;
            }
            if (hit) list[off + __popcll(bal & ((1ULL << lane) - 1ULL))] = r;
            __syncthreads();

            // ---- process the list in passes of BWD_TB RoIs ----
            for (int lb = 0; lb < total; lb += BWD_TB) {
                const int nb = min(BWD_TB, total - lb);
                // sample tables
                for (int t = tid; t < nb * psum; t += BWD_THREADS) {
                    const int j = t / psum;
                    const int q = t % psum;
                    const int rr = list[lb + j];
                    const float *bx = p.boxes + (long long)rr * (2 * DIM);
                    AxisEntry e;
                    if (q < p.ph) e = axis_entry(bx[0], bx[2], p.H, p.ph, q);
                    else if (q < p.ph + p.pw) e = axis_entry(bx[1], bx[3], p.W, p.pw, q - p.ph);
                    else {
                        if (DIM == 3) e = axis_entry(bx[4], bx[5], p.D, p.pd, q - p.ph - p.pw);
                        else { e.lo = 0; e.lerp = 0.0f; }
                    }
                    tab[j * psum + q] = e;
                }
                __syncthreads();
                // per-RoI metadata: range of sample rows py that reach [y_lo, y_hi], footprints
                if (tid < nb) {
                    const AxisEntry *ty = tab + tid * psum;
                    const AxisEntry *tx = ty + p.ph;
                    const AxisEntry *tz = tx + p.pw;
                    int py0 = p.ph, py1 = -1;
                    for (int q = 0; q < p.ph; ++q) {
                        const int lo = ty[q].lo, hi = entry_hi(ty[q]);
                        if (lo <= y_hi && hi >= y_lo) { py0 = min(py0, q); py1 = max(py1, q); }
                    }
                    const int npy = (py1 >= py0) ? (py1 - py0 + 1) : 0;
                    int *m = roi_meta + tid * 8;
                    m[0] = list[lb + tid];
                    m[1] = py0;
                    m[2] = npy;
                    m[3] = (npy * p.pw * p.pd <= BWD_SLAB_FLOATS) ? 1 : 0;
                    m[4] = min(tx[0].lo, tx[p.pw - 1].lo);
                    m[5] = max(entry_hi(tx[0]), entry_hi(tx[p.pw - 1]));
                    m[6] = (DIM == 3) ? min(tz[0].lo, tz[p.pd - 1].lo) : 0;
                    m[7] = (DIM == 3) ? max(entry_hi(tz[0]), entry_hi(tz[p.pd - 1])) : 0;
                }
                __syncthreads();
                // stage gradient slabs (rows py0..py0+npy-1 of grads[r, c]) -- contiguous in memory
                for (int j = 0; j < nb; ++j) {
                    const int *m = roi_meta + j * 8;
                    if (!m[3]) continue;
                    const int cnt = m[2] * p.pw * p.pd;
                    const float *src = p.grads + ((long long)m[0] * p.C + c) * P + (long long)m[1] * p.pw * p.pd;
                    float *dst = slab + j * BWD_SLAB_FLOATS;
                    for (int t = tid; t < cnt; t += BWD_THREADS) dst[t] = src[t];
                }
                __syncthreads();

                // ---- gather ----
                for (int j = 0; j < nb; ++j) {
                    const int *m = roi_meta + j * 8;
                    const int npy = m[2];
                    if (npy == 0) continue;
                    const int py0 = m[1];
                    const bool staged = m[3] != 0;
                    const int fxlo = m[4], fxhi = m[5], fzlo = m[6], fzhi = m[7];
                    const AxisEntry *ty = tab + j * psum;
                    const AxisEntry *tx = ty + p.ph;
                    const AxisEntry *tz = tx + p.pw;
                    const float *gsl = staged ? (slab + j * BWD_SLAB_FLOATS)
                                              : (p.grads + ((long long)m[0] * p.C + c) * P + (long long)py0 * p.pw * p.pd);
#pragma unroll
                    for (int k = 0; k < BWD_K; ++k) {
                        const int u = u_base + k * BWD_THREADS + tid;
                        if (u >= u_end) continue;
                        const int vy = uy[k];
                        if (DIM == 3) {
                            const int vx = ux[k], vz = uz[k];
                            if (vx < fxlo || vx > fxhi || vz + VEC - 1 < fzlo || vz > fzhi) continue;
                            for (int q = 0; q < npy; ++q) {
                                const AxisEntry ey = ty[py0 + q];
                                const bool mt = (ey.lo == vy);
                                const bool mb = (entry_hi(ey) == vy);
                                if (!(mt || mb)) continue;
                                for (int px = 0; px < p.pw; ++px) {
                                    const AxisEntry ex = tx[px];
                                    const bool ml = (ex.lo == vx);
                                    const bool mr = (entry_hi(ex) == vx);
                                    if (!(ml || mr)) continue;
                                    const float *grow = gsl + ((long long)q * p.pw + px) * p.pd;
                                    for (int pz = 0; pz < p.pd; ++pz) {
                                        const AxisEntry ez = tz[pz];
                                        const int zf = ez.lo, zb = entry_hi(ez);
                                        if (zb < vz || zf > vz + VEC - 1) continue;
                                        const float g = grow[pz];
                                        // corner order of kernel.cu:256-301: y outer, x middle, z inner
#pragma unroll
                                        for (int cy = 0; cy < 2; ++cy) {
                                            if (!(cy == 0 ? mt : mb)) continue;
                                            const float wy = cy == 0 ? (1.0f - ey.lerp) : ey.lerp;
#pragma unroll
                                            for (int cx = 0; cx < 2; ++cx) {
                                                if (!(cx == 0 ? ml : mr)) continue;
                                                const float wx = cx == 0 ? (1.0f - ex.lerp) : ex.lerp;
#pragma unroll
                                                for (int cz = 0; cz < 2; ++cz) {
                                                    const int zi = cz == 0 ? zf : zb;
                                                    const float wz = cz == 0 ? (1.0f - ez.lerp) : ez.lerp;
                                                    const float val = wx * wz * wy * g;
#pragma unroll
                                                    for (int v = 0; v < VEC; ++v)
                                                        if (zi == vz + v) acc[k][v] = acc[k][v] + val;
                                                }
                                            }
                                        }
                                    }
                                }
                            }
                        } else {
                            const int vx = ux[k];
                            if (vx + VEC - 1 < fxlo || vx > fxhi) continue;
                            for (int q = 0; q < npy; ++q) {
                                const AxisEntry ey = ty[py0 + q];
                                const bool mt = (ey.lo == vy);
                                const bool mb = (entry_hi(ey) == vy);
                                if (!(mt || mb)) continue;
                                const float *grow = gsl + (long long)q * p.pw;
                                for (int px = 0; px < p.pw; ++px) {
                                    const AxisEntry ex = tx[px];
                                    const int xl = ex.lo, xr = entry_hi(ex);
                                    if (xr < vx || xl > vx + VEC - 1) continue;
                                    const float g = grow[px];
                                    // 2D kernel.cu:175-192: dtop = (1-y_lerp)*g, then (1-x_lerp)*dtop, x_lerp*dtop; then bottom
#pragma unroll
                                    for (int cy = 0; cy < 2; ++cy) {
                                        if (!(cy == 0 ? mt : mb)) continue;
                                        const float dy = (cy == 0 ? (1.0f - ey.lerp) : ey.lerp) * g;
#pragma unroll
                                        for (int cx = 0; cx < 2; ++cx) {
                                            const int xi = cx == 0 ? xl : xr;
                                            const float val = (cx == 0 ? (1.0f - ex.lerp) : ex.lerp) * dy;
#pragma unroll
                                            for (int v = 0; v < VEC; ++v)
                                                if (xi == vx + v) acc[k][v] = acc[k][v] + val;
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                __syncthreads();  // tab / slab / roi_meta reused by the next pass
            }
        }

        // ---- stream the tile out ----
        float *ovol = p.out + (long long)vol * p.units_per_vol * VEC;
#pragma unroll
        for (int k = 0; k < BWD_K; ++k) {
            const int u = u_base + k * BWD_THREADS + tid;
            if (u >= u_end) continue;
            if (VEC == 4) {
                v4f o = {acc[k][0], acc[k][1], acc[k][2], acc[k][3]};
                __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(ovol) + u);
            } else {
                __builtin_nontemporal_store(acc[k][0], ovol + u);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// backward, atomic A/B variant: zero-fill + global fp32 atomics (reference algorithm)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zero_fill_kernel(v4f *__restrict__ out4, long long n4,
                                                        float *__restrict__ tail, int ntail)
{
    const v4f z = {0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(z, out4 + i);
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.0f;
}

__global__ __launch_bounds__(256) void crop_bwd3d_atomic_kernel(
    const float *__restrict__ grads, const float *__restrict__ boxes,
    const int *__restrict__ box_ind, long long total, int B, int H, int W, int D,
    int ch, int cw, int cd, int C, float *__restrict__ out)
{
    for (long long out_idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; out_idx < total;
         out_idx += (long long)gridDim.x * blockDim.x) {
        long long idx = out_idx;
        const int z = (int)(idx % cd); idx /= cd;
        const int x = (int)(idx % cw); idx /= cw;
        const int y = (int)(idx % ch); idx /= ch;
        const int c = (int)(idx % C);
        const int n = (int)(idx / C);
        const int b_in = box_ind[n];
        if (b_in < 0 || b_in >= B) continue;
        const float *bx = boxes + (long long)n * 6;
        const AxisEntry ey = axis_entry(bx[0], bx[2], H, ch, y);
        const AxisEntry ex = axis_entry(bx[1], bx[3], W, cw, x);
        const AxisEntry ez = axis_entry(bx[4], bx[5], D, cd, z);
        const int top = ey.lo, bottom = entry_hi(ey);
        const int left = ex.lo, right = entry_hi(ex);
        const int front = ez.lo, back = entry_hi(ez);
        float *pimage = out + ((long long)b_in * C + c) * H * W * D;
        const float g = grads[out_idx];
        const float xl = ex.lerp, yl = ey.lerp, zl = ez.lerp;
        atomicAdd(pimage + front + (long long)D * (left + (long long)W * top), (1 - xl) * (1 - zl) * (1 - yl) * g);
        atomicAdd(pimage + back + (long long)D * (left + (long long)W * top), (1 - xl) * zl * (1 - yl) * g);
        atomicAdd(pimage + front + (long long)D * (right + (long long)W * top), xl * (1 - zl) * (1 - yl) * g);
        atomicAdd(pimage + back + (long long)D * (right + (long long)W * top), xl * zl * (1 - yl) * g);
        atomicAdd(pimage + front + (long long)D * (left + (long long)W * bottom), (1 - xl) * (1 - zl) * yl * g);
        atomicAdd(pimage + back + (long long)D * (left + (long long)W * bottom), (1 - xl) * zl * yl * g);
        atomicAdd(pimage + front + (long long)D * (right + (long long)W * bottom), xl * (1 - zl) * yl * g);
        atomicAdd(pimage + back + (long long)D * (right + (long long)W * bottom), xl * zl * yl * g);
    }
}

inline int check_launch()
{
    return hipGetLastError() == hipSuccess ? MDT_OK : MDT_ERR_LAUNCH_FAILED;
}

template <int DIM>
int launch_fwd(const float *image, const float *boxes, const int *box_ind, int N, int B,
               int H, int W, int D, int ch, int cw, int cd, int C, float *crops, hipStream_t s)
{
    if (N < 0 || B <= 0 || H <= 0 || W <= 0 || D <= 0 || ch <= 0 || cw <= 0 || cd <= 0 || C <= 0)
        return MDT_ERR_INVALID_ARGUMENT;
    const long long per_roi = (long long)C * ch * cw * cd;
    if (N == 0 || per_roi == 0) return MDT_OK;
    if (per_roi > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    const size_t lds = (size_t)(ch + cw + cd) * sizeof(AxisEntry);
    if (lds > 60 * 1024) return MDT_ERR_UNSUPPORTED;
    const long long slabs = (per_roi + FWD_SLAB - 1) / FWD_SLAB;
    if (slabs > 65535) return MDT_ERR_UNSUPPORTED;
    dim3 grid((unsigned)N, (unsigned)slabs);
    hipLaunchKernelGGL(crop_fwd_kernel<DIM>, grid, dim3(FWD_THREADS), lds, s,
                       image, boxes, box_ind, B, H, W, D, ch, cw, cd, C, crops);
    return check_launch();
}

template <int DIM>
int launch_bwd(const float *grads, const float *boxes, const int *box_ind, int N, int B,
               int H, int W, int D, int ph, int pw, int pd, int C, float *out, hipStream_t s)
{
    if (N < 0 || B <= 0 || H <= 0 || W <= 0 || D <= 0 || ph <= 0 || pw <= 0 || pd <= 0 || C <= 0)
        return MDT_ERR_INVALID_ARGUMENT;
    const long long vol = (long long)H * W * D;
    if (vol > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    const int contig = (DIM == 3) ? D : W;
    const int vec = (contig % 4 == 0 && (((uintptr_t)out) & 15) == 0) ? 4 : 1;
    BwdParams p;
    p.grads = grads; p.boxes = boxes; p.box_ind = box_ind; p.out = out;
    p.N = N; p.B = B; p.C = C; p.H = H; p.W = W; p.D = D; p.ph = ph; p.pw = pw; p.pd = pd;
    p.units_per_vol = (int)(vol / vec);
    p.tiles_per_vol = (p.units_per_vol + BWD_TILE_UNITS - 1) / BWD_TILE_UNITS;
    p.tiles_total = (long long)B * C * p.tiles_per_vol;
    const int psum = ph + pw + pd;
    const size_t lds = (size_t)BWD_TB * psum * sizeof(AxisEntry) + (size_t)BWD_TB * BWD_SLAB_FLOATS * sizeof(float) +
                       (size_t)(BWD_THREADS + 4 + BWD_TB * 8) * sizeof(int);
    if (lds > 64 * 1024) return MDT_ERR_UNSUPPORTED;
    long long grid = p.tiles_total < 2048 ? p.tiles_total : 2048;
    if (grid <= 0) return MDT_OK;
    if (vec == 4)
        hipLaunchKernelGGL((crop_bwd_gather_kernel<DIM, 4>), dim3((unsigned)grid), dim3(BWD_THREADS), lds, s, p);
    else
        hipLaunchKernelGGL((crop_bwd_gather_kernel<DIM, 1>), dim3((unsigned)grid), dim3(BWD_THREADS), lds, s, p);
    return check_launch();
}

}  // namespace

extern "C" {

int mdt_crop_and_resize_3d_forward(const float *image, const float *boxes, const int *box_ind,
                                   int num_boxes, int batch, int H, int W, int D,
                                   int ch, int cw, int cd, int depth,
                                   float extrapolation_value, float *crops, void *stream)
{
    (void)extrapolation_value;
    return launch_fwd<3>(image, boxes, box_ind, num_boxes, batch, H, W, D, ch, cw, cd, depth, crops,
                         (hipStream_t)stream);
}

int mdt_crop_and_resize_2d_forward(const float *image, const float *boxes, const int *box_ind,
                                   int num_boxes, int batch, int H, int W,
                                   int ch, int cw, int depth,
                                   float extrapolation_value, float *crops, void *stream)
{
    (void)extrapolation_value;
    return launch_fwd<2>(image, boxes, box_ind, num_boxes, batch, H, W, 1, ch, cw, 1, depth, crops,
                         (hipStream_t)stream);
}

int mdt_crop_and_resize_3d_backward(const float *grads, const float *boxes, const int *box_ind,
                                    int num_boxes, int batch, int H, int W, int D,
                                    int ch, int cw, int cd, int depth,
                                    float *grads_image, void *stream)
{
    return launch_bwd<3>(grads, boxes, box_ind, num_boxes, batch, H, W, D, ch, cw, cd, depth, grads_image,
                         (hipStream_t)stream);
}

int mdt_crop_and_resize_2d_backward(const float *grads, const float *boxes, const int *box_ind,
                                    int num_boxes, int batch, int H, int W,
                                    int ch, int cw, int depth,
                                    float *grads_image, void *stream)
{
    return launch_bwd<2>(grads, boxes, box_ind, num_boxes, batch, H, W, 1, ch, cw, 1, depth, grads_image,
                         (hipStream_t)stream);
}

int mdt_crop_and_resize_3d_backward_atomic(const float *grads, const float *boxes, const int *box_ind,
                                           int num_boxes, int batch, int H, int W, int D,
                                           int ch, int cw, int cd, int depth,
                                           float *grads_image, void *stream)
{
    if (num_boxes < 0 || batch <= 0 || H <= 0 || W <= 0 || D <= 0 || ch <= 0 || cw <= 0 || cd <= 0 || depth <= 0)
        return MDT_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)batch * depth * H * W * D;
    const long long n4 = ((((uintptr_t)grads_image) & 15) == 0) ? n / 4 : 0;
    const int ntail = (int)(n - n4 * 4);
    if (ntail > 256) {  // unaligned output: scalar fill through the tail path is not worth optimising
        if (hipMemsetAsync(grads_image, 0, (size_t)n * sizeof(float), s) != hipSuccess) return MDT_ERR_LAUNCH_FAILED;
    } else {
        long long blocks = (n4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                           reinterpret_cast<v4f *>(grads_image), n4, grads_image + n4 * 4, ntail);
    }
    const long long total = (long long)num_boxes * depth * ch * cw * cd;
    if (total > 0) {
        long long blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(crop_bwd3d_atomic_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                           grads, boxes, box_ind, total, batch, H, W, D, ch, cw, cd, depth, grads_image);
    }
    return check_launch();
}

}  // extern "C"
