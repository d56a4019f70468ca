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

#include <type_traits>
#include "roi_align_common.h"

using namespace mdt_ra;

namespace {

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
constexpr int FWD_THREADS = 256;
constexpr int FWD_PER_THREAD = 4;
constexpr int FWD_SLAB = FWD_THREADS * FWD_PER_THREAD;

// DIM == 3: image [B,C,H,W,D], boxes [N,6], crops [N,C,ch,cw,cd]
// DIM == 2: image [B,C,H,W],   boxes [N,4], crops [N,C,ch,cw]      (D = cd = 1)
template <int DIM, typename TIN>
__device__ __forceinline__ void crop_fwd_body(
    const TIN *__restrict__ image, const float *__restrict__ boxes,
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
        const TIN *pimage = image + ((long long)b_in * C + c) * vol;

        if (DIM == 3) {
            const AxisEntry ez = tab[ch + cw + z];
            const int front = ez.lo, back = entry_hi(ez);
            const long long rt_l = (long long)D * (left + (long long)W * top);
            const long long rt_r = (long long)D * (right + (long long)W * top);
            const long long rb_l = (long long)D * (left + (long long)W * bottom);
            const long long rb_r = (long long)D * (right + (long long)W * bottom);
            const float tlf = ld(pimage, front + rt_l), trf = ld(pimage, front + rt_r);
            const float blf = ld(pimage, front + rb_l), brf = ld(pimage, front + rb_r);
            const float tlb = ld(pimage, back + rt_l), trb = ld(pimage, back + rt_r);
            const float blb = ld(pimage, back + rb_l), brb = ld(pimage, back + rb_r);
            const float top_front = tlf + (trf - tlf) * ex.lerp;
            const float bottom_front = blf + (brf - blf) * ex.lerp;
            const float top_back = tlb + (trb - tlb) * ex.lerp;
            const float bottom_back = blb + (brb - blb) * ex.lerp;
            const float frontv = top_front + (bottom_front - top_front) * ey.lerp;
            const float backv = top_back + (bottom_back - top_back) * ey.lerp;
            out[e] = frontv + (backv - frontv) * ez.lerp;
        } else {
            const float tl = ld(pimage, (long long)top * W + left);
            const float tr = ld(pimage, (long long)top * W + right);
            const float bl = ld(pimage, (long long)bottom * W + left);
            const float br = ld(pimage, (long long)bottom * W + right);
            const float topv = tl + (tr - tl) * ex.lerp;
            const float bottomv = bl + (br - bl) * ex.lerp;
            out[e] = topv + (bottomv - topv) * ey.lerp;
        }
    }
}

template <int DIM, typename TIN>
__global__ __launch_bounds__(FWD_THREADS) void crop_fwd_kernel(
    const TIN *__restrict__ image, const float *__restrict__ boxes,
    const int *__restrict__ box_ind, int B, int H, int W, int D,
    int ch, int cw, int cd, int C, float *__restrict__ crops)
{
    crop_fwd_body<DIM, TIN>(image, boxes, box_ind, B, H, W, D, ch, cw, cd, C, crops);
}

// All pyramid levels in one launch (mrcnn.py:373-457 pools every RoI on exactly one level and restores the order:
// here the RoI's workgroups read their level's map directly and write the RoI's row, so the order never changes).
template <int DIM, typename TIN>
__global__ __launch_bounds__(FWD_THREADS) void crop_fwd_pyramid_kernel(
    PyramidMaps maps, const float *__restrict__ boxes, const int *__restrict__ box_ind, const int *__restrict__ level,
    int B, int ch, int cw, int cd, int C, float *__restrict__ crops)
{
    int l = level[blockIdx.x];
    int b_limit = B;
    if (l < 0 || l >= maps.n_levels) { l = 0; b_limit = 0; }      // no level: the row is zero-filled like a skipped RoI
    crop_fwd_body<DIM, TIN>(reinterpret_cast<const TIN *>(maps.image[l]), boxes, box_ind, b_limit,
                            maps.H[l], maps.W[l], maps.D[l], ch, cw, cd, C, crops);
}

// ---------------------------------------------------------------------------
// backward, gather form (LDS tile, footprint-compacted)
// ---------------------------------------------------------------------------
// Internal axes (slow -> fast): 3D (y, x, z);  2D (y, x).  The contiguous axis is
// vectorised: VEC = 4 when its extent is a multiple of 4 (16-byte stores), else 1.
//
// A workgroup walks a contiguous run of tiles; a tile is 1024 vector units (16 KB
// for VEC = 4) of one (b, c) volume.  Per tile:
//   1. every thread compares "its" RoI's (box_ind, y-footprint) -- cached in
//      registers for the first 256 RoIs -- with the tile; ballot + prefix gives an
//      ORDERED list of the RoIs reaching the tile.
//   2. none: stream zeros (the common case: the op is a 151 MB fill).
//   3. else the tile lives in LDS.  Per pass of up to BWD_TB RoIs: sample tables
//      (floor index, lerp), per-index sample ranges, gradient slabs are staged in
//      LDS; then, RoI by RoI, the threads spread over the voxels of
//      footprint(RoI) x tile -- every lane busy, one owner per voxel, no atomics --
//      and add that RoI's terms to the voxel's running sum in the order the
//      sequential reference loop would (sample y, x, z; corner order of
//      kernel.cu:256-301).  Finally the tile is streamed out with 16-byte stores.
constexpr int BWD_THREADS = 256;
constexpr int BWD_K = 4;                         // vector units per thread per tile
constexpr int BWD_TILE_UNITS = BWD_THREADS * BWD_K;
constexpr int BWD_TB_MAX = 4;                    // RoIs staged per pass
constexpr int BWD_SLAB_FLOATS = 1024;            // LDS floats per staged RoI gradient slab

struct PRange {
    short first, last;                           // sample indices p whose [lo, hi] contains this voxel index
};

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
    int tb;             // RoIs per pass (<= BWD_TB_MAX)
    int rows_cap;       // max y rows a tile can span
};

struct RoiMeta {
    int r, py0, npy, staged;
    int vy0, ny, fxlo, nx, fzlo, nz;
    int pad0, pad1;
};

__device__ __forceinline__ void roi_y_footprint(const float *bx, int H, int ph, int &lo, int &hi)
{
    const AxisEntry e0 = axis_entry(bx[0], bx[2], H, ph, 0);
    const AxisEntry e1 = axis_entry(bx[0], bx[2], H, ph, ph - 1);
    lo = min(e0.lo, e1.lo);
    hi = max(entry_hi(e0), entry_hi(e1));
}

template <int DIM, int VEC>
__global__ __launch_bounds__(BWD_THREADS) void crop_bwd_gather_kernel(BwdParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int psum = p.ph + p.pw + p.pd;
    const int TB = p.tb;
    // ---- LDS carve (every offset a multiple of 16) ----
    float *tile = reinterpret_cast<float *>(smem_raw);                       // [BWD_TILE_UNITS * VEC]
    float *slab = tile + BWD_TILE_UNITS * VEC;                               // [TB][BWD_SLAB_FLOATS]
    AxisEntry *tab = reinterpret_cast<AxisEntry *>(slab + TB * BWD_SLAB_FLOATS);   // [TB][psum]
    const int rng_stride = p.rows_cap + p.W + p.D;
    PRange *rng = reinterpret_cast<PRange *>(tab + ((TB * psum + 1) & ~1));  // [TB][rows_cap + W + D]
    RoiMeta *meta = reinterpret_cast<RoiMeta *>(rng + ((TB * rng_stride + 3) & ~3));  // [TB]
    int *list = reinterpret_cast<int *>(meta + TB);                          // [BWD_THREADS]
    int *wave_cnt = list + BWD_THREADS;                                      // [4]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int Dv = p.D / VEC;
    const int P = p.ph * p.pw * p.pd;
    const int row_units = (DIM == 3) ? p.W * Dv : (p.W / VEC);  // vector units per y row

    // descriptor of "my" RoI of the first chunk, cached across tiles
    int my_b = -1, my_ylo = 0, my_yhi = -1;
    if (tid < p.N) {
        my_b = p.box_ind[tid];
        if (my_b >= 0 && my_b < p.B) roi_y_footprint(p.boxes + (long long)tid * (2 * DIM), p.H, p.ph, my_ylo, my_yhi);
        else my_b = -1;
    }

    const long long per_wg = (p.tiles_total + gridDim.x - 1) / gridDim.x;
    long long t0 = (long long)blockIdx.x * per_wg;
    long long t1 = t0 + per_wg;
    if (t1 > p.tiles_total) t1 = p.tiles_total;

    for (long long tl = t0; tl < t1; ++tl) {
        const int vol = (int)(tl / p.tiles_per_vol);
        const int chunk = (int)(tl % p.tiles_per_vol);
        const int b = vol / p.C;
        const int c = vol % p.C;
        const int u_base = chunk * BWD_TILE_UNITS;
        int u_end = u_base + BWD_TILE_UNITS;
        if (u_end > p.units_per_vol) u_end = p.units_per_vol;
        const int y_lo = u_base / row_units;
        const int y_hi = (u_end - 1) / row_units;
        float *ovol = p.out + (long long)vol * p.units_per_vol * VEC;
        bool tile_dirty = false;   // uniform: LDS tile holds data

        for (int rb = 0; rb < p.N; rb += BWD_THREADS) {
            bool hit;
            if (rb == 0) {
                hit = (my_b == b) && (my_ylo <= y_hi) && (my_yhi >= y_lo);
            } else {
                hit = false;
                const int r = rb + tid;
                if (r < p.N && p.box_ind[r] == b) {
                    int lo, hi;
                    roi_y_footprint(p.boxes + (long long)r * (2 * DIM), p.H, p.ph, lo, hi);
                    hit = (lo <= y_hi) && (hi >= y_lo);
                }
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
            if (total == 0) { __syncthreads(); continue; }   // wave_cnt is rewritten by the next chunk
            if (hit) list[off + __popcll(bal & ((1ULL << lane) - 1ULL))] = rb + tid;
            if (!tile_dirty) {
                tile_dirty = true;
#pragma unroll
                for (int k = 0; k < BWD_K; ++k) {
                    if (VEC == 4) reinterpret_cast<v4f *>(tile)[k * BWD_THREADS + tid] = v4f{0.f, 0.f, 0.f, 0.f};
                    else tile[k * BWD_THREADS + tid] = 0.0f;
                }
            }
            __syncthreads();

            for (int lb = 0; lb < total; lb += TB) {
                const int nb = min(TB, total - lb);
                // (1) sample tables
                for (int t = tid; t < nb * psum; t += BWD_THREADS) {
                    const int j = t / psum;
                    const int q = t - j * psum;
                    const float *bx = p.boxes + (long long)list[lb + j] * (2 * DIM);
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
                // (2) per-RoI metadata
                if (tid < nb) {
                    const AxisEntry *ty = tab + tid * psum;
                    const AxisEntry *tx = ty + p.ph;
                    const AxisEntry *tz = tx + p.pw;
                    int py0 = p.ph, py1 = -1, fylo = 0x7fffffff, fyhi = -1;
                    for (int q = 0; q < p.ph; ++q) {
                        const int lo = ty[q].lo, hi = entry_hi(ty[q]);
                        if (lo <= y_hi && hi >= y_lo) { py0 = min(py0, q); py1 = max(py1, q); }
                        fylo = min(fylo, lo); fyhi = max(fyhi, hi);
                    }
                    int fxlo = 0x7fffffff, fxhi = -1;
                    for (int q = 0; q < p.pw; ++q) { fxlo = min(fxlo, tx[q].lo); fxhi = max(fxhi, entry_hi(tx[q])); }
                    int fzlo = 0, fzhi = 0;
                    if (DIM == 3) {
                        fzlo = 0x7fffffff; fzhi = -1;
                        for (int q = 0; q < p.pd; ++q) { fzlo = min(fzlo, tz[q].lo); fzhi = max(fzhi, entry_hi(tz[q])); }
                    }
                    RoiMeta m;
                    m.r = list[lb + tid];
                    m.py0 = py0;
                    m.npy = (py1 >= py0) ? (py1 - py0 + 1) : 0;
                    m.staged = (m.npy * p.pw * p.pd <= BWD_SLAB_FLOATS) ? 1 : 0;
                    m.vy0 = max(fylo, y_lo);
                    m.ny = (m.npy > 0) ? (min(fyhi, y_hi) - m.vy0 + 1) : 0;
                    if (m.ny < 0) m.ny = 0;
                    m.fxlo = fxlo; m.nx = fxhi - fxlo + 1;
                    m.fzlo = fzlo; m.nz = fzhi - fzlo + 1;
                    m.pad0 = m.pad1 = 0;
                    meta[tid] = m;
                }
                __syncthreads();
                // (3) per-index sample ranges + gradient slabs
                for (int j = 0; j < nb; ++j) {
                    const RoiMeta m = meta[j];
                    if (m.ny == 0) continue;
                    const AxisEntry *ty = tab + j * psum;
                    PRange *rj = rng + j * rng_stride;
                    const int n_idx = m.ny + m.nx + ((DIM == 3) ? m.nz : 0);
                    for (int t = tid; t < n_idx; t += BWD_THREADS) {
                        const AxisEntry *ta;
                        int np_, idx;
                        if (t < m.ny) { ta = ty; np_ = p.ph; idx = m.vy0 + t; }
                        else if (t < m.ny + m.nx) { ta = ty + p.ph; np_ = p.pw; idx = m.fxlo + (t - m.ny); }
                        else { ta = ty + p.ph + p.pw; np_ = p.pd; idx = m.fzlo + (t - m.ny - m.nx); }
                        int first = 32767, last = -1;
                        for (int q = 0; q < np_; ++q) {
                            const int lo = ta[q].lo, hi = entry_hi(ta[q]);
                            if (lo == idx || hi == idx) { if (first == 32767) first = q; last = q; }
                        }
                        PRange pr;
                        pr.first = (short)first;
                        pr.last = (short)last;
                        // layout: [0, rows_cap) y | [rows_cap, rows_cap + W) x | then z
                        const int slot = (t < m.ny) ? t : (t < m.ny + m.nx) ? (p.rows_cap + (t - m.ny))
                                                                           : (p.rows_cap + p.W + (t - m.ny - m.nx));
                        rj[slot] = pr;
                    }
                    if (m.staged) {
                        const int cnt = m.npy * p.pw * p.pd;
                        const float *src = p.grads + ((long long)m.r * p.C + c) * P + (long long)m.py0 * p.pw * p.pd;
                        float *dst = slab + j * BWD_SLAB_FLOATS;
                        for (int t = tid; t < cnt; t += BWD_THREADS) dst[t] = src[t];
                    }
                }
                __syncthreads();

                // (4) gather, RoI by RoI (order matters for bit-exactness and for voxel ownership)
                for (int j = 0; j < nb; ++j) {
                    const RoiMeta m = meta[j];
                    const int nvox = m.ny * m.nx * ((DIM == 3) ? m.nz : 1);
                    if (nvox > 0) {
                        const AxisEntry *ty = tab + j * psum;
                        const AxisEntry *tx = ty + p.ph;
                        const AxisEntry *tz = tx + p.pw;
                        const PRange *ry = rng + j * rng_stride;
                        const PRange *rx = ry + p.rows_cap;
                        const PRange *rz = rx + p.W;
                        const float *gsl = m.staged ? (slab + j * BWD_SLAB_FLOATS)
                                                    : (p.grads + ((long long)m.r * p.C + c) * P + (long long)m.py0 * p.pw * p.pd);
                        for (int v = tid; v < nvox; v += BWD_THREADS) {
                            int vz = 0, rest = v;
                            if (DIM == 3) { vz = v % m.nz; rest = v / m.nz; }
                            const int vx = rest % m.nx;
                            const int vyi = rest / m.nx;
                            const int y = m.vy0 + vyi, x = m.fxlo + vx, z = m.fzlo + vz;
                            int local;
                            if (DIM == 3) {
                                const int u = (y * p.W + x) * Dv + z / VEC;
                                if (u < u_base || u >= u_end) continue;
                                local = (u - u_base) * VEC + (z % VEC);
                            } else {
                                const int u = y * row_units + x / VEC;
                                if (u < u_base || u >= u_end) continue;
                                local = (u - u_base) * VEC + (x % VEC);
                            }
                            const PRange qy = ry[vyi], qx = rx[vx];
                            float acc = tile[local];
                            for (int py = qy.first; py <= qy.last; ++py) {
                                const AxisEntry ey = ty[py];
                                const bool mt = (ey.lo == y), mb = (entry_hi(ey) == y);
                                if (!(mt || mb)) continue;
                                const float wyt = 1.0f - ey.lerp, wyb = ey.lerp;
                                for (int px = qx.first; px <= qx.last; ++px) {
                                    const AxisEntry ex = tx[px];
                                    const bool ml = (ex.lo == x), mr = (entry_hi(ex) == x);
                                    if (!(ml || mr)) continue;
                                    const float wxl = 1.0f - ex.lerp, wxr = ex.lerp;
                                    if (DIM == 3) {
                                        const PRange qz = rz[vz];
                                        const float *grow = gsl + ((py - m.py0) * p.pw + px) * p.pd;
                                        for (int pz = qz.first; pz <= qz.last; ++pz) {
                                            const AxisEntry ez = tz[pz];
                                            const bool mf = (ez.lo == z), mk = (entry_hi(ez) == z);
                                            if (!(mf || mk)) continue;
                                            const float g = grow[pz];
                                            const float wzf = 1.0f - ez.lerp, wzb = ez.lerp;
                                            // corner order of kernel.cu:256-301 (y outer, x middle, z inner);
                                            // weight product order (wx * wz) * wy * g
                                            if (mt) {
                                                if (ml) {
                                                    if (mf) acc = acc + wxl * wzf * wyt * g;
                                                    if (mk) acc = acc + wxl * wzb * wyt * g;
                                                }
                                                if (mr) {
                                                    if (mf) acc = acc + wxr * wzf * wyt * g;
                                                    if (mk) acc = acc + wxr * wzb * wyt * g;
                                                }
                                            }
                                            if (mb) {
                                                if (ml) {
                                                    if (mf) acc = acc + wxl * wzf * wyb * g;
                                                    if (mk) acc = acc + wxl * wzb * wyb * g;
                                                }
                                                if (mr) {
                                                    if (mf) acc = acc + wxr * wzf * wyb * g;
                                                    if (mk) acc = acc + wxr * wzb * wyb * g;
                                                }
                                            }
                                        }
                                    } else {
                                        // 2D kernel.cu:175-192: dtop = (1-y_lerp)*g then (1-x_lerp)*dtop, x_lerp*dtop; then bottom
                                        const float g = gsl[(py - m.py0) * p.pw + px];
                                        if (mt) {
                                            const float dtop = wyt * g;
                                            if (ml) acc = acc + wxl * dtop;
                                            if (mr) acc = acc + wxr * dtop;
                                        }
                                        if (mb) {
                                            const float dbot = wyb * g;
                                            if (ml) acc = acc + wxl * dbot;
                                            if (mr) acc = acc + wxr * dbot;
                                        }
                                    }
                                }
                            }
                            tile[local] = acc;
                        }
                    }
                    __syncthreads();
                }
            }
        }

        // ---- stream the tile out ----
#pragma unroll
        for (int k = 0; k < BWD_K; ++k) {
            const int u = u_base + k * BWD_THREADS + tid;
            if (u >= u_end) continue;
            if (VEC == 4) {
                v4f o = {0.f, 0.f, 0.f, 0.f};
                if (tile_dirty) o = reinterpret_cast<const v4f *>(tile)[k * BWD_THREADS + tid];
                reinterpret_cast<v4f *>(ovol)[u] = o;
            } else {
                const float o = tile_dirty ? tile[k * BWD_THREADS + tid] : 0.0f;
                ovol[u] = o;
            }
        }
        if (tile_dirty) __syncthreads();   // tile is re-zeroed by the next dirty tile
    }
}

// ---------------------------------------------------------------------------
// backward, default: separable two-phase form (deterministic, atomic-free)
// ---------------------------------------------------------------------------
// dF[b,c,iy,ix,iz] = sum_r sum_{py,px,pz} g[r,c,py,px,pz] * Wy_r[py,iy] * Wx_r[px,ix] * Wz_r[pz,iz]
// with each W row holding <= 2 non-zeros ((1-lerp) at floor, lerp at ceil).
//   phase A (expand): one workgroup per (RoI, channel group).  The touched indices per axis are
//     compressed (<= 2P of them); the gradient block is pushed through (Wz,Wx) and Wy in LDS (two
//     dense stages, every lane busy, next channel's block prefetched meanwhile) and the compact
//     block E[r,c] (<= 8P floats) goes to the workspace, together with a per-RoI header and the
//     index -> compact-position tables.
//   phase B (fill + gather): a workgroup walks a contiguous run of 16 KB tiles of the gradient
//     feature map.  The headers/tables of the current batch element's RoIs are cached in LDS
//     (rebuilt only when the batch index changes), so the per-tile work has no barriers: a voxel
//     adds, in RoI order, one value per overlapping RoI read from E, and the tile leaves with
//     16-byte non-temporal stores.  Tiles no RoI reaches are a pure zero stream -- for the
//     reference shapes this IS the op (151 MB of 158 MB).
// Summation is reassociated relative to the reference's flat 8-corner scatter, so results agree to
// fp32 rounding (~1e-6 of the summed magnitudes; bar 1e-4) rather than bit-for-bit; run-to-run it
// is deterministic.
typedef unsigned long long u64;

constexpr int EXP_THREADS = 256;
constexpr int EXP_LDS_FLOATS_MAX = 12288;   // 2*gl + t2 budget (48 KB)

constexpr int FILL_LIST = 256;              // RoIs of one batch element handled per list chunk
constexpr int HDR_INTS = 12;                // b, ylo, yhi, xlo, xhi, zlo, zhi, nuy, nux, nuz, pad, pad

struct FastParams {
    const float *grads;
    const float *boxes;
    const int *box_ind;
    float *out;
    int *hdr;            // workspace: [N][HDR_INTS]
    short *pos;          // workspace: [N][pos_stride]  (y | x | z tables, -1 = untouched)
    short *ul;           // workspace: [N][ul_stride]   (Uy | Ux | Uz: voxel index per compact position)
    int ul_stride, ul_x_off, ul_z_off;
    float *E;            // workspace: [N][C][slot_floats]
    int N, B, C;
    int H, W, D;
    int ph, pw, pd;
    int slot_floats;     // 8 * P (2D: 4 * P)
    int ch_per_wg;
    int wy, wx, wz;      // u64 words per axis bitmap
    int pos_x_off, pos_z_off, pos_stride;   // H4, H4 + W4, H4 + W4 + D4 (each rounded up to 4)
    int units_per_vol, tiles_per_vol;
    long long tiles_total;
    int lds_cache;       // RoIs whose header/tables phase B caches in LDS
};

__device__ __forceinline__ int bitmap_pos(const u64 *words, const int *prefix, int idx)
{
    const u64 w = words[idx >> 6];
    const int bit = idx & 63;
    if (!((w >> bit) & 1ULL)) return -1;
    return prefix[idx >> 6] + __popcll(w & ((1ULL << bit) - 1ULL));
}

// ---- phase A -----------------------------------------------------------------
template <int DIM>
__device__ __forceinline__ void expand_role(const FastParams &p, char *smem_raw, const int r, const int cgroup)
{
    const int tid = threadIdx.x;
    const int b_in = p.box_ind[r];
    if (b_in < 0 || b_in >= p.B) {
        if (cgroup == 0 && tid == 0) p.hdr[(long long)r * HDR_INTS] = -1;
        return;
    }
    const int psum = p.ph + p.pw + p.pd;
    const int P = p.ph * p.pw * p.pd;
    const int nwords = p.wy + p.wx + p.wz;
    const int nuy_max = 2 * p.ph, nux_max = 2 * p.pw, nuz_max = (DIM == 3) ? 2 * p.pd : 1;
    const int P4 = (P + 3) & ~3;

    // LDS carve
    float *gl0 = reinterpret_cast<float *>(smem_raw);                // [2][P4]  double-buffered gradient block
    float *t2 = gl0 + 2 * P4;                                        // [ph][nux][nuz]
    u64 *bits = reinterpret_cast<u64 *>(t2 + ((p.ph * nux_max * nuz_max + 3) & ~3));
    AxisEntry *tab = reinterpret_cast<AxisEntry *>(bits + nwords);   // [psum]
    int *prefix = reinterpret_cast<int *>(tab + psum);               // [nwords]
    int *nu = prefix + nwords;                                       // [4]
    short *U = reinterpret_cast<short *>(nu + 4);                    // touched index per compact position
    PRange *R = reinterpret_cast<PRange *>(U + ((nuy_max + nux_max + nuz_max + 1) & ~1));  // sample range per position

    const int c0 = cgroup * p.ch_per_wg;
    const int c1 = min(p.C, c0 + p.ch_per_wg);
    // prefetch the first channel's gradient block while the tables are being built
    constexpr int GREG = 8;   // supports P <= 2048 through registers; larger blocks are loaded directly
    float greg[GREG];
    const bool use_reg = (P <= GREG * EXP_THREADS);
    {
        const float *src = p.grads + ((long long)r * p.C + c0) * P;
        if (use_reg) {
#pragma unroll
            for (int q = 0; q < GREG; ++q) { const int t = tid + q * EXP_THREADS; greg[q] = (t < P) ? src[t] : 0.0f; }
        }
    }

    const float *bx = p.boxes + (long long)r * (2 * DIM);
    int *fl = reinterpret_cast<int *>(R + ((nuy_max + nux_max + nuz_max + 1) & ~1));   // [2][H + W + D] first / last sample per index
    const int Ltot = p.H + p.W + ((DIM == 3) ? p.D : 0);
    for (int t = tid; t < nwords; t += EXP_THREADS) bits[t] = 0ULL;
    for (int t = tid; t < Ltot; t += EXP_THREADS) { fl[t] = 32767; fl[Ltot + t] = -1; }
    __syncthreads();
    for (int q = tid; q < psum; q += EXP_THREADS) {
        AxisEntry e;
        u64 *bw;
        int base, pq;
        if (q < p.ph) { e = axis_entry(bx[0], bx[2], p.H, p.ph, q); bw = bits; base = 0; pq = q; }
        else if (q < p.ph + p.pw) { e = axis_entry(bx[1], bx[3], p.W, p.pw, q - p.ph); bw = bits + p.wy; base = p.H; pq = q - p.ph; }
        else {
            bw = bits + p.wy + p.wx; base = p.H + p.W; pq = q - p.ph - p.pw;
            if (DIM == 3) e = axis_entry(bx[4], bx[5], p.D, p.pd, q - p.ph - p.pw);
            else { e.lo = 0; e.lerp = 0.0f; }
        }
        tab[q] = e;
        const int hi = entry_hi(e);
        atomicOr(&bw[e.lo >> 6], 1ULL << (e.lo & 63));
        atomicOr(&bw[hi >> 6], 1ULL << (hi & 63));
        if (DIM == 3 || q < p.ph + p.pw) {
            // first / last sample touching each voxel index: one LDS atomicMin/Max per (sample, floor|ceil)
            atomicMin(&fl[base + e.lo], pq); atomicMax(&fl[Ltot + base + e.lo], pq);
            atomicMin(&fl[base + hi], pq);   atomicMax(&fl[Ltot + base + hi], pq);
        }
    }
    __syncthreads();
    if (tid < 3) {
        const int off = (tid == 0) ? 0 : (tid == 1) ? p.wy : p.wy + p.wx;
        const int nw = (tid == 0) ? p.wy : (tid == 1) ? p.wx : p.wz;
        int run = 0;
        for (int w = 0; w < nw; ++w) { prefix[off + w] = run; run += __popcll(bits[off + w]); }
        nu[tid] = run;
    }
    __syncthreads();
    const int nuy = nu[0], nux = nu[1], nuz = (DIM == 3) ? nu[2] : 1;

    // touched index list U, per-position sample range R; the c-group-0 workgroup also publishes
    // the header and the index -> position tables for phase B.
    {
        const int L[3] = {p.H, p.W, (DIM == 3) ? p.D : 1};
        const int woff[3] = {0, p.wy, p.wy + p.wx};
        const int uoff[3] = {0, nuy_max, nuy_max + nux_max};
        const int loff[3] = {0, p.H, p.H + p.W};
        const int goff[3] = {0, p.pos_x_off, p.pos_z_off};
        short *gpos = p.pos + (long long)r * p.pos_stride;
        for (int a = 0; a < DIM; ++a) {
            for (int idx = tid; idx < L[a]; idx += EXP_THREADS) {
                const int pos = bitmap_pos(bits + woff[a], prefix + woff[a], idx);
                if (cgroup == 0) gpos[goff[a] + idx] = (short)pos;
                if (pos < 0) continue;
                U[uoff[a] + pos] = (short)idx;
                PRange pr; pr.first = (short)fl[loff[a] + idx]; pr.last = (short)fl[Ltot + loff[a] + idx];
                R[uoff[a] + pos] = pr;
            }
        }
        __syncthreads();
        if (cgroup == 0) {
            short *gul = p.ul + (long long)r * p.ul_stride;
            for (int t = tid; t < nuy; t += EXP_THREADS) gul[t] = U[t];
            for (int t = tid; t < nux; t += EXP_THREADS) gul[p.ul_x_off + t] = U[nuy_max + t];
            if (DIM == 3) for (int t = tid; t < nuz; t += EXP_THREADS) gul[p.ul_z_off + t] = U[nuy_max + nux_max + t];
        }
        if (cgroup == 0 && tid == 0) {
            const int nus[3] = {nuy, nux, nuz};
            int lo_idx[3] = {0, 0, 0}, hi_idx[3] = {0, 0, 0};
            for (int a = 0; a < DIM; ++a) { lo_idx[a] = U[uoff[a]]; hi_idx[a] = U[uoff[a] + nus[a] - 1]; }
            int *h = p.hdr + (long long)r * HDR_INTS;
            h[0] = b_in;
            h[1] = lo_idx[0]; h[2] = hi_idx[0];
            h[3] = lo_idx[1]; h[4] = hi_idx[1];
            h[5] = lo_idx[2]; h[6] = hi_idx[2];
            h[7] = nuy; h[8] = nux; h[9] = nuz; h[10] = 0; h[11] = 0;
        }
    }

    const AxisEntry *ty = tab, *tx = tab + p.ph, *tz = tab + p.ph + p.pw;
    const short *Uy = U, *Ux = U + nuy_max, *Uz = U + nuy_max + nux_max;
    const PRange *Ry = R, *Rx = R + nuy_max, *Rz = R + nuy_max + nux_max;

    for (int c = c0; c < c1; ++c) {
        float *gl = gl0 + ((c - c0) & 1) * P4;
        if (use_reg) {
#pragma unroll
            for (int q = 0; q < GREG; ++q) { const int t = tid + q * EXP_THREADS; if (t < P) gl[t] = greg[q]; }
        } else {
            const float *src = p.grads + ((long long)r * p.C + c) * P;
            for (int t = tid; t < P; t += EXP_THREADS) gl[t] = src[t];
        }
        __syncthreads();   // gl visible; also orders the previous channel's Y-stage reads of t2 before this XZ stage
        if (use_reg && c + 1 < c1) {   // next channel's block flies during this channel's stages
            const float *src = p.grads + ((long long)r * p.C + c + 1) * P;
#pragma unroll
            for (int q = 0; q < GREG; ++q) { const int t = tid + q * EXP_THREADS; greg[q] = (t < P) ? src[t] : 0.0f; }
        }
        // XZ stage: t2[py][ix][iz] = sum_px wx * sum_pz wz * g[py][px][pz]
        {
            const int n2 = p.ph * nux * nuz;
            for (int o = tid; o < n2; o += EXP_THREADS) {
                const int iz = o % nuz;
                const int rest = o / nuz;
                const int ix = rest % nux;
                const int py = rest / nux;
                const int xidx = Ux[ix];
                const PRange prx = Rx[ix];
                float acc = 0.0f;
                if (DIM == 3) {
                    const int zidx = Uz[iz];
                    const PRange prz = Rz[iz];
                    for (int qx = prx.first; qx <= prx.last; ++qx) {
                        const float *grow = gl + (py * p.pw + qx) * p.pd;
                        float az = 0.0f;
                        for (int qz = prz.first; qz <= prz.last; ++qz) az = az + axis_weight(tz[qz], zidx) * grow[qz];
                        acc = acc + axis_weight(tx[qx], xidx) * az;
                    }
                } else {
                    for (int qx = prx.first; qx <= prx.last; ++qx)
                        acc = acc + axis_weight(tx[qx], xidx) * gl[py * p.pw + qx];
                }
                t2[o] = acc;
            }
        }
        __syncthreads();
        // Y stage: E[iy][ix][iz] = sum_py wy * t2[py][ix][iz]
        {
            float *dst = p.E + ((long long)r * p.C + c) * p.slot_floats;
            const int plane = nux * nuz;
            const int n3 = nuy * plane;
            for (int o = tid; o < n3; o += EXP_THREADS) {
                const int iy = o / plane;
                const int rem = o - iy * plane;
                const int idx = Uy[iy];
                const PRange pr = Ry[iy];
                float acc = 0.0f;
                for (int q = pr.first; q <= pr.last; ++q) acc = acc + axis_weight(ty[q], idx) * t2[q * plane + rem];
                dst[o] = acc;
            }
        }
        // no barrier here: the next iteration writes the OTHER gl buffer, and its first barrier
        // separates this Y stage (reads t2) from the next XZ stage (writes t2)
    }
}

// Kernel 1: role-split launch.  Workgroups [0, n_expand) expand one (RoI, channel group) each;
// the remaining workgroups stream zeros over the whole gradient feature map.  The two roles are
// independent (no inter-workgroup communication), so the expand work hides under the fill.
template <int DIM>
__global__ __launch_bounds__(EXP_THREADS) void crop_bwd_expand_zero_kernel(FastParams p, int n_expand, int groups_per_roi,
                                                                            long long n_vec4, long long n_scalar_tail_begin,
                                                                            long long n_total)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // interleave the two roles in block order so both kinds are resident from the start
    const int nzero = (int)gridDim.x - n_expand;
    const int npair = min(n_expand, nzero);
    int role_expand, idx;
    if ((int)blockIdx.x < 2 * npair) { role_expand = !(blockIdx.x & 1); idx = blockIdx.x >> 1; }
    else { role_expand = n_expand > nzero; idx = blockIdx.x - npair; }
    if (role_expand) {
        expand_role<DIM>(p, smem_raw, idx / groups_per_roi, idx % groups_per_roi);
        return;
    }
    const long long zb = idx;
    const long long nz = nzero;
    const v4f z = {0.f, 0.f, 0.f, 0.f};
    v4f *o4 = reinterpret_cast<v4f *>(p.out);
    // plain stores: measured 22.4 us for 151 MB (6.7 TB/s) vs 32 us with the non-temporal hint on gfx950
    for (long long i = zb * EXP_THREADS + threadIdx.x; i < n_vec4; i += nz * EXP_THREADS) o4[i] = z;
    for (long long i = n_scalar_tail_begin + zb * EXP_THREADS + threadIdx.x; i < n_total; i += nz * EXP_THREADS) p.out[i] = 0.0f;
}

// ---- phase B -----------------------------------------------------------------
// Kernel 2: patch.  Overwrites exactly the voxels at least one RoI reaches (everything else was
// zeroed by kernel 1).  One thread per element of a compact block E[r, c]: it maps the element to
// its voxel, sums -- in ascending RoI order -- the contribution of every RoI of that batch element
// whose footprint covers the voxel (position-table lookups into that RoI's compact block) and
// stores the result.  A voxel covered by k RoIs is stored k times with the identical value, so
// there is no ordering hazard, no LDS image and no atomics; the kernel is flat and fully parallel.
constexpr int PATCH_THREADS = 256;
constexpr int PATCH_NB_LDS = 12;            // neighbour RoIs whose position tables are cached in LDS

template <int DIM>
__global__ __launch_bounds__(PATCH_THREADS) void crop_bwd_patch_kernel(FastParams p, int groups_per_roi)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int r = blockIdx.x / groups_per_roi;
    const int cgroup = blockIdx.x % groups_per_roi;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int *hr = p.hdr + (long long)r * HDR_INTS;
    const int b = hr[0];
    if (b < 0) return;
    const int nuy = hr[7], nux = hr[8], nuz = hr[9];
    const int r_ylo = hr[1], r_yhi = hr[2], r_xlo = hr[3], r_xhi = hr[4], r_zlo = hr[5], r_zhi = hr[6];

    // LDS carve
    int *s_nb = reinterpret_cast<int *>(smem_raw);                     // [FILL_LIST] neighbour RoI ids (ascending)
    int *s_nbh = s_nb + FILL_LIST;                                     // [FILL_LIST][HDR_INTS]
    int *wave_cnt = s_nbh + FILL_LIST * HDR_INTS;                      // [4]
    int *s_cnt = wave_cnt + 4;                                         // [4]
    short *s_ul = reinterpret_cast<short *>(s_cnt + 4);                // [ul_stride] Uy|Ux|Uz of RoI r
    short *s_pos = s_ul + p.ul_stride;                                 // [PATCH_NB_LDS][pos_stride]

    for (int t = tid; t < p.ul_stride; t += PATCH_THREADS) s_ul[t] = p.ul[(long long)r * p.ul_stride + t];

    // neighbours: RoIs of the same batch element whose bounding footprint intersects r's (includes r)
    int n_nb = 0;
    for (int rb = 0; rb < p.N; rb += PATCH_THREADS) {
        const int j = rb + tid;
        bool hit = false;
        if (j < p.N) {
            const int *h = p.hdr + (long long)j * HDR_INTS;
            hit = (h[0] == b) && !(h[2] < r_ylo || h[1] > r_yhi || h[4] < r_xlo || h[3] > r_xhi);
            if (DIM == 3) hit = hit && !(h[6] < r_zlo || h[5] > r_zhi);
        }
        const u64 bal = __ballot(hit);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = n_nb, total = 0;
#pragma unroll
        for (int w = 0; w < PATCH_THREADS / 64; ++w) {
            const int cnt = wave_cnt[w];
            if (w < wave) off += cnt;
            total += cnt;
        }
        if (hit) {
            const int slot = off + __popcll(bal & ((1ULL << lane) - 1ULL));
            if (slot < FILL_LIST) s_nb[slot] = j;
        }
        n_nb += total;
        __syncthreads();
    }
    const bool overflow = n_nb > FILL_LIST;       // pathological overlap count: handled by the scan fallback below
    const int nb = overflow ? 0 : n_nb;
    for (int t = tid; t < nb * HDR_INTS; t += PATCH_THREADS) s_nbh[t] = p.hdr[(long long)s_nb[t / HDR_INTS] * HDR_INTS + (t % HDR_INTS)];
    {
        const int ncache = min(nb, PATCH_NB_LDS);
        const int words = p.pos_stride / 2;
        const int *gsrc = reinterpret_cast<const int *>(p.pos);
        int *ldst = reinterpret_cast<int *>(s_pos);
        for (int t = tid; t < ncache * words; t += PATCH_THREADS) {
            const int q = t / words;
            ldst[t] = gsrc[(long long)s_nb[q] * words + (t - q * words)];
        }
    }
    __syncthreads();

    const short *Uy = s_ul, *Ux = s_ul + p.ul_x_off, *Uz = s_ul + p.ul_z_off;
    const int plane = nux * nuz;
    const int n_el = nuy * plane;
    const long long vol = (long long)p.H * p.W * p.D;
    const int c0 = cgroup * p.ch_per_wg;
    const int c1 = min(p.C, c0 + p.ch_per_wg);

    for (int e = tid; e < n_el; e += PATCH_THREADS) {
        const int iy = e / plane;
        const int rem = e - iy * plane;
        int ix, iz = 0;
        if (DIM == 3) { ix = rem / nuz; iz = rem - ix * nuz; } else { ix = rem; }
        const int y = Uy[iy], x = Ux[ix], z = (DIM == 3) ? Uz[iz] : 0;
        const long long vox = ((long long)y * p.W + x) * p.D + z;
        for (int c = c0; c < c1; ++c) {
            float val = 0.0f;
            if (!overflow) {
                for (int q = 0; q < nb; ++q) {
                    const int *h = s_nbh + q * HDR_INTS;
                    if (y < h[1] || y > h[2] || x < h[3] || x > h[4]) continue;
                    if (DIM == 3 && (z < h[5] || z > h[6])) continue;
                    const int j = s_nb[q];
                    const short *pj = (q < PATCH_NB_LDS) ? (s_pos + q * p.pos_stride) : (p.pos + (long long)j * p.pos_stride);
                    const int piy = pj[y];
                    const int pix = pj[p.pos_x_off + x];
                    const int piz = (DIM == 3) ? pj[p.pos_z_off + z] : 0;
                    if (piy < 0 || pix < 0 || piz < 0) continue;
                    const float *Ej = p.E + ((long long)j * p.C + c) * p.slot_floats;
                    val = val + Ej[(piy * h[8] + pix) * h[9] + piz];
                }
            } else {
                for (int j = 0; j < p.N; ++j) {     // rare: more than FILL_LIST overlapping RoIs
                    const int *h = p.hdr + (long long)j * HDR_INTS;
                    if (h[0] != b || y < h[1] || y > h[2] || x < h[3] || x > h[4]) continue;
                    if (DIM == 3 && (z < h[5] || z > h[6])) continue;
                    const short *pj = p.pos + (long long)j * p.pos_stride;
                    const int piy = pj[y];
                    const int pix = pj[p.pos_x_off + x];
                    const int piz = (DIM == 3) ? pj[p.pos_z_off + z] : 0;
                    if (piy < 0 || pix < 0 || piz < 0) continue;
                    const float *Ej = p.E + ((long long)j * p.C + c) * p.slot_floats;
                    val = val + Ej[(piy * h[8] + pix) * h[9] + piz];
                }
            }
            p.out[((long long)b * p.C + c) * vol + vox] = val;
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
        out4[i] = z;
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

template <int DIM, typename TIN>
int launch_fwd(const TIN *image, const float *boxes, const int *box_ind, int N, int B,
               int H, int W, int D, int ch, int cw, int cd, int C, float *crops, hipStream_t s)
{
    if (N < 0 || B <= 0 || H <= 0 || W <= 0 || D <= 0 || ch <= 0 || cw <= 0 || cd <= 0 || C <= 0)
        return MDT_ERR_INVALID_ARGUMENT;
    const long long per_roi = (long long)C * ch * cw * cd;
    if (N == 0 || per_roi == 0) return MDT_OK;
    if (per_roi > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    const size_t tab_bytes = (size_t)(ch + cw + cd) * sizeof(AxisEntry);
    if (tab_bytes > 24 * 1024) return MDT_ERR_UNSUPPORTED;
    // 3D: the channel-quad form (roi_align_fwd.hip, round 5); 2D and what is outside its budgets: the direct gather below.
    // (History, profiles/: a round-1 LDS-staged form lost to the direct gather -- 58 vs 35 us at N = 600 (7,7,3); round 4's wave-staged
    // form reached 41 us at N = 240 (14,14,5) against 57 us direct; both are gone from the product library.)
    if (DIM == 3) {
        PyramidMaps one;
        one.n_levels = 1; one.image[0] = image; one.H[0] = H; one.W[0] = W; one.D[0] = D;
        const int rq = launch_fwd_cq<TIN>(one, boxes, box_ind, nullptr, N, B, ch, cw, cd, C, crops, s);
        if (rq != MDT_ERR_UNSUPPORTED) return rq;
    }
    const long long slabs = (per_roi + FWD_SLAB - 1) / FWD_SLAB;
    if (slabs > 65535) return MDT_ERR_UNSUPPORTED;
    dim3 grid((unsigned)N, (unsigned)slabs);
    (void)hipGetLastError();
    hipLaunchKernelGGL((crop_fwd_kernel<DIM, TIN>), grid, dim3(FWD_THREADS), tab_bytes, s,
                       image, boxes, box_ind, B, H, W, D, ch, cw, cd, C, crops);
    return check_launch();
}

template <int DIM, typename TIN>
int launch_fwd_pyramid(int n_levels, const void *const *images, const int *H, const int *W, const int *D,
                       const float *boxes, const int *box_ind, const int *level, int N, int B,
                       int ch, int cw, int cd, int C, float *crops, hipStream_t s)
{
    if (n_levels < 1 || n_levels > PYR_MAX_LEVELS || N < 0 || B <= 0 || ch <= 0 || cw <= 0 || cd <= 0 || C <= 0)
        return MDT_ERR_INVALID_ARGUMENT;
    PyramidMaps maps;
    maps.n_levels = n_levels;
    for (int l = 0; l < n_levels; ++l) {
        const int Dl = (DIM == 3) ? D[l] : 1;
        if (H[l] <= 0 || W[l] <= 0 || Dl <= 0 || images[l] == nullptr) return MDT_ERR_INVALID_ARGUMENT;
        maps.image[l] = images[l]; maps.H[l] = H[l]; maps.W[l] = W[l]; maps.D[l] = Dl;
    }
    const long long per_roi = (long long)C * ch * cw * cd;
    if (N == 0) return MDT_OK;
    if (per_roi > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    const size_t tab_bytes = (size_t)(ch + cw + cd) * sizeof(AxisEntry);
    const long long slabs = (per_roi + FWD_SLAB - 1) / FWD_SLAB;
    if (tab_bytes > 24 * 1024 || slabs > 65535) return MDT_ERR_UNSUPPORTED;
    if (DIM == 3) {
        const int rq = launch_fwd_cq<TIN>(maps, boxes, box_ind, level, N, B, ch, cw, cd, C, crops, s);
        if (rq != MDT_ERR_UNSUPPORTED) return rq;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL((crop_fwd_pyramid_kernel<DIM, TIN>), dim3((unsigned)N, (unsigned)slabs), dim3(FWD_THREADS), tab_bytes, s,
                       maps, boxes, box_ind, level, B, ch, cw, cd, C, crops);
    return check_launch();
}

template <int DIM>
int launch_bwd(const float *grads, const float *boxes, const int *box_ind, int N, int B,
               int H, int W, int D, int ph, int pw, int pd, int C, float *out, hipStream_t s)
{
    (void)hipGetLastError();   // drop stale error state of earlier runtime calls on this thread
    if (N < 0 || B <= 0 || H <= 0 || W <= 0 || D <= 0 || ph <= 0 || pw <= 0 || pd <= 0 || C <= 0)
        return MDT_ERR_INVALID_ARGUMENT;
    const long long vol = (long long)H * W * D;
    if (vol > 0x7fffffffLL || ph > 32000 || pw > 32000 || pd > 32000) return MDT_ERR_UNSUPPORTED;
    const int contig = (DIM == 3) ? D : W;
    const int vec = (contig % 4 == 0 && (((uintptr_t)out) & 15) == 0) ? 4 : 1;
    BwdParams p;
    p.grads = grads; p.boxes = boxes; p.box_ind = box_ind; p.out = out;
    p.N = N; p.B = B; p.C = C; p.H = H; p.W = W; p.D = D; p.ph = ph; p.pw = pw; p.pd = pd;
    p.units_per_vol = (int)(vol / vec);
    p.tiles_per_vol = (p.units_per_vol + BWD_TILE_UNITS - 1) / BWD_TILE_UNITS;
    p.tiles_total = (long long)B * C * p.tiles_per_vol;
    const int row_units = (DIM == 3) ? W * (D / vec) : (W / vec);
    p.rows_cap = (BWD_TILE_UNITS + row_units - 1) / row_units + 1;
    if (p.rows_cap > H) p.rows_cap = H;
    const int psum = ph + pw + pd;
    size_t lds = 0;
    int tb = BWD_TB_MAX;
    for (; tb >= 1; --tb) {
        const int rng_stride = p.rows_cap + W + D;
        lds = (size_t)BWD_TILE_UNITS * vec * sizeof(float) + (size_t)tb * BWD_SLAB_FLOATS * sizeof(float) +
              (size_t)((tb * psum + 1) & ~1) * sizeof(AxisEntry) + (size_t)((tb * rng_stride + 3) & ~3) * sizeof(PRange) +
              (size_t)tb * sizeof(RoiMeta) + (size_t)(BWD_THREADS + 4) * sizeof(int);
        if (lds <= 40 * 1024) break;
    }
    if (tb < 1) {
        tb = 1;
        if (lds > 64 * 1024) return MDT_ERR_UNSUPPORTED;
    }
    p.tb = tb;
    long long grid = p.tiles_total < 2048 ? p.tiles_total : 2048;
    if (grid <= 0) return MDT_OK;
    if (vec == 4) hipLaunchKernelGGL((crop_bwd_gather_kernel<DIM, 4>), dim3((unsigned)grid), dim3(BWD_THREADS), lds, s, p);
    else hipLaunchKernelGGL((crop_bwd_gather_kernel<DIM, 1>), dim3((unsigned)grid), dim3(BWD_THREADS), lds, s, p);
    return check_launch();
}

struct FastLayout {
    size_t hdr_off, pos_off, ul_off, e_off, total;
    int pos_x_off, pos_z_off, pos_stride, ul_stride, ul_x_off, ul_z_off;
};

inline FastLayout fast_layout(int dim, int N, int C, int H, int W, int D, int ph, int pw, int pd)
{
    FastLayout L;
    const size_t P = (size_t)ph * pw * pd;
    const size_t slot = (dim == 3 ? 8 : 4) * P;
    const int H4 = (H + 3) & ~3, W4 = (W + 3) & ~3, D4 = dim == 3 ? ((D + 3) & ~3) : 0;
    L.pos_x_off = H4; L.pos_z_off = H4 + W4; L.pos_stride = H4 + W4 + D4;
    const size_t n = (size_t)(N > 0 ? N : 0);
    L.hdr_off = 0;
    L.pos_off = (n * HDR_INTS * sizeof(int) + 255) & ~(size_t)255;
    L.ul_x_off = 2 * ph;
    L.ul_z_off = 2 * ph + 2 * pw;
    L.ul_stride = (2 * ph + 2 * pw + (dim == 3 ? 2 * pd : 0) + 3) & ~3;
    L.ul_off = (L.pos_off + n * L.pos_stride * sizeof(short) + 255) & ~(size_t)255;
    L.e_off = (L.ul_off + n * L.ul_stride * sizeof(short) + 255) & ~(size_t)255;
    L.total = (L.e_off + n * C * slot * sizeof(float) + 255) & ~(size_t)255;
    return L;
}

// returns MDT_ERR_UNSUPPORTED when the shape does not fit the LDS budgets (caller falls back)
template <int DIM>
int launch_bwd_fast(const float *grads, const float *boxes, const int *box_ind, int N, int B,
                    int H, int W, int D, int ph, int pw, int pd, int C, float *out,
                    void *ws, size_t ws_bytes, hipStream_t s)
{
    if (N < 0 || B <= 0 || H <= 0 || W <= 0 || D <= 0 || ph <= 0 || pw <= 0 || pd <= 0 || C <= 0)
        return MDT_ERR_INVALID_ARGUMENT;
    const long long vol = (long long)H * W * D;
    if (vol > 0x7fffffffLL || H > 32000 || W > 32000 || D > 32000) return MDT_ERR_UNSUPPORTED;
    const FastLayout L = fast_layout(DIM, N, C, H, W, D, ph, pw, pd);
    if (ws == nullptr || ws_bytes < L.total || (((uintptr_t)ws) & 15) != 0) return MDT_ERR_WORKSPACE_TOO_SMALL;
    const int P = ph * pw * pd;
    const int nuz_max = (DIM == 3) ? 2 * pd : 1;
    FastParams p;
    p.grads = grads; p.boxes = boxes; p.box_ind = box_ind; p.out = out;
    char *wsb = reinterpret_cast<char *>(ws);
    p.hdr = reinterpret_cast<int *>(wsb + L.hdr_off);
    p.pos = reinterpret_cast<short *>(wsb + L.pos_off);
    p.ul = reinterpret_cast<short *>(wsb + L.ul_off);
    p.ul_stride = L.ul_stride; p.ul_x_off = L.ul_x_off; p.ul_z_off = L.ul_z_off;
    p.E = reinterpret_cast<float *>(wsb + L.e_off);
    p.N = N; p.B = B; p.C = C; p.H = H; p.W = W; p.D = D; p.ph = ph; p.pw = pw; p.pd = pd;
    p.slot_floats = (DIM == 3 ? 8 : 4) * P;
    p.wy = (H + 63) / 64; p.wx = (W + 63) / 64; p.wz = (D + 63) / 64;
    p.pos_x_off = L.pos_x_off; p.pos_z_off = L.pos_z_off; p.pos_stride = L.pos_stride;
    p.units_per_vol = 0; p.tiles_per_vol = 0; p.tiles_total = 0;
    const int psum = ph + pw + pd;
    const int nwords = p.wy + p.wx + p.wz;

    // phase A LDS
    const size_t fl = (size_t)2 * ((P + 3) & ~3) + (size_t)((ph * 2 * pw * nuz_max + 3) & ~3);
    if (fl > EXP_LDS_FLOATS_MAX) return MDT_ERR_UNSUPPORTED;
    const int nu_tot = 2 * ph + 2 * pw + nuz_max;
    const size_t ldsA = fl * sizeof(float) + (size_t)nwords * sizeof(u64) + (size_t)psum * sizeof(AxisEntry) +
                        (size_t)(nwords + 4) * sizeof(int) + (size_t)((nu_tot + 1) & ~1) * sizeof(short) +
                        (size_t)((nu_tot + 1) & ~1) * sizeof(PRange) + (size_t)2 * (H + W + D) * sizeof(int) + 16;
    if (ldsA > 64 * 1024) return MDT_ERR_UNSUPPORTED;
    // kernel 2 LDS: neighbour list + headers + own index lists + cached neighbour position tables
    const size_t ldsB = (size_t)(FILL_LIST + FILL_LIST * HDR_INTS + 8) * sizeof(int) +
                        (size_t)(L.ul_stride + PATCH_NB_LDS * L.pos_stride) * sizeof(short) + 16;
    if (ldsB > 64 * 1024) return MDT_ERR_UNSUPPORTED;
    p.lds_cache = 0;
    // kernel 1: expand role || zero-fill role
    int n_expand = 0, gy = 1;
    p.ch_per_wg = 1;
    if (N > 0) {
        // one channel per expand workgroup up to ~2048 workgroups: more channels per workgroup measured slower
        // (43-66 us vs 38 us) although it would amortise the per-RoI table build
        int cpw = (int)(((long long)N * C + 2047) / 2048);
        if (cpw < 1) cpw = 1;
        if (cpw > C) cpw = C;
        p.ch_per_wg = cpw;
        gy = (C + cpw - 1) / cpw;
        if ((long long)N * gy > 0x3fffffffLL) return MDT_ERR_UNSUPPORTED;
        n_expand = N * gy;
    }
    const long long n_total = (long long)B * C * vol;
    const bool aligned = (((uintptr_t)out) & 15) == 0;
    const long long n_vec4 = aligned ? n_total / 4 : 0;
    const long long tail_begin = n_vec4 * 4;
    long long n_zero = (n_total / 4 + EXP_THREADS - 1) / EXP_THREADS;
    if (n_zero > 4096) n_zero = 4096;
    if (n_zero < 1) n_zero = 1;
    // (a variant with the zero-fill on a forked internal stream measured 57-65 us vs 29-52 us: the event
    //  fork/join costs more than the freed workgroup slots gain -- DESIGN.md 4.1)
    (void)hipGetLastError();
    hipLaunchKernelGGL(crop_bwd_expand_zero_kernel<DIM>, dim3((unsigned)(n_expand + n_zero)), dim3(EXP_THREADS), ldsA, s,
                       p, n_expand, gy, n_vec4, tail_begin, n_total);
    if (check_launch() != MDT_OK) return MDT_ERR_LAUNCH_FAILED;
    if (N == 0) return MDT_OK;
    // kernel 2: patch the touched voxels, one workgroup per (RoI, channel group)
    (void)hipGetLastError(); hipLaunchKernelGGL(crop_bwd_patch_kernel<DIM>, dim3((unsigned)n_expand), dim3(PATCH_THREADS), ldsB, s, p, gy);
    return check_launch();
}

// The single-launch territory form walks the RoIs of one batch element inside one workgroup (rounds of <= 8): right
// for the training call sites (<= 6 RoIs per element, mrcnn.py:1075; lidc configs.py:258), slower than the two-kernel
// form once a batch element carries dozens of RoIs (measured: N = 600 on P2 1.4 ms vs 0.25 ms).  The launcher only
// knows N, so the switch is on N.
constexpr int BWD_TERRITORY_MAX_BOXES = 128;
// ... and one scatter workgroup per (batch element, channel) volume: with thousands of small volumes (2D Mask R-CNN:
// 20 x 192 maps of 72 x 72) the per-workgroup prologue dominates and the two-kernel form wins (N = 120, (7,7):
// 222 us vs 90 us), so the single-launch form is used up to this many volumes.
constexpr long long BWD_TERRITORY_MAX_VOLUMES = 1024;

// Which single-launch form runs first: the round-3 gather kernel (roi_align_bwd_v3.hip) unless MDT_BWD_KERNEL=r2 asks for
// the round-2 territory kernel (same-box A/B rows of tools/microbench.py).  Read once.
inline bool use_gather_kernel()
{
    static const bool v = [] { const char *e = getenv("MDT_BWD_KERNEL"); return !(e && e[0] == 'r' && e[1] == '2'); }();
    return v;
}

// the workspace query has no batch argument: it reports the two-kernel size whenever that form might be chosen
inline bool two_phase_possible(int dim, int depth) { return dim == 2 || depth > 128; }

}  // namespace

extern "C" {

int mdt_crop_and_resize_3d_forward(const float *image, const float *boxes, const int *box_ind,
                                   int num_boxes, int batch, int H, int W, int D,
                                   int ch, int cw, int cd, int depth,
                                   float extrapolation_value, float *crops, void *stream)
{
    (void)extrapolation_value;
    return launch_fwd<3, float>(image, boxes, box_ind, num_boxes, batch, H, W, D, ch, cw, cd, depth, crops,
                                (hipStream_t)stream);
}

int mdt_crop_and_resize_3d_forward_bf16(const uint16_t *image, const float *boxes, const int *box_ind,
                                        int num_boxes, int batch, int H, int W, int D,
                                        int ch, int cw, int cd, int depth, float *crops, void *stream)
{
    return launch_fwd<3, bf16raw>(reinterpret_cast<const bf16raw *>(image), boxes, box_ind, num_boxes, batch, H, W, D, ch, cw, cd, depth,
                                  crops, (hipStream_t)stream);
}

int mdt_crop_and_resize_3d_forward_u8(const uint8_t *image, const float *boxes, const int *box_ind,
                                      int num_boxes, int batch, int H, int W, int D,
                                      int ch, int cw, int cd, int depth, float *crops, void *stream)
{
    return launch_fwd<3, u8raw>(reinterpret_cast<const u8raw *>(image), boxes, box_ind, num_boxes, batch, H, W, D, ch, cw, cd, depth,
                                crops, (hipStream_t)stream);
}

int mdt_crop_and_resize_2d_forward_u8(const uint8_t *image, const float *boxes, const int *box_ind,
                                      int num_boxes, int batch, int H, int W,
                                      int ch, int cw, int depth, float *crops, void *stream)
{
    return launch_fwd<2, u8raw>(reinterpret_cast<const u8raw *>(image), boxes, box_ind, num_boxes, batch, H, W, 1, ch, cw, 1, depth,
                                crops, (hipStream_t)stream);
}

int mdt_crop_and_resize_2d_forward_bf16(const uint16_t *image, const float *boxes, const int *box_ind,
                                        int num_boxes, int batch, int H, int W,
                                        int ch, int cw, int depth, float *crops, void *stream)
{
    return launch_fwd<2, bf16raw>(reinterpret_cast<const bf16raw *>(image), boxes, box_ind, num_boxes, batch, H, W, 1, ch, cw, 1, depth,
                                  crops, (hipStream_t)stream);
}

int mdt_crop_and_resize_2d_forward(const float *image, const float *boxes, const int *box_ind,
                                   int num_boxes, int batch, int H, int W,
                                   int ch, int cw, int depth,
                                   float extrapolation_value, float *crops, void *stream)
{
    (void)extrapolation_value;
    return launch_fwd<2, float>(image, boxes, box_ind, num_boxes, batch, H, W, 1, ch, cw, 1, depth, crops,
                                (hipStream_t)stream);
}

size_t mdt_crop_and_resize_backward_workspace_bytes(int dim, int num_boxes, int depth,
                                                   int image_height, int image_width, int image_zdepth,
                                                   int crop_height, int crop_width, int crop_zdepth)
{
    if (num_boxes <= 0 || depth <= 0 || crop_height <= 0 || crop_width <= 0 || image_height <= 0 || image_width <= 0)
        return 256;
    const int d3 = dim == 3;
    if (num_boxes <= BWD_TERRITORY_MAX_BOXES && !two_phase_possible(dim, depth) &&
        bwd_territory_supported(d3 ? 3 : 2, num_boxes, 1, image_height, image_width, d3 ? image_zdepth : 1,
                                crop_height, crop_width, d3 ? crop_zdepth : 1, depth))
        return 256;   // default single-launch form needs no workspace
    return fast_layout(d3 ? 3 : 2, num_boxes, depth, image_height, image_width, d3 ? image_zdepth : 1,
                       crop_height, crop_width, d3 ? crop_zdepth : 1).total;
}

int mdt_crop_and_resize_3d_backward_ordered(const float *grads, const float *boxes, const int *box_ind,
                                            int num_boxes, int batch, int H, int W, int D,
                                            int ch, int cw, int cd, int depth,
                                            float *grads_image, void *stream)
{
    return launch_bwd<3>(grads, boxes, box_ind, num_boxes, batch, H, W, D, ch, cw, cd, depth, grads_image,
                         (hipStream_t)stream);
}

int mdt_crop_and_resize_2d_backward_ordered(const float *grads, const float *boxes, const int *box_ind,
                                            int num_boxes, int batch, int H, int W,
                                            int ch, int cw, int depth,
                                            float *grads_image, void *stream)
{
    return launch_bwd<2>(grads, boxes, box_ind, num_boxes, batch, H, W, 1, ch, cw, 1, depth, grads_image,
                         (hipStream_t)stream);
}

int mdt_crop_and_resize_3d_backward(const float *grads, const float *boxes, const int *box_ind,
                                    int num_boxes, int batch, int H, int W, int D,
                                    int ch, int cw, int cd, int depth,
                                    float *grads_image, void *workspace, size_t workspace_bytes, void *stream)
{
    const bool many_volumes = (long long)batch * depth > BWD_TERRITORY_MAX_VOLUMES && workspace != nullptr &&
        workspace_bytes >= mdt_crop_and_resize_backward_twophase_workspace_bytes(3, num_boxes, depth, H, W, D, ch, cw, cd);
    if (num_boxes <= BWD_TERRITORY_MAX_BOXES && !many_volumes) {
        if (use_gather_kernel()) {
            const int rg = launch_bwd_gather(3, 1, grads, boxes, box_ind, nullptr, num_boxes, batch, depth, &H, &W, &D, ch, cw, cd,
                                             &grads_image, (hipStream_t)stream);
            if (rg != MDT_ERR_UNSUPPORTED) return rg;
        }
        const int rt = launch_bwd_territory(3, grads, boxes, box_ind, num_boxes, batch, H, W, D, ch, cw, cd, depth,
                                            grads_image, (hipStream_t)stream);
        if (rt != MDT_ERR_UNSUPPORTED) return rt;
    }
    const int rc = launch_bwd_fast<3>(grads, boxes, box_ind, num_boxes, batch, H, W, D, ch, cw, cd, depth,
                                      grads_image, workspace, workspace_bytes, (hipStream_t)stream);
    if (rc == MDT_ERR_UNSUPPORTED)   // pool extents beyond the LDS budget: exact-order kernel handles any shape
        return launch_bwd<3>(grads, boxes, box_ind, num_boxes, batch, H, W, D, ch, cw, cd, depth, grads_image,
                             (hipStream_t)stream);
    return rc;
}

int mdt_crop_and_resize_2d_backward(const float *grads, const float *boxes, const int *box_ind,
                                    int num_boxes, int batch, int H, int W,
                                    int ch, int cw, int depth,
                                    float *grads_image, void *workspace, size_t workspace_bytes, void *stream)
{
    const bool many_volumes = (long long)batch * depth > BWD_TERRITORY_MAX_VOLUMES && workspace != nullptr &&
        workspace_bytes >= mdt_crop_and_resize_backward_twophase_workspace_bytes(2, num_boxes, depth, H, W, 1, ch, cw, 1);
    if (num_boxes <= BWD_TERRITORY_MAX_BOXES && !many_volumes) {
        if (use_gather_kernel()) {
            const int one = 1;
            const int rg = launch_bwd_gather(2, 1, grads, boxes, box_ind, nullptr, num_boxes, batch, depth, &H, &W, &one, ch, cw, 1,
                                             &grads_image, (hipStream_t)stream);
            if (rg != MDT_ERR_UNSUPPORTED) return rg;
        }
        const int rt = launch_bwd_territory(2, grads, boxes, box_ind, num_boxes, batch, H, W, 1, ch, cw, 1, depth,
                                            grads_image, (hipStream_t)stream);
        if (rt != MDT_ERR_UNSUPPORTED) return rt;
    }
    const int rc = launch_bwd_fast<2>(grads, boxes, box_ind, num_boxes, batch, H, W, 1, ch, cw, 1, depth,
                                      grads_image, workspace, workspace_bytes, (hipStream_t)stream);
    if (rc == MDT_ERR_UNSUPPORTED)
        return launch_bwd<2>(grads, boxes, box_ind, num_boxes, batch, H, W, 1, ch, cw, 1, depth, grads_image,
                             (hipStream_t)stream);
    return rc;
}

size_t mdt_crop_and_resize_backward_twophase_workspace_bytes(int dim, int num_boxes, int depth,
                                                            int image_height, int image_width, int image_zdepth,
                                                            int crop_height, int crop_width, int crop_zdepth)
{
    if (num_boxes <= 0 || depth <= 0 || crop_height <= 0 || crop_width <= 0 || image_height <= 0 || image_width <= 0)
        return 256;
    const int d3 = dim == 3;
    return fast_layout(d3 ? 3 : 2, num_boxes, depth, image_height, image_width, d3 ? image_zdepth : 1,
                       crop_height, crop_width, d3 ? crop_zdepth : 1).total;
}

int mdt_crop_and_resize_3d_backward_twophase(const float *grads, const float *boxes, const int *box_ind,
                                             int num_boxes, int batch, int H, int W, int D,
                                             int ch, int cw, int cd, int depth,
                                             float *grads_image, void *workspace, size_t workspace_bytes, void *stream)
{
    return launch_bwd_fast<3>(grads, boxes, box_ind, num_boxes, batch, H, W, D, ch, cw, cd, depth,
                              grads_image, workspace, workspace_bytes, (hipStream_t)stream);
}

int mdt_crop_and_resize_2d_backward_twophase(const float *grads, const float *boxes, const int *box_ind,
                                             int num_boxes, int batch, int H, int W,
                                             int ch, int cw, int depth,
                                             float *grads_image, void *workspace, size_t workspace_bytes, void *stream)
{
    return launch_bwd_fast<2>(grads, boxes, box_ind, num_boxes, batch, H, W, 1, ch, cw, 1, depth,
                              grads_image, workspace, workspace_bytes, (hipStream_t)stream);
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
        (void)hipGetLastError(); hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                           reinterpret_cast<v4f *>(grads_image), n4, grads_image + n4 * 4, ntail);
    }
    const long long total = (long long)num_boxes * depth * ch * cw * cd;
    if (total > 0) {
        long long blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        (void)hipGetLastError(); hipLaunchKernelGGL(crop_bwd3d_atomic_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                           grads, boxes, box_ind, total, batch, H, W, D, ch, cw, cd, depth, grads_image);
    }
    return check_launch();
}

// ---- all pyramid levels in one launch -------------------------------------------------------------------------
int mdt_pyramid_roi_align_forward(int dim, int n_levels, const void *const *images, int bf16, const int *H, const int *W,
                                  const int *D, const float *boxes, const int *batch_ix, const int *level, int num_boxes,
                                  int batch, int depth, int ch, int cw, int cd, float *crops, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dim != 2 && dim != 3) return MDT_ERR_INVALID_ARGUMENT;
    if (dim == 2) cd = 1;
    if (dim == 3) return bf16 ? launch_fwd_pyramid<3, bf16raw>(n_levels, images, H, W, D, boxes, batch_ix, level, num_boxes, batch, ch, cw, cd, depth, crops, s)
                              : launch_fwd_pyramid<3, float>(n_levels, images, H, W, D, boxes, batch_ix, level, num_boxes, batch, ch, cw, cd, depth, crops, s);
    return bf16 ? launch_fwd_pyramid<2, bf16raw>(n_levels, images, H, W, D, boxes, batch_ix, level, num_boxes, batch, ch, cw, cd, depth, crops, s)
                : launch_fwd_pyramid<2, float>(n_levels, images, H, W, D, boxes, batch_ix, level, num_boxes, batch, ch, cw, cd, depth, crops, s);
}

int mdt_pyramid_roi_align_backward(int dim, int n_levels, const float *grads, const float *boxes, const int *batch_ix,
                                   const int *level, int num_boxes, int batch, int depth, const int *H, const int *W,
                                   const int *D, int ch, int cw, int cd, float *const *grads_images, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dim != 2 && dim != 3) return MDT_ERR_INVALID_ARGUMENT;
    if (num_boxes < 0 || batch <= 0 || depth <= 0 || ch <= 0 || cw <= 0 || (dim == 3 && cd <= 0)) return MDT_ERR_INVALID_ARGUMENT;
    if (dim == 2) cd = 1;
    if (num_boxes > BWD_TERRITORY_MAX_BOXES || (long long)batch * depth > BWD_TERRITORY_MAX_VOLUMES) return MDT_ERR_UNSUPPORTED;
    if (use_gather_kernel()) {
        const int rg = launch_bwd_gather(dim, n_levels, grads, boxes, batch_ix, level, num_boxes, batch, depth, H, W, D, ch, cw, cd,
                                         grads_images, s);
        if (rg != MDT_ERR_UNSUPPORTED) return rg;
    }
    return launch_bwd_territory_multi(dim, n_levels, grads, boxes, batch_ix, level, num_boxes, batch, depth, H, W, D, ch, cw, cd,
                                      grads_images, s);
}

}  // extern "C"
