// roi_align_ab.hip -- SUPERSEDED generations of the RoIAlign backward, kept for A/B measurements and as test subjects only.
// Built into libmdt_hip_ab.so (include/mdt_hip_ab.h); NOT part of the product library libmdt_hip.so.
//   * round 1: separable two-kernel form (expand + zero-fill, then patch), needs a workspace;
//   * the reference's algorithm: zero-fill + fp32 global atomics (order-nondeterministic).
// The round-2 territory kernel is ../roi_align_bwd.hip (same library).  Reference: crop_and_resize_kernel.cu:154-304.
#include <type_traits>
#include "roi_align_ab_common.h"

using namespace mdt_ra;

namespace {

struct PRange {
    short first, last;                           // sample indices p whose [lo, hi] contains this voxel index
};

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

}  // namespace

extern "C" {

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


/* the round-2 single-launch territory kernel (roi_align_bwd.hip), one map: dim 2 or 3 */
int mdt_ab_crop_and_resize_backward_territory(int dim, const float *grads, const float *boxes, const int *box_ind, int num_boxes, int batch,
                                              int H, int W, int D, int ch, int cw, int cd, int depth, float *grads_image, void *stream)
{
    return launch_bwd_territory(dim, grads, boxes, box_ind, num_boxes, batch, H, W, dim == 3 ? D : 1, ch, cw, dim == 3 ? cd : 1, depth,
                                grads_image, (hipStream_t)stream);
}

}  // extern "C"
