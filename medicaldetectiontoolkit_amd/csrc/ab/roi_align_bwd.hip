// roi_align_bwd.hip -- default 2D/3D RoIAlign backward for gfx950: ONE launch, every byte of grads_image written
// exactly once, no atomics, no workspace, deterministic.  Arithmetic follows the reference scatter
// (cuda_functions/roi_align_3D/roi_align/src/cuda/crop_and_resize_kernel.cu:154-304; 2D:
// roi_align_2D/.../crop_and_resize_kernel.cu:102-194) with the per-axis interpolation factored out
// (dF = Wz^T (Wy^T (Wx^T g))), so sums are reassociated relative to the flat 8-corner scatter: values agree to fp32
// rounding (test bar 2e-6 * sum|terms|; north-star bar 1e-4).  The exact-order form lives in roi_align.hip.
//
// The op is a 4*B*C*V-byte fill (151 MB on the P2 level) plus a few MB of scattered gradient.  Design:
//
//   * "Territory".  The map is cut into segments of S contiguous floats (S = 8 = 32 B when the contiguous
//     extent is a multiple of 8 -- finer segments keep the territory close to the RoI footprints -- else 32 or
//     one whole row).  The territory of batch element b is the set of segments
//     inside the index bounding box of any RoI with box_ind == b: a bitmap of R*nseg bits (P2: 4096 bits) that
//     every workgroup recomputes from `boxes` in LDS (one global round trip, a few hundred lane-ops; no
//     inter-workgroup communication, no prepass launch).
//   * Role split inside the one launch.  The roles write disjoint bytes, so nothing orders them and nothing is
//     written twice:
//       zero role    as many workgroups as stay resident beside the scatter workgroups (occupancy query), each
//                    streaming 16-byte zero stores over ONE contiguous run of rows of the [B*C*R] row space and
//                    skipping territory segments -- the bitmap prologue (a global round trip) is paid once per
//                    workgroup.  Measured: the fill alone runs at the torch.zero_ rate (23 us for 151 MB)
//                    regardless of the LDS carve (2..8 workgroups per CU).
//       scatter role (B*C*ssplit workgroups, lowest block indices so they start first, raised wave priority):
//                    owns the territory of one (b, c) volume and computes it from LDS only.  The arithmetic is
//                    ~10 MFLOP per launch, so this role is pure latency; its stages are organised around
//                    "one LDS round trip = many independent loads":
//                      (a) sample tables (floor index, lerp) of the RoIs of this batch element; their gradient
//                          blocks g[r, c] start travelling global -> LDS by LDS-DMA (no staging registers) and
//                          stay in flight across the LDS-only barriers of (a)/(b);
//                      (b) one wave per (RoI, axis): the touched indices are compacted through an LDS bit mask
//                          (index -> position table; every sample learns the position of its floor index);
//                      (c) separable streaming passes (3D: x, y, z; 2D: y, x): one lane per LINE along the
//                          contracted axis fetches values and sample entries four at a time with independent
//                          loads and streams the interpolated sums to the compact positions; blocks ping-pong
//                          between two LDS regions: g[r,c] -> [py][ix][pz] -> [iy][ix][pz] -> E_r[iy][ix][iz];
//                      (d) one lane per four territory voxels adds, RoI ascending, E_r[pos] of every RoI covering
//                          them (all bounding boxes tested from one batch of 16-byte reads) and stores 16 bytes
//                          (128 B per 8 lanes).
//                    More RoIs on one batch element than fit the LDS budget: further rounds continue the running
//                    sum (same lane, same voxels: program order).
//   * Small volumes (P4/P5, 2D maps): no zero role, the scatter workgroup also zero-fills the rest of its volume.
//
// HBM-bound, no MFMA.  Algorithmic bytes per launch: 4*B*C*V (grads_image once) + 4*N*C*P (grads once) + 28*N.
#include "roi_align_ab_common.h"

using namespace mdt_ra;

namespace {

typedef unsigned long long u64;

constexpr int T_CAND = 64;         // RoIs scanned per chunk (N <= T_CAND: box table read once per workgroup)
constexpr int T_GMAX = 12;         // RoIs staged per round (upper bound)
constexpr int T_SLOTS = 8;         // gradient blocks resident in LDS at a time (pass x runs per group of T_SLOTS)
constexpr int T_HDR = 16;          // ints per staged RoI: bbq[4] (packed bounds, E offset), r, nuy, nux, nuz, aoff, boff, inv, line prefixes of passes 2 / 3
constexpr int T_LDS_MAX = 80 * 1024;   // gfx950: 160 KB per CU, two workgroups of this kernel stay resident per CU
constexpr int T_SEGLIST = 2048;         // territory segments listed in LDS (beyond: rank search per segment)

struct SEntry {                    // one sample of one axis: lerp towards the ceil index, compact position of the floor index
    float lerp;
    int plo;
};

struct TParams {
    const float *grads;
    const float *boxes;
    const int *box_ind;
    const int *level;             // optional per-RoI pyramid level; only RoIs with level[r] == level_id belong to this map
    int level_id;
    float *out;
    int N, B, C;
    int H, W, D;                   // D == 1 for 2D
    int ph, pw, pd;                // pd == 1 for 2D
    int R;                         // rows per (b, c) volume: 3D H*W, 2D H
    int L;                         // contiguous extent: 3D D, 2D W
    int S, nseg;                   // segment length (floats), segments per row
    int bw;                        // u64 words of the territory bitmap
    int G;                         // table capacity (RoIs per round)
    int P, P4;
    int a_floats, b_floats;        // LDS budgets of the two ping-pong regions
    int ssplit;                    // scatter workgroups per volume
    int parts, rows_per_part;      // zero workgroups in the grid (0: merged into the scatter role), rows of each
    int zero_per_vol;              // 0: a zero workgroup's run of rows may cross volumes (bitmaps of all batch elements in LDS);
                                   // k > 0: k zero workgroups per volume, each inside one volume (large batch: one bitmap only)
    int upr, upr_shift;            // store units per row (VEC floats each), log2 or -1
    int useg_shift;                // log2(units per segment) when nseg > 1
    int H4, W4, D4, pos_stride;    // per-RoI index->position table: y | x | z, each padded to 4 bytes
    int wy, wx, wz, mask_stride;   // per-RoI touched-index bit masks (u64 words): y | x | z
    int dbg_wg;
    int scatter_base, zero_base;   // multi-level launch: first scatter / zero block index of this level in the grid
    long long *ts;                 // tuning only: wall-clock stamps (or null)
    int dbg;                       // tuning only: bit0 zero role exits at once; bit1 per-workgroup trace; bits 4.. scatter role stops after stage k
    // LDS byte offsets (region A at 0)
    int off_b, off_tab, off_pos, off_mask, off_hdr, off_bm, off_pref, off_list, off_seg, off_misc;
};

__device__ __forceinline__ void set_bits(u64 *bm, int s, int e)   // inclusive bit range
{
    for (int w = s >> 6; w <= (e >> 6); ++w) {
        const int lo = max(s, w << 6) - (w << 6);
        const int hi = min(e, (w << 6) + 63) - (w << 6);
        const u64 upto = (hi == 63) ? ~0ULL : ((1ULL << (hi + 1)) - 1ULL);
        atomicOr(&bm[w], upto & ~((1ULL << lo) - 1ULL));
    }
}

__device__ __forceinline__ void axis_bounds(float a1, float a2, int L, int P, int &lo, int &hi)
{
    // sample coordinates are monotone in p (rounding is monotone), so the extreme indices sit at p = 0 / P-1
    const AxisEntry e0 = axis_entry(a1, a2, L, P, 0);
    const AxisEntry e1 = axis_entry(a1, a2, L, P, P - 1);
    lo = min(e0.lo, e1.lo);
    hi = max(entry_hi(e0), entry_hi(e1));
}

// Territory bitmap of batch element b (all threads of the workgroup; ends with a barrier); b < 0: the bitmaps of ALL
// batch elements, bm[bi * bw + w] (zero role: its run of rows may span two batch elements).  Returns the number of
// hits of the LAST chunk of T_CAND RoIs; when keep_list != nullptr their ids (ascending) and box coordinates stay in
// keep_list / keep_box -- with N <= T_CAND the scatter role needs no further read of box_ind / boxes.
// cand: LDS scratch, T_CAND * 8 shorts.
template <int DIM, int NT>
__device__ __forceinline__ int build_bitmap(const TParams &p, int b, u64 *bm, short *cand, int *wave_cnt,
                                            int *keep_list, float *keep_box)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwords = (b < 0) ? p.B * p.bw : p.bw;
    for (int t = tid; t < nwords; t += NT) bm[t] = 0ULL;
    int total = 0;
    for (int rb = 0; rb < p.N; rb += T_CAND) {
        bool hit = false;
        int r = 0, bi = -1, lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0, lo2 = 0, hi2 = 0;
        float bx[2 * DIM];
#pragma unroll
        for (int k = 0; k < 2 * DIM; ++k) bx[k] = 0.0f;
        u64 bal = 0;
        if (tid < T_CAND) {
            r = rb + tid;
            if (r < p.N) {     // box_ind and the box itself in one round trip (no dependent second load)
                bi = p.box_ind[r];
                if (p.level != nullptr && p.level[r] != p.level_id) bi = -1;
                const float *src = p.boxes + (long long)r * (2 * DIM);
#pragma unroll
                for (int k = 0; k < 2 * DIM; ++k) bx[k] = src[k];
            }
            hit = (r < p.N) && ((b < 0) ? (bi >= 0 && bi < p.B) : (bi == b));
            if (hit) {
                axis_bounds(bx[0], bx[2], p.H, p.ph, lo0, hi0);
                axis_bounds(bx[1], bx[3], p.W, p.pw, lo1, hi1);
                if (DIM == 3) axis_bounds(bx[4], bx[5], p.D, p.pd, lo2, hi2);
            }
            bal = __ballot(hit);
            if (lane == 0) wave_cnt[wave] = __popcll(bal);
        }
        __syncthreads();            // also orders the bitmap zeroing / the previous chunk's readers of cand
        total = 0;
        int off = 0;
#pragma unroll
        for (int w = 0; w < T_CAND / 64; ++w) {
            const int cnt = wave_cnt[w];
            if (w < wave) off += cnt;
            total += cnt;
        }
        if (hit) {
            const int slot = off + __popcll(bal & ((1ULL << lane) - 1ULL));
            short *c = cand + slot * 8;
            c[0] = (short)lo0; c[1] = (short)hi0; c[2] = (short)lo1; c[3] = (short)hi1;
            // segment range along the contiguous axis
            if (DIM == 3) { c[4] = (short)(lo2 / p.S); c[5] = (short)(hi2 / p.S); }
            else { c[4] = (short)(lo1 / p.S); c[5] = (short)(hi1 / p.S); }
            c[6] = (short)((b < 0) ? bi : 0);
            if (keep_list) {
                keep_list[slot] = r;
#pragma unroll
                for (int k = 0; k < 2 * DIM; ++k) keep_box[slot * 6 + k] = bx[k];
            }
        }
        __syncthreads();
        for (int t = tid; t < total * p.H; t += NT) {
            const int e = t / p.H;
            const int y = t - e * p.H;
            const short *c = cand + e * 8;
            if (y < c[0] || y > c[1]) continue;
            u64 *bmb = bm + c[6] * p.bw;
            if (DIM == 3) {
                if (c[4] == 0 && c[5] == p.nseg - 1) {
                    set_bits(bmb, (y * p.W + c[2]) * p.nseg, (y * p.W + c[3]) * p.nseg + p.nseg - 1);
                } else {
                    for (int x = c[2]; x <= c[3]; ++x) set_bits(bmb, (y * p.W + x) * p.nseg + c[4], (y * p.W + x) * p.nseg + c[5]);
                }
            } else {
                set_bits(bmb, y * p.nseg + c[4], y * p.nseg + c[5]);
            }
        }
        __syncthreads();
    }
    if (p.N == 0) __syncthreads();   // bitmap zeroing visible
    return total;
}

// zero stores over rows [r0, r1) of volume `vol`, skipping territory segments
template <int VEC>
__device__ __forceinline__ void zero_rows(const TParams &p, const u64 *bm, int vol, int r0, int r1, int start, int stride)
{
    const int nu = (r1 - r0) * p.upr;
    float *base = p.out + ((long long)vol * p.R + r0) * p.L;
    const v4f z4 = {0.f, 0.f, 0.f, 0.f};
    for (int u = start; u < nu; u += stride) {
        int rl, ui;
        if (p.upr_shift >= 0) { rl = u >> p.upr_shift; ui = u & (p.upr - 1); }
        else { rl = u / p.upr; ui = u - rl * p.upr; }
        const int bit = (r0 + rl) * p.nseg + ((p.nseg > 1) ? (ui >> p.useg_shift) : 0);
        if ((bm[bit >> 6] >> (bit & 63)) & 1ULL) continue;
        if (VEC == 4) reinterpret_cast<v4f *>(base)[u] = z4;
        else base[u] = 0.0f;
    }
}

// n-th (0-based) set bit of w; w has more than n bits set
__device__ __forceinline__ int nth_set_bit(u64 w, int n)
{
    int pos = 0;
#pragma unroll
    for (int width = 32; width >= 1; width >>= 1) {
        const u64 lowmask = (1ULL << width) - 1ULL;
        const int c = __popcll((w >> pos) & lowmask);
        if (n >= c) { n -= c; pos += width; }
    }
    return pos;
}

// workgroup barrier that waits for this wave's LDS traffic only: global loads issued earlier (the gradient-block
// prefetch) stay in flight across it -- __syncthreads() would drain them (s_waitcnt vmcnt(0))
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

#define TSTAMP(k) do { if (p.ts && !zero_role && bid == (unsigned)p.dbg_wg && threadIdx.x == 0) { p.ts[k] = (long long)wall_clock64(); p.ts[16 + k] = (long long)clock64(); } } while (0)

// One streaming pass along one line: n samples in[q * istride] (q ascending, or descending for an inverted box, so
// that the compact positions ascend), sample q adding (1 - lerp) * v to position plo(q) and lerp * v to plo(q) + 1.
// Values and sample entries are fetched four at a time with independent loads (one LDS round trip), the running
// sums of the two open positions stay in registers and every position is stored exactly once.
__device__ __forceinline__ void stream_line(const float *in, int istride, float *out, int ostride,
                                            const SEntry *S, int n, int nu, bool inv)
{
    int p0 = -1;
    float acc0 = 0.0f, acc1 = 0.0f;
    for (int q0 = 0; q0 < n; q0 += 4) {
        float v[4];
        SEntry e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = q0 + k;
            const int qq = (q < n) ? (inv ? n - 1 - q : q) : 0;
            v[k] = in[qq * istride];
            e[k] = S[qq];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (q0 + k < n) {
                const int plo = e[k].plo;
                if (p0 < 0) p0 = plo;
                while (p0 < plo) { out[p0 * ostride] = acc0; acc0 = acc1; acc1 = 0.0f; ++p0; }
                acc0 = acc0 + (1.0f - e[k].lerp) * v[k];
                if (e[k].lerp > 0.0f) acc1 = acc1 + e[k].lerp * v[k];
            }
        }
    }
    out[p0 * ostride] = acc0;
    if (p0 + 1 < nu) out[(p0 + 1) * ostride] = acc1;
}

// SL > 0: segments of S = 4 * SL floats and 16-byte stores -> one lane per four consecutive voxels in stage (d), SL lanes
// per segment; SL == 0: scalar stage (d), 32 lanes per segment (any S)
template <int DIM, int VEC, int SL, int NT>
__device__ __forceinline__ void territory_body(const TParams &p, const bool zero_role, const unsigned bid)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *RA = reinterpret_cast<float *>(smem_raw);                              // [a_floats]  (bitmap build: cand scratch; zero role: all bitmaps)
    float *RB = reinterpret_cast<float *>(smem_raw + p.off_b);                    // [b_floats]
    SEntry *tab = reinterpret_cast<SEntry *>(smem_raw + p.off_tab);               // [G][psum]
    unsigned char *pos = reinterpret_cast<unsigned char *>(smem_raw + p.off_pos); // [G][pos_stride]  255 = untouched
    u64 *mask = reinterpret_cast<u64 *>(smem_raw + p.off_mask);                   // [G][mask_stride]
    int *hdr = reinterpret_cast<int *>(smem_raw + p.off_hdr);                     // [G][T_HDR]
    u64 *bm = reinterpret_cast<u64 *>(smem_raw + p.off_bm);                       // [bw]
    int *pref = reinterpret_cast<int *>(smem_raw + p.off_pref);                   // [bw + 1]
    int *list = reinterpret_cast<int *>(smem_raw + p.off_list);                   // [T_CAND]
    float *lbox = reinterpret_cast<float *>(list + T_CAND);                       // [T_CAND][6] boxes of the listed RoIs
    unsigned short *seglist = reinterpret_cast<unsigned short *>(smem_raw + p.off_seg);   // [T_SEGLIST] bit index of the s-th territory segment
    int *wave_cnt = reinterpret_cast<int *>(smem_raw + p.off_misc);               // [T_CAND / 64]
    int *misc = wave_cnt + 4;                                                     // [0] RoIs that fit this round, [1] / [2] lines of passes 2 / 3
    short *cand = reinterpret_cast<short *>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    struct WgTrace {     // tuning only (dbg bit 1): start / end wall clock and placement of every workgroup
        long long *slot;
        __device__ WgTrace(const TParams &p) : slot(nullptr) {
            if (p.ts && (p.dbg & 2) && threadIdx.x == 0) {
                slot = p.ts + 64 + 4 * (long long)blockIdx.x;
                slot[0] = (long long)wall_clock64();
                unsigned hw, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                slot[2] = hw; slot[3] = xcc;
            }
        }
        __device__ ~WgTrace() { if (slot) slot[1] = (long long)wall_clock64(); }
    } wg_trace(p);

    if (zero_role) {
        // ------------------------------------------------------------- zero role
        // one contiguous run of rows of the whole [B*C*R] row space per workgroup; the grid holds as many zero
        // workgroups as stay resident next to the scatter workgroups, so each pays the bitmap prologue once
        const long long zi = (long long)bid;
        u64 *bm_all = reinterpret_cast<u64 *>(smem_raw + T_CAND * 8 * sizeof(short));   // regions A/B are unused here
        if (p.zero_per_vol > 0) {
            const int vol = (int)(zi / p.zero_per_vol);
            const int r0 = (int)(zi - (long long)vol * p.zero_per_vol) * p.rows_per_part;
            const int r1 = min(p.R, r0 + p.rows_per_part);
            if (r0 >= r1 || (p.dbg & 1)) return;
            build_bitmap<DIM, NT>(p, vol / p.C, bm_all, cand, wave_cnt, nullptr, nullptr);
            zero_rows<VEC>(p, bm_all, vol, r0, r1, tid, NT);
            return;
        }
        const long long g0 = zi * p.rows_per_part;
        const long long total_rows = (long long)p.B * p.C * p.R;
        long long g1 = g0 + p.rows_per_part;
        if (g1 > total_rows) g1 = total_rows;
        if (g0 >= g1 || (p.dbg & 1)) return;
        TSTAMP(0);
        build_bitmap<DIM, NT>(p, -1, bm_all, cand, wave_cnt, nullptr, nullptr);
        TSTAMP(1);
        for (long long g = g0; g < g1;) {
            const int vol = (int)(g / p.R);
            const int r0 = (int)(g - (long long)vol * p.R);
            const int r1 = (int)min((long long)p.R, r0 + (g1 - g));
            zero_rows<VEC>(p, bm_all + (vol / p.C) * p.bw, vol, r0, r1, tid, NT);
            g += r1 - r0;
        }
        TSTAMP(2);
        return;
    }

    // ----------------------------------------------------------------- scatter role
    __builtin_amdgcn_s_setprio(3);      // latency-critical: win instruction issue against the streaming zero-role waves
    const int vol = bid / p.ssplit;
    const int split = bid - vol * p.ssplit;
    const int b = vol / p.C;
    const int c = vol - b * p.C;
    const int dbg_stop = p.dbg >> 4;
    TSTAMP(0);
    const int cnt_last = build_bitmap<DIM, NT>(p, b, bm, cand, wave_cnt, list, lbox);
    TSTAMP(1);
    if (dbg_stop == 1) return;
    if (p.parts == 0)   // merged mode: this workgroup also zero-fills the non-territory part of its volume
        zero_rows<VEC>(p, bm, vol, 0, p.R, split * NT + tid, p.ssplit * NT);
    // prefix popcounts of the bitmap -> number of territory segments
    if (tid < 64) {
        int run = 0;
        for (int w0 = 0; w0 < p.bw; w0 += 64) {
            const int w = w0 + tid;
            const int cnt = (w < p.bw) ? __popcll(bm[w]) : 0;
            int incl = cnt;   // wave inclusive scan
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d);
                if (tid >= d) incl += v;
            }
            if (w < p.bw) pref[w] = run + incl - cnt;
            run += __shfl(incl, 63);
        }
        if (tid == 0) pref[p.bw] = run;
    }
    __syncthreads();
    const int nterr = pref[p.bw];
    if (nterr == 0) return;
    // bit index of the s-th territory segment, one thread per bitmap word
    for (int w = tid; w < p.bw; w += NT) {
        u64 m = bm[w];
        int k = pref[w];
        while (m && k < T_SEGLIST) {
            const int bpos = __ffsll((long long)m) - 1;
            seglist[k++] = (unsigned short)((w << 6) + bpos);
            m &= m - 1;
        }
    }
    __syncthreads();
    TSTAMP(2);

    const int psum = p.ph + p.pw + p.pd;
    const int P = p.P;
    float *ovol = p.out + (long long)vol * p.R * p.L;
    constexpr bool QUAD = SL > 0;
    constexpr int SLOT_LANES = QUAD ? SL : 32;           // lanes per territory segment in stage (d)
    constexpr int SLOTS = NT / SLOT_LANES;
    constexpr int WSLOTS = 64 / SLOT_LANES;              // segments per wave and iteration
    const int slot = tid / SLOT_LANES, sl = tid % SLOT_LANES;
    // word prefix of the territory bitmap in registers (rank -> word search by ballot, no LDS round trips)
    const bool small_bm = p.bw <= 256;
    int prefreg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) prefreg[k] = (small_bm && lane + 64 * k < p.bw) ? pref[lane + 64 * k] : 0x7fffffff;
    int round = 0;

    for (int rb = 0; rb < p.N; rb += T_CAND) {
        // ordered list of this batch element's RoIs within the chunk (+ their boxes)
        int cnt;
        if (p.N <= T_CAND) {
            cnt = cnt_last;              // left in list / lbox by build_bitmap
        } else {
            bool hit = false;
            int rr = 0;
            float bx[2 * DIM];
#pragma unroll
            for (int k = 0; k < 2 * DIM; ++k) bx[k] = 0.0f;
            u64 bal = 0;
            __syncthreads();             // wave_cnt / list of the previous chunk are no longer read
            if (tid < T_CAND) {
                rr = rb + tid;
                int bi = -1;
                if (rr < p.N) {
                    bi = p.box_ind[rr];
                    if (p.level != nullptr && p.level[rr] != p.level_id) bi = -1;
                    const float *src = p.boxes + (long long)rr * (2 * DIM);
#pragma unroll
                    for (int k = 0; k < 2 * DIM; ++k) bx[k] = src[k];
                }
                hit = (rr < p.N) && (bi == b);
                bal = __ballot(hit);
                if (lane == 0) wave_cnt[wave] = __popcll(bal);
            }
            __syncthreads();
            int off = 0;
            cnt = 0;
#pragma unroll
            for (int w = 0; w < T_CAND / 64; ++w) {
                const int k = wave_cnt[w];
                if (w < wave) off += k;
                cnt += k;
            }
            if (hit) {
                const int sl_ = off + __popcll(bal & ((1ULL << lane) - 1ULL));
                list[sl_] = rr;
#pragma unroll
                for (int k = 0; k < 2 * DIM; ++k) lbox[sl_ * 6 + k] = bx[k];
            }
            __syncthreads();
        }

        int g0 = 0;
        while (g0 < cnt) {
            int ng = min(p.G, cnt - g0);
            if (round > 0) lds_barrier();          // previous round's readers of the tables / blocks are done
            // gradient blocks of RoIs [j0, j1) of the round: global -> LDS (region A, slot j - j0) by LDS-DMA, no staging
            // registers.  The first T_SLOTS are issued now and stay in flight across the LDS-only barriers of (a)/(b)
            auto issue_dma = [&](int j0, int j1) {
                int rj[T_SLOTS];
#pragma unroll
                for (int j = 0; j < T_SLOTS; ++j) rj[j] = (j0 + j < j1) ? list[g0 + j0 + j] : 0;
                const bool x4 = (P & 3) == 0;                       // 16-byte DMA when the blocks are 16-byte aligned
                const int per = x4 ? 256 : 64;                      // floats per wave instruction
                const int cpb = (P + per - 1) / per;                // chunks per block
                for (int ci = wave; ci < (j1 - j0) * cpb; ci += NT / 64) {
                    const int j = ci / cpb;
                    const int k = ci - j * cpb;
                    int r = rj[0];
#pragma unroll
                    for (int jj = 1; jj < T_SLOTS; ++jj) if (j == jj) r = rj[jj];
                    const float *blk = p.grads + ((long long)r * p.C + c) * P;
                    float *dstw = RA + j * p.P4 + k * per;          // wave-uniform; lane i lands at dstw + i * (4 | 16 bytes)
                    if (x4) {
                        const int t = k * 256 + lane * 4;
                        if (t < P)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(blk + t),
                                                             (__attribute__((address_space(3))) void *)dstw, 16, 0, 0);
                    } else {
                        const int t = k * 64 + lane;
                        if (t < P)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(blk + t),
                                                             (__attribute__((address_space(3))) void *)dstw, 4, 0, 0);
                    }
                }
            };
            issue_dma(0, min(ng, T_SLOTS));
            // (a) sample tables of the staged RoIs (plo = voxel index for now); masks / flags cleared
            for (int t = tid; t < ng * psum; t += NT) {
                const int j = t / psum;
                const int q = t - j * psum;
                const float *bx = lbox + (g0 + j) * 6;
                AxisEntry e;
                if (q < p.ph) e = axis_entry(bx[0], bx[2], p.H, p.ph, q);
                else if (q < p.ph + p.pw) e = axis_entry(bx[1], bx[3], p.W, p.pw, q - p.ph);
                else {
                    if (DIM == 3) e = axis_entry(bx[4], bx[5], p.D, p.pd, q - p.ph - p.pw);
                    else { e.lo = 0; e.lerp = 0.0f; }
                }
                SEntry se;
                se.lerp = e.lerp;
                se.plo = e.lo;
                tab[j * psum + q] = se;
            }
            for (int t = tid; t < ng * p.mask_stride; t += NT) mask[t] = 0ULL;
            if (tid < ng) { hdr[tid * T_HDR + 10] = 0; hdr[tid * T_HDR + 4] = list[g0 + tid]; if (DIM == 2) { hdr[tid * T_HDR + 2] = 0; hdr[tid * T_HDR + 7] = 1; } }
            lds_barrier();
            TSTAMP(3);
            // (b) one wave per (RoI, axis)
            for (int pr = wave; pr < ng * DIM; pr += NT / 64) {
                const int j = pr / DIM;
                const int a = pr - j * DIM;
                SEntry *ta = tab + j * psum + (a == 0 ? 0 : a == 1 ? p.ph : p.ph + p.pw);
                const int np_ = (a == 0) ? p.ph : (a == 1) ? p.pw : p.pd;
                const int La = (a == 0) ? p.H : (a == 1) ? p.W : p.D;
                int *h = hdr + j * T_HDR;
                int mylo = 0, myhi = 0;
                float mylerp = 0.0f;
                if (lane < np_) { const SEntry e = ta[lane]; mylo = e.plo; mylerp = e.lerp; myhi = mylo + (mylerp > 0.0f ? 1 : 0); }
                // bounding indices and direction of this axis from the first / last sample (monotone coordinates)
                const int l0 = __builtin_amdgcn_readlane(mylo, 0), l1 = __builtin_amdgcn_readlane(mylo, np_ - 1);
                const int h0 = __builtin_amdgcn_readlane(myhi, 0), h1 = __builtin_amdgcn_readlane(myhi, np_ - 1);
                const int ilo = min(l0, l1), ihi = max(h0, h1);
                if (lane == 0) {
                    h[a] = ilo | (ihi << 16);
                    const bool inv = (l1 < l0) || (l1 == l0 && __builtin_amdgcn_readlane(mylerp, np_ - 1) < __builtin_amdgcn_readlane(mylerp, 0));
                    if (inv) atomicOr(&h[10], 1 << a);
                }
                // touched-index bit mask (LDS atomics of this wave, read back in order)
                u64 *mj = mask + j * p.mask_stride + (a == 0 ? 0 : a == 1 ? p.wy : p.wy + p.wx);
                const int nw = (a == 0) ? p.wy : (a == 1) ? p.wx : p.wz;
                if (lane < np_) {
                    atomicOr(&mj[mylo >> 6], 1ULL << (mylo & 63));
                    atomicOr(&mj[myhi >> 6], 1ULL << (myhi & 63));
                }
                unsigned char *pj = pos + j * p.pos_stride + (a == 0 ? 0 : a == 1 ? p.H4 : p.H4 + p.W4);
                int run = 0, myplo = 0;
                for (int w = 0; w < nw; ++w) {
                    const u64 m = mj[w];
                    const int idx = (w << 6) + lane;
                    if (idx < La) pj[idx] = ((m >> lane) & 1ULL) ? (unsigned char)(run + __popcll(m & ((1ULL << lane) - 1ULL))) : (unsigned char)255;
                    if (lane < np_ && (mylo >> 6) == w) myplo = run + __popcll(m & ((1ULL << (mylo & 63)) - 1ULL));
                    run += __popcll(m);
                }
                if (lane < np_) { SEntry e; e.lerp = mylerp; e.plo = myplo; ta[lane] = e; }
                if (lane == 0) h[5 + a] = run;      // [5] = nuy, [6] = nux, [7] = nuz
            }
            lds_barrier();
            TSTAMP(4);
            // block offsets in the two regions; RoIs beyond the LDS budgets wait for the next round
            if (tid < 64) {
                int asz = 0, bsz = 0, l2 = 0, l3 = 0;
                if (tid < ng) {
                    const int *h = hdr + tid * T_HDR;
                    const int nuy = h[5], nux = h[6], nuz = h[7];
                    if (DIM == 3) { asz = nuy * nux * p.pd; bsz = max(p.ph * nux * p.pd, nuy * nux * nuz); l2 = nux * p.pd; l3 = nuy * nux; }
                    else { asz = nuy * nux; bsz = nuy * p.pw; l2 = nuy; }
                }
                int ai = asz, bi = bsz;
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) {
                    const int va = __shfl_up(ai, d), vb = __shfl_up(bi, d);
                    if (tid >= d) { ai += va; bi += vb; }
                }
                const bool fits = (tid < ng) && (ai <= p.a_floats) && (bi <= p.b_floats);
                const u64 fb = __ballot(fits);
                if (!fits) { l2 = 0; l3 = 0; }
                int l2i = l2, l3i = l3;
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) {
                    const int v2 = __shfl_up(l2i, d), v3 = __shfl_up(l3i, d);
                    if (tid >= d) { l2i += v2; l3i += v3; }
                }
                if (tid < ng) {
                    int *h = hdr + tid * T_HDR;
                    h[8] = ai - asz; h[9] = bi - bsz; h[11] = l2i - l2; h[12] = l3i - l3;
                    h[3] = (DIM == 3) ? (bi - bsz) : (ai - asz);        // offset of the block stage (d) reads
                }
                const int nfit = __popcll(fb);            // prefix sums are monotone: the fitting RoIs are a prefix
                if (tid == 0) misc[0] = nfit;
                if (nfit > 0 && tid == nfit - 1) { misc[1] = l2i; misc[2] = l3i; }
            }
            __syncthreads();                              // also awaits the gradient blocks (vmcnt(0))
            ng = misc[0];
            TSTAMP(5);
            if (dbg_stop == 2) return;

            // (c) streaming passes
            if (DIM == 3) {
                // pass x: lines (j, py, pz): A[py][px][pz] -> B[py][ix][pz], T_SLOTS RoIs at a time
                for (int jg = 0; jg < ng; jg += T_SLOTS) {
                    const int nj = min(T_SLOTS, ng - jg);
                    if (jg > 0) {          // later groups: their blocks are fetched now (one exposed round trip)
                        __syncthreads();
                        issue_dma(jg, jg + nj);
                        __syncthreads();
                    }
                    const int lpr = p.ph * p.pd;                    // lines per RoI
                    for (int t = tid; t < nj * lpr; t += NT) {
                        const int js = t / lpr;
                        const int j = jg + js;
                        const int l = t - js * lpr;
                        const int py = l / p.pd, pz = l - py * p.pd;
                        const int *h = hdr + j * T_HDR;
                        const int nux = h[6];
                        stream_line(RA + js * p.P4 + py * p.pw * p.pd + pz, p.pd, RB + h[9] + py * nux * p.pd + pz, p.pd,
                                    tab + j * psum + p.ph, p.pw, nux, (h[10] & 2) != 0);
                    }
                }
                __syncthreads();
                {   // pass y: lines (j, ix, pz): B[py][ix][pz] -> A[iy][ix][pz]
                    const int total = misc[1];
                    for (int t = tid; t < total; t += NT) {
                        int j = 0;
#pragma unroll
                        for (int jj = 1; jj < T_GMAX; ++jj) if (jj < ng && t >= hdr[jj * T_HDR + 11]) j = jj;
                        const int *h = hdr + j * T_HDR;
                        const int lines = h[6] * p.pd;
                        const int l = t - h[11];
                        stream_line(RB + h[9] + l, lines, RA + h[8] + l, lines, tab + j * psum, p.ph, h[5], (h[10] & 1) != 0);
                    }
                }
                __syncthreads();
                {   // pass z: lines (j, iy, ix): A[iy][ix][pz] -> B[iy][ix][iz]
                    const int total = misc[2];
                    for (int t = tid; t < total; t += NT) {
                        int j = 0;
#pragma unroll
                        for (int jj = 1; jj < T_GMAX; ++jj) if (jj < ng && t >= hdr[jj * T_HDR + 12]) j = jj;
                        const int *h = hdr + j * T_HDR;
                        const int nuz = h[7];
                        const int l = t - h[12];
                        stream_line(RA + h[8] + l * p.pd, 1, RB + h[9] + l * nuz, 1, tab + j * psum + p.ph + p.pw, p.pd, nuz, (h[10] & 4) != 0);
                    }
                }
            } else {
                // pass y: lines (j, px): A[py][px] -> B[iy][px], T_SLOTS RoIs at a time
                for (int jg = 0; jg < ng; jg += T_SLOTS) {
                    const int nj = min(T_SLOTS, ng - jg);
                    if (jg > 0) {
                        __syncthreads();
                        issue_dma(jg, jg + nj);
                        __syncthreads();
                    }
                    for (int t = tid; t < nj * p.pw; t += NT) {
                        const int js = t / p.pw;
                        const int j = jg + js;
                        const int l = t - js * p.pw;
                        const int *h = hdr + j * T_HDR;
                        stream_line(RA + js * p.P4 + l, p.pw, RB + h[9] + l, p.pw, tab + j * psum, p.ph, h[5], (h[10] & 1) != 0);
                    }
                }
                __syncthreads();
                {   // pass x: lines (j, iy): B[iy][px] -> A[iy][ix]
                    const int total = misc[1];
                    for (int t = tid; t < total; t += NT) {
                        int j = 0;
#pragma unroll
                        for (int jj = 1; jj < T_GMAX; ++jj) if (jj < ng && t >= hdr[jj * T_HDR + 11]) j = jj;
                        const int *h = hdr + j * T_HDR;
                        const int l = t - h[11];
                        stream_line(RB + h[9] + l * p.pw, 1, RA + h[8] + l * h[6], 1, tab + j * psum + p.ph, p.pw, h[6], (h[10] & 2) != 0);
                    }
                }
            }
            __syncthreads();
            const float *E = (DIM == 3) ? RB : RA;        // E_j[iy][ix][iz] (3D) / [iy][ix] (2D)
            TSTAMP(6);
            if (dbg_stop == 3) return;

            // (d) territory voxels: sum of the covering RoIs' compact-block entries, RoI ascending
            for (int s0 = split * SLOTS; s0 < nterr; s0 += SLOTS * p.ssplit) {
                const int s = s0 + slot;
                const bool valid = s < nterr;
                int bit;
                if (s0 + SLOTS <= T_SEGLIST) {          // uniform: the whole iteration is inside the list
                    if (!valid) continue;
                    bit = seglist[s];
                } else {
                    int w;
                    if (small_bm) {        // word holding the s-th set bit: popcount of (word prefix <= s) over the wave
                        w = 0;
#pragma unroll
                        for (int k = 0; k < WSLOTS; ++k) {
                            const int tgt = __builtin_amdgcn_readlane(s, k * SLOT_LANES);
                            const int wk = __popcll(__ballot(prefreg[0] <= tgt)) + __popcll(__ballot(prefreg[1] <= tgt)) +
                                           __popcll(__ballot(prefreg[2] <= tgt)) + __popcll(__ballot(prefreg[3] <= tgt)) - 1;
                            if ((lane / SLOT_LANES) == k) w = wk;
                        }
                    } else {
                        int wlo = 0, whi = p.bw - 1;
                        while (wlo < whi) {
                            const int mid = (wlo + whi + 1) >> 1;
                            if (pref[mid] <= s) wlo = mid; else whi = mid - 1;
                        }
                        w = wlo;
                    }
                    if (!valid) continue;
                    bit = (w << 6) + nth_set_bit(bm[w], s - pref[w]);
                }
                if (s0 == split * SLOTS) TSTAMP(8);
                const int row = bit / p.nseg;
                const int seg = bit - row * p.nseg;
                int y, x = 0;
                if (DIM == 3) { y = row / p.W; x = row - y * p.W; } else { y = row; }
                if (s0 == split * SLOTS) TSTAMP(9);
                if (QUAD) {
                    const int ci0 = seg * (4 * SL) + sl * 4;   // four consecutive indices along the contiguous axis
                    v4f *dst = reinterpret_cast<v4f *>(ovol + (long long)row * p.L + ci0);
                    v4f acc = {0.f, 0.f, 0.f, 0.f};
                    bool touched = false;
                    for (int j0 = 0; j0 < ng; j0 += 4) {
                    // bounding boxes of four staged RoIs from one batch of 16-byte reads
                    int4 bbq[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) bbq[k] = *reinterpret_cast<const int4 *>(hdr + ((j0 + k < ng) ? j0 + k : 0) * T_HDR);
#pragma unroll
                    for (int jk = 0; jk < 4; ++jk) {
                        const int j = j0 + jk;
                        if (j >= ng) break;
                        const int4 bb = bbq[jk];
                        if (y < (bb.x & 0xffff) || y > (bb.x >> 16)) continue;
                        const int *h = hdr + j * T_HDR;
                        const unsigned char *pj = pos + j * p.pos_stride;
                        int base;
                        unsigned int pq;        // four packed positions along the contiguous axis
                        if (DIM == 3) {
                            if (x < (bb.y & 0xffff) || x > (bb.y >> 16) || ci0 + 3 < (bb.z & 0xffff) || ci0 > (bb.z >> 16)) continue;
                            const int piy = pj[y], pix = pj[p.H4 + x];
                            pq = *reinterpret_cast<const unsigned int *>(pj + p.H4 + p.W4 + ci0);
                            if (piy == 255 || pix == 255) continue;
                            base = (piy * h[6] + pix) * h[7];
                        } else {
                            if (ci0 + 3 < (bb.y & 0xffff) || ci0 > (bb.y >> 16)) continue;
                            const int piy = pj[y];
                            pq = *reinterpret_cast<const unsigned int *>(pj + p.H4 + ci0);
                            if (piy == 255) continue;
                            base = piy * h[6];
                        }
                        if (pq == 0xffffffffu) continue;
                        if (!touched) { touched = true; if (round > 0) acc = *dst; }
                        const float *Ej = E + bb.w + base;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned int pk = (pq >> (8 * k)) & 255u;
                            if (pk != 255u) acc[k] = acc[k] + Ej[pk];
                        }
                    }
                    }
                    if (round == 0 || touched) *dst = acc;
                    if (s0 == split * SLOTS) TSTAMP(10);
                } else {
                    for (int ci = seg * p.S + sl; ci < (seg + 1) * p.S; ci += SLOT_LANES) {
                        float *dst = ovol + (long long)row * p.L + ci;
                        float acc = 0.0f;
                        bool touched = false;
                        for (int j = 0; j < ng; ++j) {
                            const int4 bb = *reinterpret_cast<const int4 *>(hdr + j * T_HDR);
                            if (y < (bb.x & 0xffff) || y > (bb.x >> 16)) continue;
                            const int *h = hdr + j * T_HDR;
                            const unsigned char *pj = pos + j * p.pos_stride;
                            int e;
                            if (DIM == 3) {
                                if (x < (bb.y & 0xffff) || x > (bb.y >> 16) || ci < (bb.z & 0xffff) || ci > (bb.z >> 16)) continue;
                                const int piy = pj[y], pix = pj[p.H4 + x], piz = pj[p.H4 + p.W4 + ci];
                                if (piy == 255 || pix == 255 || piz == 255) continue;
                                e = (piy * h[6] + pix) * h[7] + piz;
                            } else {
                                if (ci < (bb.y & 0xffff) || ci > (bb.y >> 16)) continue;
                                const int piy = pj[y], pix = pj[p.H4 + ci];
                                if (piy == 255 || pix == 255) continue;
                                e = piy * h[6] + pix;
                            }
                            if (!touched) { touched = true; if (round > 0) acc = *dst; }
                            acc = acc + E[bb.w + e];
                        }
                        if (round == 0 || touched) *dst = acc;
                    }
                }
            }
            TSTAMP(7);
            g0 += ng;
            ++round;
        }
    }
}

template <int DIM, int VEC, int SL, int NT>
__global__ __launch_bounds__(NT, NT / 128) void crop_bwd_territory_kernel(TParams p)
{
    const unsigned n_scatter = (unsigned)(p.B * p.C * p.ssplit);
    const bool zero_role = blockIdx.x >= n_scatter;
    territory_body<DIM, VEC, SL, NT>(p, zero_role, zero_role ? blockIdx.x - n_scatter : blockIdx.x);
}

// Multi-level form: one launch for all pyramid levels (mrcnn.py:373-457 routes every RoI to one level).  Block order:
// level 0 scatter, level 0 zero, level 1 scatter, ... -- the finest level (the 151 MB fill) starts first.
constexpr int T_MAX_LEVELS = 5;
struct TMulti {
    TParams lev[T_MAX_LEVELS];
    int n_levels;
    unsigned end[2 * T_MAX_LEVELS];         // exclusive block-index ends of: level 0 scatter, level 0 zero, level 1 scatter, ...
};

template <int DIM, int VEC, int SL, int NT>
__global__ __launch_bounds__(NT, NT / 128) void crop_bwd_territory_multi_kernel(TMulti m)
{
    const unsigned b = blockIdx.x;
    int r = 0;
    unsigned base = 0u;
    for (int k = 0; k < 2 * m.n_levels - 1; ++k)
        if (b >= m.end[k]) { r = k + 1; base = m.end[k]; }
    territory_body<DIM, VEC, SL, NT>(m.lev[r >> 1], (r & 1) != 0, b - base);
}

inline int ilog2_exact(int v)
{
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

// Launch-geometry knobs (MDT_BWD_*) exist for the tuning scripts under tools/; they are honoured only when
// MDT_BWD_TUNE is set in the environment, so a normal call never touches getenv beyond the first one.
inline int env_int(const char *name, int dflt)
{
    static const bool tune = getenv("MDT_BWD_TUNE") != nullptr;
    if (!tune) return dflt;
    const char *v = getenv(name);
    return (v && v[0]) ? atoi(v) : dflt;
}

inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// Fills the shape-dependent part of TParams; returns MDT_ERR_UNSUPPORTED when the LDS budgets do not fit.
int territory_plan(int dim, int N, int B, int H, int W, int D, int ph, int pw, int pd, int C, int vec,
                   TParams &p, size_t &lds)
{
    if (N < 0 || B <= 0 || H <= 0 || W <= 0 || D <= 0 || ph <= 0 || pw <= 0 || pd <= 0 || C <= 0)
        return MDT_ERR_INVALID_ARGUMENT;
    // pool extents: one wave lane per sample in stage (b)
    if (H > 32000 || W > 32000 || D > 32000 || ph > 64 || pw > 64 || pd > 64) return MDT_ERR_UNSUPPORTED;
    const long long vol_floats = (long long)H * W * D;
    if (vol_floats > 0x3fffffffLL || (long long)B * C > 0x3fffffLL) return MDT_ERR_UNSUPPORTED;
    p.N = N; p.B = B; p.C = C; p.H = H; p.W = W; p.D = D; p.ph = ph; p.pw = pw; p.pd = pd;
    p.R = (dim == 3) ? H * W : H;
    p.L = (dim == 3) ? D : W;
    {
        int seg = env_int("MDT_BWD_SEG", 8);
        if (seg != 32 && seg != 16 && seg != 8) seg = 8;
        while (seg < 32 && ((long long)p.R * (p.L / seg) > 16384)) seg *= 2;    // keep the bitmap small
        p.S = (vec == 4 && p.L % seg == 0) ? seg : ((p.L % 32 == 0) ? 32 : p.L);
    }
    p.nseg = p.L / p.S;
    const long long nbits = (long long)p.R * p.nseg;
    if (nbits > 65536) return MDT_ERR_UNSUPPORTED;
    p.bw = (int)((nbits + 63) / 64);
    p.P = ph * pw * pd;
    p.P4 = (p.P + 3) & ~3;
    p.upr = p.L / vec;
    p.upr_shift = ilog2_exact(p.upr);
    p.useg_shift = (p.nseg > 1) ? ilog2_exact(p.S / vec) : 0;
    p.H4 = (H + 3) & ~3; p.W4 = (W + 3) & ~3; p.D4 = (dim == 3) ? ((D + 3) & ~3) : 0;
    p.pos_stride = p.H4 + p.W4 + p.D4;
    p.wy = (H + 63) / 64; p.wx = (W + 63) / 64; p.wz = (dim == 3) ? (D + 63) / 64 : 0;
    p.mask_stride = p.wy + p.wx + p.wz;
    const int nuy_max = (2 * ph < H) ? 2 * ph : H, nux_max = (2 * pw < W) ? 2 * pw : W;
    const int nuz_max = (dim == 3) ? ((2 * pd < D) ? 2 * pd : D) : 1;
    if (nuy_max > 254 || nux_max > 254 || nuz_max > 254) return MDT_ERR_UNSUPPORTED;   // positions are bytes, 255 = untouched
    // largest blocks one RoI can need in the two regions (see stage (c)); a RoI that large gets a round to itself
    size_t a_max, b_max;
    if (dim == 3) {
        a_max = (size_t)nuy_max * nux_max * pd;
        b_max = (size_t)nuy_max * nux_max * nuz_max; if (b_max < (size_t)ph * nux_max * pd) b_max = (size_t)ph * nux_max * pd;
    } else {
        a_max = (size_t)nuy_max * nux_max;
        b_max = (size_t)nuy_max * pw;
    }
    if (a_max < (size_t)p.P4) a_max = p.P4;
    // zero role keeps the bitmaps of all batch elements behind the cand scratch (regions A/B are idle there)
    size_t zero_need = (size_t)T_CAND * 8 * sizeof(short) + (size_t)B * p.bw * sizeof(u64);
    p.zero_per_vol = 0;
    if (zero_need > (size_t)T_LDS_MAX / 2) {      // large batch: zero workgroups stay inside one volume, one bitmap
        p.zero_per_vol = 1;
        zero_need = (size_t)T_CAND * 8 * sizeof(short) + (size_t)p.bw * sizeof(u64);
    }
    const int psum = ph + pw + pd;
    const size_t lds_cap = (size_t)env_int("MDT_BWD_LDS_CAP", T_LDS_MAX);
    int G = T_GMAX;
    const int gforce = env_int("MDT_BWD_G", 0);
    if (gforce > 0 && gforce < G) G = gforce;
    for (; G >= 1; --G) {
        // everything but the two regions
        const size_t rest = align16((size_t)G * psum * sizeof(SEntry)) + align16((size_t)G * p.pos_stride) +
                            align16((size_t)G * p.mask_stride * sizeof(u64)) +
                            align16((size_t)G * T_HDR * sizeof(int)) + align16((size_t)p.bw * sizeof(u64)) +
                            align16((size_t)(p.bw + 1) * sizeof(int)) + (size_t)T_CAND * 7 * sizeof(int) + (size_t)T_SEGLIST * sizeof(unsigned short) + 16 * sizeof(int);
        size_t a_fl = (size_t)((G < T_SLOTS) ? G : T_SLOTS) * p.P4;   // the resident gradient blocks
        if (a_fl < a_max) a_fl = a_max;
        if (rest + (a_fl + b_max) * sizeof(float) + 64 > lds_cap) continue;
        size_t b_fl = (lds_cap - rest) / sizeof(float) - 8 - a_fl;
        a_fl &= ~(size_t)3; b_fl &= ~(size_t)3;
        if (a_fl * sizeof(float) < (size_t)T_CAND * 8 * sizeof(short)) continue;
        p.a_floats = (int)a_fl; p.b_floats = (int)b_fl;
        size_t off = a_fl * sizeof(float);
        p.off_b = (int)off;    off += b_fl * sizeof(float);
        p.off_tab = (int)off;  off += align16((size_t)G * psum * sizeof(SEntry));
        p.off_pos = (int)off;  off += align16((size_t)G * p.pos_stride);
        p.off_mask = (int)off; off += align16((size_t)G * p.mask_stride * sizeof(u64));
        p.off_hdr = (int)off;  off += align16((size_t)G * T_HDR * sizeof(int));
        p.off_bm = (int)off;   off += align16((size_t)p.bw * sizeof(u64));
        p.off_pref = (int)off; off += align16((size_t)(p.bw + 1) * sizeof(int));
        p.off_list = (int)off; off += (size_t)T_CAND * 7 * sizeof(int);
        p.off_seg = (int)off;  off += (size_t)T_SEGLIST * sizeof(unsigned short);
        p.off_misc = (int)off; off += 16 * sizeof(int);
        lds = off;
        if (off <= lds_cap && zero_need <= off) break;
    }
    if (G < 1) return MDT_ERR_UNSUPPORTED;
    p.G = G;
    return MDT_OK;
}

// resident workgroups per CU of one kernel variant (cached per variant / LDS size); 0 if the query fails
template <typename K>
int resident_per_cu(K kernel, int nt, size_t lds)
{
    static int cached_lds = -1, cached = 0;
    if (cached_lds != (int)lds) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, nt, lds) != hipSuccess) n = 0;
        (void)hipGetLastError();
        cached = n; cached_lds = (int)lds;
    }
    return cached;
}

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

long long *g_bwd_ts = nullptr;

}  // namespace

extern "C" void mdt_debug_bwd_timestamps(long long *dev_buf) { g_bwd_ts = dev_buf; }

namespace mdt_ra {

bool bwd_territory_supported(int dim, int N, int B, int H, int W, int D, int ph, int pw, int pd, int C)
{
    TParams p;
    size_t lds = 0;
    return territory_plan(dim, N, B, H, W, D, ph, pw, pd, C, 1, p, lds) == MDT_OK;
}

// role geometry of one level (everything of TParams the plan did not fill); `zero_budget`: zero workgroups this level may use
static void territory_geometry(TParams &p, long long vol_floats, int zero_budget)
{
    const long long nvol = (long long)p.B * p.C;
    const long long vol_bytes = vol_floats * 4;
    int ssplit = env_int("MDT_BWD_SSPLIT", 1);
    if (ssplit < 1) ssplit = 1;
    p.ssplit = ssplit;
    p.dbg = env_int("MDT_BWD_DBG", 0);
    p.dbg_wg = env_int("MDT_BWD_DBG_WG", 0);
    p.ts = g_bwd_ts;
    p.scatter_base = 0; p.zero_base = 0;
    p.parts = 0;
    p.rows_per_part = p.R;
    if (vol_bytes > 32 * 1024) {
        int z = env_int("MDT_BWD_ZERO_WGS", 0);
        if (z <= 0) z = zero_budget;
        const long long total_rows = nvol * p.R;
        if (z > total_rows) z = (int)total_rows;
        if (z < 1) z = 1;
        if (p.zero_per_vol > 0) {
            int k = (int)((z + nvol / 2) / nvol);
            if (k < 1) k = 1;
            if (k > p.R) k = p.R;
            p.rows_per_part = (p.R + k - 1) / k;
            p.zero_per_vol = (p.R + p.rows_per_part - 1) / p.rows_per_part;
            p.parts = (int)(nvol * p.zero_per_vol);
        } else {
            p.rows_per_part = (int)((total_rows + z - 1) / z);
            p.parts = (int)((total_rows + p.rows_per_part - 1) / p.rows_per_part);
        }
    } else {
        p.zero_per_vol = 0;
    }
}

template <typename K>
static int zero_budget_for(K kernel, int nt, size_t lds, long long n_scatter)
{
    const int cus = cu_count();
    const long long slots = (long long)resident_per_cu(kernel, nt, lds) * cus;
    long long z = slots - n_scatter;
    if (z < cus / 2) z = cus;
    return (int)z;
}

template <typename K>
static void optin_lds(K kernel)
{
    static bool done = false;
    if (!done) {   // more than 64 KB of dynamic LDS needs the explicit opt-in
        (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_MAX);
        (void)hipGetLastError();
        done = true;
    }
}

template <int DIM, int VEC, int SL, int NT>
static void launch_single(TParams &p, size_t lds, long long vol_floats, hipStream_t s)
{
    auto kernel = crop_bwd_territory_kernel<DIM, VEC, SL, NT>;
    optin_lds(kernel);
    const long long n_scatter = (long long)p.B * p.C * env_int("MDT_BWD_SSPLIT", 1);
    territory_geometry(p, vol_floats, vol_floats * 4 > 32 * 1024 ? zero_budget_for(kernel, NT, lds, n_scatter) : 0);
    const long long grid = (long long)p.B * p.C * p.ssplit + p.parts;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(NT), lds, s, p);
}

template <int DIM, int NT>
static void launch_nt(TParams &p, size_t lds, int vec, long long vol_floats, hipStream_t s)
{
    if (vec == 4 && p.S == 32) launch_single<DIM, 4, 8, NT>(p, lds, vol_floats, s);
    else if (vec == 4 && p.S == 16) launch_single<DIM, 4, 4, NT>(p, lds, vol_floats, s);
    else if (vec == 4 && p.S == 8) launch_single<DIM, 4, 2, NT>(p, lds, vol_floats, s);
    else if (vec == 4) launch_single<DIM, 4, 0, NT>(p, lds, vol_floats, s);
    else launch_single<DIM, 1, 0, NT>(p, lds, vol_floats, s);
}

int launch_bwd_territory(int dim, const float *grads, const float *boxes, const int *box_ind, int N, int B,
                         int H, int W, int D, int ph, int pw, int pd, int C, float *out, hipStream_t s)
{
    TParams p;
    size_t lds = 0;
    const int L_ = (dim == 3) ? D : W;
    const int vec = (L_ > 0 && L_ % 4 == 0 && (((uintptr_t)out) & 15) == 0) ? 4 : 1;
    const int prc = territory_plan(dim, N, B, H, W, D, ph, pw, pd, C, vec, p, lds);
    if (prc != MDT_OK) return prc;
    p.grads = grads; p.boxes = boxes; p.box_ind = box_ind; p.out = out;
    p.level = nullptr; p.level_id = 0;
    const long long vol_floats = (long long)H * W * D;
    if ((long long)B * C * 8 > 0x3fffffffLL) return MDT_ERR_UNSUPPORTED;
    int nt = env_int("MDT_BWD_THREADS", 512);
    if (nt != 256 && nt != 512) nt = 512;
    if (vol_floats * 4 <= 32 * 1024) nt = 256;
    (void)hipGetLastError();
    if (dim == 3) { if (nt == 256) launch_nt<3, 256>(p, lds, vec, vol_floats, s); else launch_nt<3, 512>(p, lds, vec, vol_floats, s); }
    else { if (nt == 256) launch_nt<2, 256>(p, lds, vec, vol_floats, s); else launch_nt<2, 512>(p, lds, vec, vol_floats, s); }
    return check_launch();
}

// All pyramid levels in ONE launch.  Needs every level on the 8-float-segment, 16-byte-store variant (contiguous extent a
// multiple of 8, 16-byte aligned maps); otherwise MDT_ERR_UNSUPPORTED and the caller launches level by level.
int launch_bwd_territory_multi(int dim, int n_levels, const float *grads, const float *boxes, const int *batch_ix, const int *level,
                               int N, int B, int C, const int *H, const int *W, const int *D, int ph, int pw, int pd,
                               float *const *outs, hipStream_t s)
{
    if (n_levels < 1 || n_levels > T_MAX_LEVELS) return MDT_ERR_INVALID_ARGUMENT;
    TMulti m;
    m.n_levels = n_levels;
    size_t lds = 0;
    long long vols[T_MAX_LEVELS];
    for (int l = 0; l < n_levels; ++l) {
        const int Dl = (dim == 3) ? D[l] : 1;
        const int L_ = (dim == 3) ? Dl : W[l];
        if (L_ % 8 != 0 || (((uintptr_t)outs[l]) & 15) != 0) return MDT_ERR_UNSUPPORTED;
        size_t lds_l = 0;
        const int prc = territory_plan(dim, N, B, H[l], W[l], Dl, ph, pw, pd, C, 4, m.lev[l], lds_l);
        if (prc != MDT_OK) return prc;
        if (m.lev[l].S != 8) return MDT_ERR_UNSUPPORTED;
        if (lds_l > lds) lds = lds_l;
        m.lev[l].grads = grads; m.lev[l].boxes = boxes; m.lev[l].box_ind = batch_ix; m.lev[l].out = outs[l];
        m.lev[l].level = level; m.lev[l].level_id = l;
        vols[l] = (long long)H[l] * W[l] * Dl;
    }
    constexpr int NT = 512;
    (void)hipGetLastError();
    const long long nvol = (long long)B * C;
    unsigned run = 0;
    auto finish = [&](auto kernel) {
        optin_lds(kernel);
        // zero workgroups: what stays resident beside the first level's scatter workgroups goes to the levels that have
        // a zero role, in proportion to their bytes (the finest level takes nearly all of it)
        const int budget = zero_budget_for(kernel, NT, lds, nvol);
        long long big_bytes = 0;
        for (int l = 0; l < n_levels; ++l) if (vols[l] * 4 > 32 * 1024) big_bytes += vols[l];
        for (int l = 0; l < n_levels; ++l) {
            int zb = 0;
            if (vols[l] * 4 > 32 * 1024) { zb = (int)((double)budget * (double)vols[l] / (double)big_bytes); if (zb < 16) zb = 16; }
            territory_geometry(m.lev[l], vols[l], zb);
            run += (unsigned)(nvol * m.lev[l].ssplit);
            m.end[2 * l] = run;
            run += (unsigned)m.lev[l].parts;
            m.end[2 * l + 1] = run;
        }
        hipLaunchKernelGGL(kernel, dim3(run), dim3(NT), lds, s, m);
    };
    if (dim == 3) finish(crop_bwd_territory_multi_kernel<3, 4, 2, NT>);
    else finish(crop_bwd_territory_multi_kernel<2, 4, 2, NT>);
    return check_launch();
}

}  // namespace mdt_ra
