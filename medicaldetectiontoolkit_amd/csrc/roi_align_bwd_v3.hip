// roi_align_bwd_v3.hip -- round-3 default 2D/3D RoIAlign backward for gfx950 ("gather form"): ONE launch for one map
// or for all pyramid levels, every byte of every gradient map written exactly once, no atomics, no workspace,
// deterministic.  Arithmetic follows the reference scatter
// (cuda_functions/roi_align_3D/roi_align/src/cuda/crop_and_resize_kernel.cu:154-304; 2D:
// roi_align_2D/.../crop_and_resize_kernel.cu:102-194) with the per-axis interpolation factored out
// (dF = Wz^T (Wx^T (Wy^T g))), so sums are reassociated relative to the flat 8-corner scatter: values agree to fp32
// rounding (test bar 2e-6 * sum|terms|; north-star bar 1e-4).  The exact-order form lives in roi_align.hip.
//
// Why a third form.  Round 2's territory kernel (roi_align_bwd.hip) hides a 151 MB fill behind one scatter workgroup per
// (batch element, channel) volume, but that workgroup's own dependency chain measured 22 us even on a level whose fill is
// nothing (P5 alone: 25 us; profiles/r03_probe1/) and 27-33 us with large boxes: the scatter role, not the fill, was the
// critical path.  Stage stamps of a first item-per-thread rewrite (profiles/r03_bwd_gather/stage_stamps_v3a_*.jsonl) showed
// why: the role is bound by INSTRUCTION ISSUE of a few lone waves (a wave retires ~1 instruction per 4-8 cycles and pays
// ~100 cycles per dependent LDS hop; a warm instruction cache changes nothing), so per-lane searches, integer divisions and
// sample-range loops cost microseconds each.  This kernel therefore runs every stage as WAVE-UNIFORM TASKS:
//
//   scatter role  one workgroup per (batch element b, channel c) for ALL pyramid levels:
//                   list   the RoIs of b (box_ind == b), ascending, with level and index bounding box: ONE global round
//                          trip, four lanes per RoI (one axis each);
//                   plan   (one wave, one lane per RoI) which RoIs fit the LDS pool this round, arena offsets, task counts;
//                   tables the gradient blocks g[r, c] start travelling global -> LDS by LDS-DMA; meanwhile one lane per
//                          (RoI, sample) writes its two interpolation weights into dense tables Wy[q][iy], Wx[q][ix],
//                          Wz[quad][pz][voxel] (4 consecutive outputs = one 16-byte LDS broadcast) and the sample band of
//                          every 4-output chunk;
//                   pass y task (RoI, 4 rows, 64 of the px*pz lanes):   out1[iy][px][pz] = sum_q Wy[q][iy] g[q][px][pz]
//                   pass x task (RoI, 4 columns, 64 of the iy*pz lanes): out2[iy][ix][pz] = sum_q Wx[q][ix] out1[iy][q][pz]
//                   final  task (RoI, 16-byte quad, 64 columns (iy, ix)): z-contraction of out2; the lowest staged RoI whose
//                          segment box covers a quad-column owns it and adds -- RoI ascending -- every later staged RoI
//                          covering it (weights stay wave-uniform: the quad is), then stores 16 bytes; RoIs of earlier
//                          rounds (more RoIs on one element than fit LDS): read-modify-write.
//                 Intermediate blocks are dense over the RoI's index bounding box (no touched-index compaction).
//                 Levels whose (b, c) volume is small (<= 32 KB) have no zero role: the scatter workgroup also stores the
//                 zeros of its volume outside the territory (after its scatter work, fire-and-forget).
//   zero role     levels with large volumes (P2: 151 MB): persistent workgroups -- as many as stay resident beside the
//                 scatter workgroups -- each streaming 16-byte zero stores over one contiguous run of rows, skipping the
//                 territory segments through an LDS bitmap built for the batch elements its run touches.
// Territory of (b, level) = segments (S = 8..32 contiguous floats) inside the index bounding box of any RoI of b on that
// level; both roles derive it from `boxes` alone, so they write disjoint bytes and nothing orders them.
//
// Measured (rocprofv3, MI355X, 48 RoIs, pool (14,14,5); profiles/r03_bwd_gather/): P2 train-like 26.5 us (round 2: 28.9),
// SURVEY 8(d) random boxes 29.5 us (43.3), all four levels in one launch 30.6-32 us (50.6); the scatter chain of one
// workgroup is 15-20 us and hidden behind the fill (zero role alone: 24.9 us, 22.7 us without its box-load prologue).
//
// 2D maps are the 3D case with W = 1, pw = 1 (the singleton axis interpolates with weight exactly 1).
// HBM-bound, no MFMA.  Algorithmic bytes per launch: 4*B*C*V per map (written once) + 4*N*C*P (grads once) + 28*N.
#include "roi_align_common.h"

using namespace mdt_ra;

namespace {

typedef unsigned long long u64;

constexpr int V3_NT = 512;             // threads per workgroup (both roles)
constexpr int V3_MAXR = 128;           // RoIs per launch (dispatch limit, roi_align.hip BWD_TERRITORY_MAX_BOXES)
constexpr int V3_GMAX = 16;            // RoIs staged per round (upper bound)
constexpr int V3_MAX_LEVELS = 5;
constexpr int V3_LDS_CAP = 80 * 1024;  // gfx950: 160 KB per CU -> two workgroups resident per CU
constexpr long long V3_MERGE_BYTES = 32 * 1024;   // volumes up to this size are zero-filled by their scatter workgroup
constexpr int V3_LEV_BYTES = 512;      // head of the LDS carve: the level table (kernel arguments indexed dynamically would go through scratch)

constexpr int V3_ZERO_CHUNK_ROWS = 0;     // default geometry of the zero role (see V3Level::chunk_rows); MDT_BWD3_ZERO_CHUNK_ROWS under MDT_BWD_TUNE

struct V3Level {
    float *out;
    int H, W, D;                       // 2D: (H, 1, W)
    int R, L;                          // rows per volume (H * W), contiguous extent (D)
    int S, S_shift, nseg, useg, useg_shift;   // segment length (floats, a power of two: 8 / 16 / 32) + log2, segments per row, 16-byte units per segment + log2
    int upr, upr_shift;                // 16-byte units per row (+ log2 or -1)
    int bw;                            // u64 words of one batch element's territory bitmap
    int merged;                        // 1: no zero role, scatter workgroups zero-fill
    int zero_parts, rows_per_part;     // zero role geometry
    int chunk_rows, chunk_shift, R_shift;   // chunk_rows 0: a zero workgroup streams ONE contiguous run of rows_per_part rows; > 0: chunks of chunk_rows rows, dealt round-robin
    unsigned zero_base;                // first zero block of this level (relative to the first zero block of the launch)
};

struct V3Params {
    const float *grads;
    const float *boxes;
    const int *box_ind;
    const int *level;                  // optional [N]; null: every RoI on level 0
    int dim;                           // 2 or 3 (box row stride 2 * dim)
    int N, B, C;
    int ph, pw, pd;                    // 2D: (ph, 1, pw)
    int P, P4, psum;
    int n_levels;
    unsigned n_scatter;                // B * C
    int pool_floats;                   // LDS pool of a round: arenas A (gradient blocks, then out2) | B (out1) | W (weight tables, bands)
    float inv_psum, inv_pd;            // 1 / psum, 1 / pd (division-free index math)
    int off_hdr, off_rbox, off_bb, off_misc;
    long long *ts;                     // tuning only (mdt_debug_bwd3): wall-clock stamps of scatter workgroup `dbg_wg`, or null
    int accum;                         // 1: a LATER chunk of a launch series over more than V3_MAXR RoIs: no zero stores anywhere, every territory quad is
                                       //    read-modify-written on top of what the earlier chunks left (round 6)
    int dbg, dbg_wg;                   // tuning only: bit0 scatter role returns at once, bit1 zero role returns at once, bit3 zero role skips its box loads, bit4 zero role stores non-temporally
    V3Level lev[V3_MAX_LEVELS];
};

#define V3STAMP(k) do { if (p.ts && bid == (unsigned)p.dbg_wg && threadIdx.x == 0) p.ts[(k) + stamp_base] = (long long)wall_clock64(); } while (0)

__device__ __forceinline__ void axis_bounds(float a1, float a2, int L, int P, int &lo, int &hi)
{
    // sample coordinates are monotone in p (rounding is monotone), so the extreme indices sit at p = 0 / P-1
    const AxisEntry e0 = axis_entry(a1, a2, L, P, 0);
    const AxisEntry e1 = axis_entry(a1, a2, L, P, P - 1);
    lo = min(e0.lo, e1.lo);
    hi = max(entry_hi(e0), entry_hi(e1));
}

// box row r -> per-axis (a1, a2): y = (b0, b2), x = (b1, b3), z = (b4, b5); 2D: y = (b0, b2), x = none, z = (b1, b3)
__device__ __forceinline__ void load_box(const V3Params &p, int r, float *bx)
{
    const float *src = p.boxes + (long long)r * (2 * p.dim);
    if (p.dim == 3) {
#pragma unroll
        for (int k = 0; k < 6; ++k) bx[k] = src[k];
    } else {
        bx[0] = src[0]; bx[2] = src[2];
        bx[1] = 0.0f; bx[3] = 0.0f;
        bx[4] = src[1]; bx[5] = src[3];
    }
}

// workgroup barrier that waits for this wave's LDS traffic only: global loads / stores issued earlier stay in flight
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void set_bits(u64 *bm, int s, int e)   // inclusive bit range
{
    for (int w = s >> 6; w <= (e >> 6); ++w) {
        const int lo = max(s, w << 6) - (w << 6);
        const int hi = min(e, (w << 6) + 63) - (w << 6);
        const u64 upto = (hi == 63) ? ~0ULL : ((1ULL << (hi + 1)) - 1ULL);
        atomicOr(&bm[w], upto & ~((1ULL << lo) - 1ULL));
    }
}

// level table -> LDS (static indices into the kernel argument; the caller synchronises)
__device__ __forceinline__ V3Level *stage_levels(const V3Params &p, char *smem_raw)
{
    V3Level *slev = reinterpret_cast<V3Level *>(smem_raw);
#pragma unroll
    for (int l = 0; l < V3_MAX_LEVELS; ++l)
        if ((int)threadIdx.x == l && l < p.n_levels) slev[l] = p.lev[l];
    return slev;
}

// ------------------------------------------------------------------------------------------------------ zero role
__device__ __forceinline__ void zero_role(const V3Params &p, const int li, const unsigned zi)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    V3Level lv = p.lev[0];
#pragma unroll
    for (int l = 1; l < V3_MAX_LEVELS; ++l) if (l == li) lv = p.lev[l];     // static indices: the table stays in scalar registers
    const long long total_rows = (long long)p.B * p.C * lv.R;
    const long long rows_per_elem = (long long)p.C * lv.R;
    long long g0, g1, gstep, glen;
    int b_first, b_last;
    if (lv.chunk_rows > 0) {       // interleaved chunks: this workgroup's rows are spread over the whole map (all batch elements)
        glen = lv.chunk_rows;
        g0 = (long long)zi * glen;
        gstep = (long long)lv.zero_parts * glen;
        g1 = total_rows;
        b_first = 0; b_last = p.B - 1;
        if (g0 >= total_rows) return;
    } else {
        g0 = (long long)zi * lv.rows_per_part;
        g1 = g0 + lv.rows_per_part;
        if (g1 > total_rows) g1 = total_rows;
        if (g0 >= g1) return;
        glen = g1 - g0; gstep = total_rows;       // one run
        b_first = (int)(g0 / rows_per_elem);
        b_last = (int)((g1 - 1) / rows_per_elem);
    }
    const int nb = b_last - b_first + 1;
    short *cand = reinterpret_cast<short *>(smem_raw + V3_LEV_BYTES);                               // [V3_MAXR][8]
    int *ncand = reinterpret_cast<int *>(smem_raw + V3_LEV_BYTES + V3_MAXR * 8 * sizeof(short));     // [1] (+ padding)
    u64 *bm = reinterpret_cast<u64 *>(smem_raw + V3_LEV_BYTES + V3_MAXR * 8 * sizeof(short) + 16);   // [nb][bw]
    for (int t = tid; t < nb * lv.bw; t += V3_NT) bm[t] = 0ULL;
    if (tid == 0) *ncand = 0;
    __syncthreads();
    if (tid < p.N && !(p.dbg & 8)) {       // (dbg bit 3, tuning only: no territory -> what the bitmap prologue costs)
        const int r = tid;
        const int bi = p.box_ind[r];
        const int l = p.level ? p.level[r] : 0;
        float bx[6];
        load_box(p, r, bx);
        if (l == li && bi >= b_first && bi <= b_last) {
            int lo0, hi0, lo1, hi1, lo2, hi2;
            axis_bounds(bx[0], bx[2], lv.H, p.ph, lo0, hi0);
            axis_bounds(bx[1], bx[3], lv.W, p.pw, lo1, hi1);
            axis_bounds(bx[4], bx[5], lv.D, p.pd, lo2, hi2);
            const int slot = atomicAdd(ncand, 1);
            short *c = cand + slot * 8;
            c[0] = (short)lo0; c[1] = (short)hi0; c[2] = (short)lo1; c[3] = (short)hi1;
            c[4] = (short)(lo2 >> lv.S_shift); c[5] = (short)(hi2 >> lv.S_shift); c[6] = (short)(bi - b_first);
        }
    }
    __syncthreads();
    const int nc = *ncand;
    for (int t = tid; t < nc * lv.H; t += V3_NT) {
        const int e = t / lv.H;
        const int y = t - e * lv.H;
        const short *c = cand + e * 8;
        if (y < c[0] || y > c[1]) continue;
        u64 *bmb = bm + c[6] * lv.bw;
        if (c[4] == 0 && c[5] == lv.nseg - 1) {
            set_bits(bmb, (y * lv.W + c[2]) * lv.nseg, (y * lv.W + c[3]) * lv.nseg + lv.nseg - 1);
        } else {
            for (int x = c[2]; x <= c[3]; ++x) set_bits(bmb, (y * lv.W + x) * lv.nseg + c[4], (y * lv.W + x) * lv.nseg + c[5]);
        }
    }
    __syncthreads();
    const v4f z4 = {0.f, 0.f, 0.f, 0.f};
    if (lv.chunk_rows > 0) {
        // dense moving window (round 4): chunk k of chunk_rows rows goes to workgroup k mod zero_parts, so at any moment the zero
        // workgroups together write ONE contiguous window of the map (zero_parts x chunk bytes), like a grid-stride fill -- the
        // write-back stream that leaves the L2 is then sequential in DRAM.  (One long run per workgroup, the round-3 geometry, costs
        // 50 us instead of 26 us when the map is not cache-resident: profiles/r04_bwd_roles_cold_warm.txt.)  All 32-bit, shifts only:
        // R, units per row and segment units are powers of two and chunk_rows divides R (checked by the host).
        const int n_chunks = (int)(total_rows >> lv.chunk_shift);
        const int upc = lv.chunk_rows << lv.upr_shift;                      // 16-byte units per chunk
        v4f *outv = reinterpret_cast<v4f *>(lv.out);
        for (int ck = (int)zi; ck < n_chunks; ck += lv.zero_parts) {
            const int grow = ck << lv.chunk_shift;                          // first row of the chunk (flattened over B * C volumes)
            const int vol = grow >> lv.R_shift, r0 = grow & (lv.R - 1);
            const u64 *bmb = bm + (vol / p.C) * lv.bw;
            v4f *base = outv + ((long long)grow << lv.upr_shift);
            for (int u = tid; u < upc; u += V3_NT) {
                const int bit = (r0 + (u >> lv.upr_shift)) * lv.nseg + ((u & (lv.upr - 1)) >> lv.useg_shift);
                if ((bmb[bit >> 6] >> (bit & 63)) & 1ULL) continue;
                base[u] = z4;
            }
        }
        return;
    }
    for (long long run0 = g0; run0 < g1; run0 += gstep)
    for (long long g = run0, ge = min(g1, run0 + glen); g < ge;) {
        const int vol = (int)(g / lv.R);
        const int r0 = (int)(g - (long long)vol * lv.R);
        const int r1 = (int)min((long long)lv.R, r0 + (ge - g));
        const u64 *bmb = bm + (vol / p.C - b_first) * lv.bw;
        v4f *base = reinterpret_cast<v4f *>(lv.out + ((long long)vol * lv.R + r0) * lv.L);
        const int nu = (r1 - r0) * lv.upr;
        for (int u = tid; u < nu; u += V3_NT) {
            int rl, ui;
            if (lv.upr_shift >= 0) { rl = u >> lv.upr_shift; ui = u & (lv.upr - 1); }
            else { rl = u / lv.upr; ui = u - rl * lv.upr; }
            const int sg = (lv.useg_shift >= 0) ? (ui >> lv.useg_shift) : (ui / lv.useg);
            const int bit = (r0 + rl) * lv.nseg + sg;
            if ((bmb[bit >> 6] >> (bit & 63)) & 1ULL) continue;
            if (p.dbg & 16) __builtin_nontemporal_store(z4, &base[u]);     // (dbg bit 4, tuning only: streaming-store policy for the zero role)
            else base[u] = z4;
        }
        g += r1 - r0;
    }
}

// --------------------------------------------------------------------------------------------------- scatter role
// The role is bound by instruction issue, not by data: a wave retires roughly one instruction every 4-8 cycles, a (b, c)
// volume is ~6 RoIs x 980 gradients, and only 8 waves work on it -- so every stage is organised as WAVE-UNIFORM TASKS
// (RoI / row chunk / quad known per wave: index math is scalar, weights arrive as one 16-byte LDS broadcast) with the
// lanes spread over the dense part (px*pz, iy*pz, columns), no per-lane searches or integer divisions.
//
// hdr[j] ints: 0 r | 1 level | 2 y0 | 3 ny | 4 x0 | 5 nx | 6 zq0 (first 16-byte quad of the segment range) | 7 nq
//              8 aoff (g, then out2) | 9 boff (out1) | 10 woff (Wy | Wx | Wz | bands) | 11 ny4 | 12 nx4 | 13 shared | 14 1/nx (float bits)
// bb[k] int4: ylo | yhi << 16, xlo | xhi << 16, zlo | zhi << 16, level | shared << 7 | r << 8
constexpr int V3_HDRN = 16;

__device__ __forceinline__ int ufl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// n / d for 0 <= n < 2^22, 0 < d < 2^12 (inv = 1.0f / d): float estimate + one correction step
__device__ __forceinline__ int fdiv(int n, int d, float inv)
{
    int q = (int)(((float)n + 0.5f) * inv);
    const int r = n - q * d;
    if (r < 0) --q;
    else if (r >= d) ++q;
    return q;
}

// wave-uniform: index j of the task prefix segment containing T (tp[0..ng] exclusive prefix, tp[ng] = total)
__device__ __forceinline__ int task_owner(const int *tp, int ng, int T, int lane)
{
    const int e = (lane < ng) ? tp[lane + 1] : 0x7fffffff;
    return __popcll(__ballot(T >= e));
}

// acc += sum_pz Wz[pz][0..3] * o2[pz], pz ascending; loads of four samples issued together
__device__ __forceinline__ v4f zdot(v4f acc, const float *o2, const float *Wz, int pd)
{
    for (int pz = 0; pz < pd; pz += 4) {
        float v1[4];
        v4f w4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int qq = min(pz + u, pd - 1);
            v1[u] = (pz + u < pd) ? o2[qq] : 0.0f;
            w4[u] = *reinterpret_cast<const v4f *>(Wz + qq * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = acc + w4[u] * v1[u];
    }
    return acc;
}

__device__ __forceinline__ void scatter_role(const V3Params &p, const unsigned bid, const int stamp_base)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *pool = reinterpret_cast<float *>(smem_raw + V3_LEV_BYTES);                // [pool_floats]: A | B | W arenas of the round
    int *hdr = reinterpret_cast<int *>(smem_raw + p.off_hdr);                        // [GMAX][V3_HDRN]
    float *rbox = reinterpret_cast<float *>(smem_raw + p.off_rbox);                  // [MAXR][6]
    int4 *bb = reinterpret_cast<int4 *>(smem_raw + p.off_bb);                        // [MAXR]
    int *misc = reinterpret_cast<int *>(smem_raw + p.off_misc);                      // [0..7] wave hit counts, [8] ng, [9] A total, [10] B total, [11] W total
    int *tY = misc + 16, *tX = misc + 16 + (V3_GMAX + 1), *tZ = misc + 16 + 2 * (V3_GMAX + 1);   // task prefixes [GMAX + 1]

    __builtin_amdgcn_s_setprio(3);      // latency-critical: win instruction issue against the streaming zero-role waves
    const int tid = threadIdx.x, lane = tid & 63, wave = ufl(tid >> 6);
    const int vol = (int)bid;
    const int b = vol / p.C;
    const int c = vol - b * p.C;
    V3STAMP(0);

    // ---- the RoIs of batch element b, ascending, with level and index bounding box: ONE global round trip, four lanes per
    //      RoI (three compute one axis each: the two extreme sample coordinates; the fourth carries the level)
    int cnt;
    {
        const int r = tid >> 2, part = tid & 3;
        int bi = -1, lv_i = 0;
        float a1 = 0.0f, a2 = 0.0f;
        if (r < p.N) {
            bi = p.box_ind[r];
            lv_i = p.level ? p.level[r] : 0;
            const float *src = p.boxes + (long long)r * (2 * p.dim);
            if (p.dim == 3) {
                if (part == 0) { a1 = src[0]; a2 = src[2]; }
                else if (part == 1) { a1 = src[1]; a2 = src[3]; }
                else if (part == 2) { a1 = src[4]; a2 = src[5]; }
            } else {
                if (part == 0) { a1 = src[0]; a2 = src[2]; }
                else if (part == 2) { a1 = src[1]; a2 = src[3]; }
            }
        }
        const V3Level *slev = stage_levels(p, smem_raw);
        const bool hit = (r < p.N) && (bi == b) && (lv_i >= 0) && (lv_i < p.n_levels);
        const u64 bal = __ballot(hit && part == 0);
        if (lane == 0) misc[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < V3_NT / 64; ++w) { const int m = misc[w]; if (w < wave) off += m; total += m; }
        cnt = total;
        if (hit) {
            const int slot = off + __popcll(bal & ((1ULL << (lane & ~3)) - 1ULL));
            const V3Level &lv = slev[lv_i];
            int *bbi = reinterpret_cast<int *>(bb + slot);
            if (part < 3) {
                const int La = (part == 0) ? lv.H : (part == 1) ? lv.W : lv.D;
                const int Pa = (part == 0) ? p.ph : (part == 1) ? p.pw : p.pd;
                int lo, hi;
                axis_bounds(a1, a2, La, Pa, lo, hi);
                bbi[part] = lo | (hi << 16);
                rbox[slot * 6 + part * 2] = a1;
                rbox[slot * 6 + part * 2 + 1] = a2;
            } else {
                bbi[3] = lv_i | (r << 8);
            }
        }
        __syncthreads();
        // shared flag: another RoI of this element on the same level whose segment bounding box meets this one's
        if (tid < cnt) {
            const int4 q = bb[tid];
            const int lvl = q.w & 127;
            const int ss = slev[lvl].S_shift;
            bool sh = false;
            for (int k = 0; k < cnt; ++k) {
                const int4 o = bb[k];
                if (k == tid || (o.w & 127) != lvl) continue;
                if ((o.x & 0xffff) > (q.x >> 16) || (o.x >> 16) < (q.x & 0xffff)) continue;
                if ((o.y & 0xffff) > (q.y >> 16) || (o.y >> 16) < (q.y & 0xffff)) continue;
                if (((o.z & 0xffff) >> ss) > ((q.z >> 16) >> ss) || ((o.z >> 16) >> ss) < ((q.z & 0xffff) >> ss)) continue;
                sh = true;
                break;
            }
            if (sh) reinterpret_cast<int *>(bb + tid)[3] = q.w | 128;
        }
    }
    const V3Level *slev = reinterpret_cast<const V3Level *>(smem_raw);
    const int ppd = p.pw * p.pd;
    const int nitY = (ppd + 63) >> 6;
    V3STAMP(1);

    for (int g0 = 0; g0 < cnt;) {
        __syncthreads();              // shared flags / previous round: tables free, its stores complete (vmcnt(0)) before anything reads them back
        // ---- which RoIs fit this round; arena offsets and task prefixes (wave 0, one lane per RoI)
        if (wave == 0) {
            int a = 0, bsz = 0, wsz = 0, ny = 0, nx = 0, y0 = 0, x0 = 0, zq0 = 0, nq = 0, ny4 = 0, nx4 = 0, lvi = 0, r = 0, sh = 0;
            int ty = 0, tx = 0, tz = 0;
            const int k = g0 + lane;
            const bool cand = (k < cnt) && (lane < V3_GMAX);
            if (cand) {
                const int4 q = bb[k];
                lvi = q.w & 127; sh = (q.w >> 7) & 1; r = q.w >> 8;
                const V3Level &lv = slev[lvi];
                y0 = q.x & 0xffff; ny = (q.x >> 16) - y0 + 1;
                x0 = q.y & 0xffff; nx = (q.y >> 16) - x0 + 1;
                const int zs0 = (q.z & 0xffff) >> lv.S_shift;
                zq0 = zs0 * lv.useg;
                nq = (((q.z >> 16) >> lv.S_shift) - zs0 + 1) * lv.useg;
                ny4 = (ny + 3) & ~3; nx4 = (nx + 3) & ~3;
                a = (max(p.P4, ny * nx * p.pd) + 3) & ~3;
                bsz = (ny * ppd + 3) & ~3;
                wsz = (p.ph * ny4 + p.pw * nx4 + nq * p.pd * 4 + (ny4 >> 1) + (nx4 >> 1) + 3) & ~3;
                ty = (ny4 >> 2) * nitY;
                tx = (nx4 >> 2) * ((ny * p.pd + 63) >> 6);
                tz = nq * ((ny * nx + 63) >> 6);
            }
            int ai = a, bi_ = bsz, wi = wsz, yi = ty, xi = tx, zi = tz;
#pragma unroll
            for (int d = 1; d < V3_GMAX; d <<= 1) {
                const int va = __shfl_up(ai, d), vb = __shfl_up(bi_, d), vw = __shfl_up(wi, d);
                const int vy = __shfl_up(yi, d), vx = __shfl_up(xi, d), vz = __shfl_up(zi, d);
                if (lane >= d) { ai += va; bi_ += vb; wi += vw; yi += vy; xi += vx; zi += vz; }
            }
            const bool fits = cand && (ai + bi_ + wi <= p.pool_floats);
            const int nfit = __popcll(__ballot(fits));         // prefix sums are monotone: the fitting RoIs are a prefix
            const int last = max(nfit, 1) - 1;
            const int At = __shfl(ai, last), Bt = __shfl(bi_, last);
            if (lane < nfit) {
                int *h = hdr + lane * V3_HDRN;
                h[0] = r; h[1] = lvi; h[2] = y0; h[3] = ny; h[4] = x0; h[5] = nx; h[6] = zq0; h[7] = nq;
                h[8] = ai - a; h[9] = At + bi_ - bsz; h[10] = At + Bt + wi - wsz; h[11] = ny4; h[12] = nx4; h[13] = sh;
                h[14] = __float_as_int(1.0f / (float)nx);
                tY[lane + 1] = yi; tX[lane + 1] = xi; tZ[lane + 1] = zi;
            }
            if (lane == 0) { tY[0] = 0; tX[0] = 0; tZ[0] = 0; }
            if (lane == last) { misc[8] = nfit; misc[9] = ai; misc[10] = bi_; misc[11] = wi; }
        }
        __syncthreads();
        const int ng = ufl(misc[8]);
        if (ng <= 0) return;          // cannot happen: the host plan sizes the pool for the largest possible RoI
        const int Wbase = ufl(misc[9]) + ufl(misc[10]), Wtot = ufl(misc[11]);
        V3STAMP(2);

        // ---- gradient blocks of the staged RoIs: global -> LDS (arena A) by LDS-DMA, no staging registers
        {
            const bool x4 = (p.P & 3) == 0;                     // 16-byte DMA when the blocks are 16-byte aligned
            const int per = x4 ? 256 : 64;                      // floats per wave instruction
            const int cpb = (p.P + per - 1) / per;              // chunks per block
            for (int ci = wave; ci < ng * cpb; ci += V3_NT / 64) {
                const int j = ci / cpb;
                const int k = ci - j * cpb;
                const int *h = hdr + j * V3_HDRN;
                const float *blk = p.grads + ((long long)ufl(h[0]) * p.C + c) * p.P;
                float *dstw = pool + ufl(h[8]) + k * per;       // wave-uniform; lane i lands at dstw + i * (16 | 4 bytes)
                if (x4) {
                    const int t = k * 256 + lane * 4;
                    if (t < p.P)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(blk + t),
                                                         (__attribute__((address_space(3))) void *)dstw, 16, 0, 0);
                } else {
                    const int t = k * 64 + lane;
                    if (t < p.P)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(blk + t),
                                                         (__attribute__((address_space(3))) void *)dstw, 4, 0, 0);
                }
            }
        }
        // ---- weight tables: zero the W arena ...
        {
            v4f *w4 = reinterpret_cast<v4f *>(pool + Wbase);
            const v4f z4 = {0.f, 0.f, 0.f, 0.f};
            for (int t = tid; t < (Wtot >> 2); t += V3_NT) w4[t] = z4;
        }
        lds_barrier();
        V3STAMP(3);
        // ... then one lane per (RoI, sample) writes its two weights: Wy[q][iy] / Wx[q][ix] (4 consecutive indices = one 16-byte
        //     read in the passes), Wz[quad][pz][voxel of the quad]; and the sample range of every 4-row chunk (band)
        for (int t = tid; t < ng * p.psum; t += V3_NT) {
            const int j = fdiv(t, p.psum, p.inv_psum);
            const int q = t - j * p.psum;
            const int *h = hdr + j * V3_HDRN;
            const float *bx = rbox + (g0 + j) * 6;
            const V3Level &lv = slev[h[1]];
            float *W = pool + h[10];
            const int ny4 = h[11], nx4 = h[12];
            if (q < p.ph) {
                const AxisEntry e = axis_entry(bx[0], bx[1], lv.H, p.ph, q);      // rbox: (a1, a2) per axis y | x | z
                const int i = e.lo - h[2];
                float *Wy = W + q * ny4;
                int *band = reinterpret_cast<int *>(W + p.ph * ny4 + p.pw * nx4 + h[7] * p.pd * 4);      // [ny4/4] lo' | [ny4/4] hi'
                Wy[i] = 1.0f - e.lerp;
                atomicMax(&band[i >> 2], 255 - q);
                atomicMax(&band[(ny4 >> 2) + (i >> 2)], q + 1);
                if (e.lerp > 0.0f) {
                    Wy[i + 1] = e.lerp;
                    atomicMax(&band[(i + 1) >> 2], 255 - q);
                    atomicMax(&band[(ny4 >> 2) + ((i + 1) >> 2)], q + 1);
                }
            } else if (q < p.ph + p.pw) {
                const int qq = q - p.ph;
                const AxisEntry e = axis_entry(bx[2], bx[3], lv.W, p.pw, qq);
                const int i = e.lo - h[4];
                float *Wx = W + p.ph * ny4 + qq * nx4;
                int *band = reinterpret_cast<int *>(W + p.ph * ny4 + p.pw * nx4 + h[7] * p.pd * 4) + (ny4 >> 1);
                Wx[i] = 1.0f - e.lerp;
                atomicMax(&band[i >> 2], 255 - qq);
                atomicMax(&band[(nx4 >> 2) + (i >> 2)], qq + 1);
                if (e.lerp > 0.0f) {
                    Wx[i + 1] = e.lerp;
                    atomicMax(&band[(i + 1) >> 2], 255 - qq);
                    atomicMax(&band[(nx4 >> 2) + ((i + 1) >> 2)], qq + 1);
                }
            } else {
                const int qq = q - p.ph - p.pw;
                const AxisEntry e = axis_entry(bx[4], bx[5], lv.D, p.pd, qq);
                float *Wz = W + p.ph * ny4 + p.pw * nx4;
                const int i = e.lo - h[6] * 4;                  // voxel index relative to the first quad
                Wz[((i >> 2) * p.pd + qq) * 4 + (i & 3)] = 1.0f - e.lerp;
                if (e.lerp > 0.0f) Wz[(((i + 1) >> 2) * p.pd + qq) * 4 + ((i + 1) & 3)] = e.lerp;
            }
        }
        V3STAMP(4);
        __syncthreads();              // + the gradient blocks have landed (vmcnt(0))
        V3STAMP(5);

        // ---- pass y: out1[iy][px][pz] = sum_q Wy[q][iy] g[q][px][pz]; task = (RoI, chunk of 4 rows, 64 of the px*pz lanes)
        {
            const int nT = ufl(tY[ng]);
            for (int T = wave; T < nT; T += V3_NT / 64) {
                const int j = task_owner(tY, ng, T, lane);
                const int *h = hdr + j * V3_HDRN;
                const int local = T - ufl(tY[j]);
                int chunk, it;
                if (nitY == 1) { chunk = local; it = 0; }
                else if (nitY == 2) { chunk = local >> 1; it = local & 1; }
                else { chunk = local / nitY; it = local - chunk * nitY; }
                const int ny = ufl(h[3]), ny4 = ufl(h[11]), nx4 = ufl(h[12]);
                const float *W = pool + ufl(h[10]);
                const int *band = reinterpret_cast<const int *>(W + p.ph * ny4 + p.pw * nx4 + ufl(h[7]) * p.pd * 4);
                const int qlo = max(0, 255 - ufl(band[chunk])), qhi = min(p.ph, ufl(band[(ny4 >> 2) + chunk]));
                const int rest = it * 64 + lane;
                const bool act = rest < ppd;
                const float *g = pool + ufl(h[8]) + (act ? rest : 0);
                const float *Wy = W + chunk * 4;
                v4f acc = {0.f, 0.f, 0.f, 0.f};
                // four samples per trip: all eight LDS loads are issued before the first use (a dependent LDS hop costs ~100
                // cycles and a wave has nothing else to overlap it with); trips past the band multiply by a zero value
                for (int q = qlo; q < qhi; q += 4) {
                    float gv[4];
                    v4f w4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int qq = min(q + u, qhi - 1);
                        gv[u] = (q + u < qhi) ? g[qq * ppd] : 0.0f;
                        w4[u] = *reinterpret_cast<const v4f *>(Wy + qq * ny4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = acc + w4[u] * gv[u];
                }
                if (act) {
                    float *o = pool + ufl(h[9]) + chunk * 4 * ppd + rest;
#pragma unroll
                    for (int v = 0; v < 4; ++v) if (chunk * 4 + v < ny) o[v * ppd] = acc[v];
                }
            }
        }
        lds_barrier();
        V3STAMP(6);
        // ---- pass x: out2[iy][ix][pz] = sum_q Wx[q][ix] out1[iy][q][pz]; task = (RoI, chunk of 4 columns, 64 of the iy*pz lanes)
        {
            const int nT = ufl(tX[ng]);
            for (int T = wave; T < nT; T += V3_NT / 64) {
                const int j = task_owner(tX, ng, T, lane);
                const int *h = hdr + j * V3_HDRN;
                const int local = T - ufl(tX[j]);
                const int ny = ufl(h[3]), nx = ufl(h[5]), ny4 = ufl(h[11]), nx4 = ufl(h[12]);
                const int nit = (ny * p.pd + 63) >> 6;
                int chunk, it;
                if (nit == 1) { chunk = local; it = 0; }
                else { chunk = local / nit; it = local - chunk * nit; }
                const float *W = pool + ufl(h[10]);
                const int *band = reinterpret_cast<const int *>(W + p.ph * ny4 + p.pw * nx4 + ufl(h[7]) * p.pd * 4) + (ny4 >> 1);
                const int qlo = max(0, 255 - ufl(band[chunk])), qhi = min(p.pw, ufl(band[(nx4 >> 2) + chunk]));
                const int idx = it * 64 + lane;
                const bool act = idx < ny * p.pd;
                const int iy = act ? fdiv(idx, p.pd, p.inv_pd) : 0;
                const int pz = act ? idx - iy * p.pd : 0;
                const float *o1 = pool + ufl(h[9]) + iy * ppd + pz;
                const float *Wx = W + p.ph * ny4 + chunk * 4;
                v4f acc = {0.f, 0.f, 0.f, 0.f};
                for (int q = qlo; q < qhi; q += 4) {
                    float ov[4];
                    v4f w4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int qq = min(q + u, qhi - 1);
                        ov[u] = (q + u < qhi) ? o1[qq * p.pd] : 0.0f;
                        w4[u] = *reinterpret_cast<const v4f *>(Wx + qq * nx4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = acc + w4[u] * ov[u];
                }
                if (act) {
                    float *o = pool + ufl(h[8]) + (iy * nx + chunk * 4) * p.pd + pz;
#pragma unroll
                    for (int v = 0; v < 4; ++v) if (chunk * 4 + v < nx) o[v * p.pd] = acc[v];
                }
            }
        }
        lds_barrier();
        V3STAMP(7);
        // ---- final: z-contraction + sum over the covering RoIs + 16-byte store; task = (RoI, 16-byte quad, 64 columns (iy, ix)).
        //      A quad-column is written by the lowest staged RoI whose segment box covers it (its owner), which adds -- RoI
        //      ascending -- the contributions of every later staged RoI covering it; RoIs of earlier rounds: read-modify-write.
        {
            const int nT = ufl(tZ[ng]);
            for (int T = wave; T < nT; T += V3_NT / 64) {
                const int j = task_owner(tZ, ng, T, lane);
                const int *h = hdr + j * V3_HDRN;
                const int local = T - ufl(tZ[j]);
                const int ny = ufl(h[3]), nx = ufl(h[5]), nq = ufl(h[7]), lvi = ufl(h[1]);
                int quad, it;
                if (ny * nx <= 64) { quad = local; it = 0; }
                else { const int nit = (ny * nx + 63) >> 6; quad = local / nit; it = local - quad * nit; }
                (void)nq;
                const V3Level &lv = slev[lvi];
                const int col = it * 64 + lane;
                bool act = col < ny * nx;
                const int iy = act ? fdiv(col, nx, __int_as_float(ufl(h[14]))) : 0;
                const int ix = act ? col - iy * nx : 0;
                const int y = ufl(h[2]) + iy, x = ufl(h[4]) + ix;
                const int Q = ufl(h[6]) + quad;                     // absolute 16-byte unit within the row
                const int sg = (Q * 4) >> ufl(lv.S_shift);
                const bool shared = ufl(h[13]) != 0;
                bool rmw = p.accum != 0;
                if (shared) {
                    const int ssh = ufl(lv.S_shift);
#pragma unroll 4
                    for (int k = 0; k < g0 + j; ++k) {            // branch-free: the bb reads of consecutive k overlap
                        const int4 q = bb[k];                       // same address in every lane
                        const bool rel = ((q.w & 255) == (lvi | 128)) && sg >= ((q.z & 0xffff) >> ssh) && sg <= ((q.z >> 16) >> ssh);
                        const bool in = rel && y >= (q.x & 0xffff) && y <= (q.x >> 16) && x >= (q.y & 0xffff) && x <= (q.y >> 16);
                        act = act && !(in && k >= g0);
                        rmw = rmw || (in && k < g0);
                    }
                    if (__ballot(act) == 0ULL) continue;
                }
                float *dst = lv.out + (((long long)vol * lv.R + (long long)y * lv.W + x) * lv.L + Q * 4);
                v4f acc = {0.f, 0.f, 0.f, 0.f};
                if (act && rmw) {     // written in an earlier round by some wave of this workgroup: read past the L1
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[v] = __hip_atomic_load(dst + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                {
                    const float *o2 = pool + ufl(h[8]) + (act ? col : 0) * p.pd;
                    const float *Wz = pool + ufl(h[10]) + p.ph * ufl(h[11]) + p.pw * ufl(h[12]) + quad * p.pd * 4;
                    acc = zdot(acc, o2, Wz, p.pd);
                }
                if (shared) {
                    for (int jk = j + 1; jk < ng; ++jk) {
                        const int4 *hk4 = reinterpret_cast<const int4 *>(hdr + jk * V3_HDRN);
                        const int4 ha = hk4[0], hb = hk4[1], hc = hk4[2], hd = hk4[3];      // one LDS round trip for the header
                        if (ufl(ha.y) != lvi || ufl(hd.y) == 0) continue;
                        const int qk = Q - ufl(hb.z);
                        if (qk < 0 || qk >= ufl(hb.w)) continue;
                        const int yk = y - ufl(ha.z), xk = x - ufl(hb.x);
                        const int nxk = ufl(hb.y);
                        const bool in = act && yk >= 0 && yk < ufl(ha.w) && xk >= 0 && xk < nxk;
                        if (__ballot(in) == 0ULL) continue;
                        const float *o2 = pool + ufl(hc.x) + (in ? (yk * nxk + xk) : 0) * p.pd;
                        const float *Wz = pool + ufl(hc.z) + p.ph * ufl(hc.w) + p.pw * ufl(hd.x) + qk * p.pd * 4;
                        const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
                        const v4f s4 = zdot(zero4, o2, Wz, p.pd);
                        if (in) acc = acc + s4;
                    }
                }
                if (act) *reinterpret_cast<v4f *>(dst) = acc;
            }
        }
        V3STAMP(8);
        g0 += ng;
    }

    // ---- levels without a zero role: this workgroup stores the zeros of its volume outside the territory; one lane per
    //      run of up to 8 16-byte units of one row
    const v4f z4 = {0.f, 0.f, 0.f, 0.f};
    for (int li = 0; li < p.n_levels; ++li) {
        const V3Level &lv = slev[li];
        if (!lv.merged || p.accum) continue;
        const int cpr = (lv.upr + 7) >> 3;                          // runs per row
        const float inv_cpr = 1.0f / (float)cpr, inv_W = 1.0f / (float)lv.W;
        v4f *base = reinterpret_cast<v4f *>(lv.out + (long long)vol * lv.R * lv.L);
        for (int t = tid; t < lv.R * cpr; t += V3_NT) {
            const int row = (cpr == 1) ? t : fdiv(t, cpr, inv_cpr);
            const int u0 = (t - row * cpr) * 8;
            const int u1 = min(u0 + 8, lv.upr);
            const int y = (lv.W == 1) ? row : fdiv(row, lv.W, inv_W);
            const int x = row - y * lv.W;
            unsigned covered = 0u;                                   // bit i: unit u0 + i lies in the territory
            for (int k = 0; k < cnt; ++k) {
                const int4 q = bb[k];
                if ((q.w & 127) != li) continue;
                if (y < (q.x & 0xffff) || y > (q.x >> 16) || x < (q.y & 0xffff) || x > (q.y >> 16)) continue;
                const int ua = max(((q.z & 0xffff) >> lv.S_shift) * lv.useg, u0);
                const int ub = min((((q.z >> 16) >> lv.S_shift) + 1) * lv.useg, u1);
                if (ub > ua) covered |= ((1u << (ub - ua)) - 1u) << (ua - u0);
            }
            v4f *rowp = base + (long long)row * lv.upr;
#pragma unroll
            for (int i = 0; i < 8; ++i) if (u0 + i < u1 && !((covered >> i) & 1u)) rowp[u0 + i] = z4;
        }
    }
    V3STAMP(9);
}

__global__ __launch_bounds__(V3_NT, V3_NT / 128) void crop_bwd_gather_kernel(V3Params p)
{
    const unsigned bid = blockIdx.x;
    if (bid < p.n_scatter) {
        if (!(p.dbg & 1)) scatter_role(p, bid, 0);
        if (p.dbg & 4) {        // tuning only: the same work again with the instruction cache warm (stamps at [16..])
            __syncthreads();
            scatter_role(p, bid, 16);
        }
        return;
    }
    if (p.dbg & 2) return;
    const unsigned z = bid - p.n_scatter;
    int li = -1;
    unsigned base = 0u;
#pragma unroll
    for (int l = 0; l < V3_MAX_LEVELS; ++l)
        if (l < p.n_levels && !p.lev[l].merged && z >= p.lev[l].zero_base) { li = l; base = p.lev[l].zero_base; }
    if (li >= 0) zero_role(p, li, z - base);
}

inline int ilog2_exact(int v)
{
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

// (rounds 3-4 read launch-geometry knobs MDT_BWD3_* from the environment for the tuning sweeps under tools/; the settled values are
// constants now -- the product library reads no switch from the environment)
inline int env_int(const char *, int dflt) { return dflt; }

inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

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

int resident_per_cu(size_t lds)
{
    static thread_local int cached_lds = -1, cached = 0;      // per thread: the pair is read and written together
    if (cached_lds != (int)lds) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, crop_bwd_gather_kernel, V3_NT, lds) != hipSuccess) n = 0;
        (void)hipGetLastError();
        cached = n; cached_lds = (int)lds;
    }
    return cached;
}

#ifdef MDT_TUNING_HOOKS        // libmdt_hip_tuning.so only (csrc/Makefile): the product library holds no mutable process state
long long *g_v3_ts = nullptr;
int g_v3_dbg = 0, g_v3_dbg_wg = 0;
#endif

}  // namespace

#ifdef MDT_TUNING_HOOKS
// tuning hook (tools/bwd3_probe.py, include/mdt_hip_ab.h): stamp buffer (device, >= 16 int64) or null, role switches, traced workgroup
extern "C" void mdt_debug_bwd3(long long *dev_buf, int dbg, int wg) { g_v3_ts = dev_buf; g_v3_dbg = dbg; g_v3_dbg_wg = wg; }
#endif

namespace mdt_ra {

// Plans and launches the gather-form backward over n_levels maps.  MDT_ERR_UNSUPPORTED: the shape is outside this
// kernel's budgets (the caller falls back to the round-2 territory kernel / the two-kernel form / the ordered kernel).
static int launch_bwd_gather_chunk(int dim, int n_levels, const float *grads, const float *boxes, const int *batch_ix, const int *level,
                                   int N, int B, int C, const int *H, const int *W, const int *D, int ph, int pw, int pd,
                                   float *const *outs, hipStream_t s, int accum);

// More than V3_MAXR RoIs (round 6; before: MDT_ERR_UNSUPPORTED and the ~15x slower exact-order kernel): a series of launches over chunks of
// V3_MAXR RoIs -- the first writes every byte of the maps (zeros included), the later ones run without any zero role and read-modify-write
// the quads their RoIs touch.  Same arithmetic per RoI; the sum over RoIs of different chunks is taken in chunk order (deterministic).
int launch_bwd_gather(int dim, int n_levels, const float *grads, const float *boxes, const int *batch_ix, const int *level,
                      int N, int B, int C, const int *H, const int *W, const int *D, int ph, int pw, int pd,
                      float *const *outs, hipStream_t s)
{
    return launch_bwd_gather_acc(dim, n_levels, grads, boxes, batch_ix, level, N, B, C, H, W, D, ph, pw, pd, outs, s, 0);
}

// accumulate = 1: the maps already hold a gradient (another RoI head's, written by an earlier launch of this kernel): nothing is zero-filled, the
// quads the RoIs touch are read-modify-written.  Two heads that pool the same pyramid then cost ONE full write of the maps instead of two
// plus a dense add (round 6, cuda_functions/_roi_align_impl.PyramidGradAccumulator).
int launch_bwd_gather_acc(int dim, int n_levels, const float *grads, const float *boxes, const int *batch_ix, const int *level,
                          int N, int B, int C, const int *H, const int *W, const int *D, int ph, int pw, int pd,
                          float *const *outs, hipStream_t s, int accumulate)
{
    if (N <= V3_MAXR) return launch_bwd_gather_chunk(dim, n_levels, grads, boxes, batch_ix, level, N, B, C, H, W, D, ph, pw, pd, outs, s, accumulate ? 1 : 0);
    if (dim != 2 && dim != 3) return MDT_ERR_INVALID_ARGUMENT;
    const long long P = (long long)ph * pw * pd;                 // (2D callers pass pd = 1)
    for (int n0 = 0; n0 < N; n0 += V3_MAXR) {
        const int n = (N - n0 < V3_MAXR) ? (N - n0) : V3_MAXR;
        const int rc = launch_bwd_gather_chunk(dim, n_levels, grads + (long long)n0 * C * P, boxes + (long long)n0 * 2 * dim, batch_ix + n0,
                                               level ? level + n0 : nullptr, n, B, C, H, W, D, ph, pw, pd, outs, s, (n0 > 0 || accumulate) ? 1 : 0);
        if (rc != MDT_OK) return (n0 == 0) ? rc : MDT_ERR_LAUNCH_FAILED;      // (a shape outside the budgets fails on the FIRST chunk, before anything was written)
    }
    return MDT_OK;
}

static int launch_bwd_gather_chunk(int dim, int n_levels, const float *grads, const float *boxes, const int *batch_ix, const int *level,
                                   int N, int B, int C, const int *H, const int *W, const int *D, int ph, int pw, int pd,
                                   float *const *outs, hipStream_t s, int accum)
{
    if (dim != 2 && dim != 3) return MDT_ERR_INVALID_ARGUMENT;
    if (n_levels < 1 || n_levels > V3_MAX_LEVELS) return MDT_ERR_UNSUPPORTED;
    if (N < 0 || B <= 0 || C <= 0 || ph <= 0 || pw <= 0 || pd <= 0) return MDT_ERR_INVALID_ARGUMENT;
    if (N > V3_MAXR) return MDT_ERR_UNSUPPORTED;
    V3Params p;
    p.grads = grads; p.boxes = boxes; p.box_ind = batch_ix; p.level = level;
    p.dim = dim; p.N = N; p.B = B; p.C = C; p.n_levels = n_levels; p.accum = accum;
#ifdef MDT_TUNING_HOOKS
    p.ts = g_v3_ts; p.dbg = g_v3_dbg; p.dbg_wg = g_v3_dbg_wg;
#else
    p.ts = nullptr; p.dbg = 0; p.dbg_wg = 0;
#endif
    if (dim == 3) { p.ph = ph; p.pw = pw; p.pd = pd; }
    else { p.ph = ph; p.pw = 1; p.pd = pw; }                   // 2D: (y, -, x)
    if (p.ph > 255 || p.pw > 255 || p.pd > 255) return MDT_ERR_UNSUPPORTED;   // sample ranges are packed in bytes
    p.P = p.ph * p.pw * p.pd;
    p.P4 = (p.P + 3) & ~3;
    p.psum = p.ph + p.pw + p.pd;
    p.inv_psum = 1.0f / (float)p.psum;
    p.inv_pd = 1.0f / (float)p.pd;
    const long long nvol = (long long)B * C;
    if (nvol > 0x3fffffLL) return MDT_ERR_UNSUPPORTED;
    size_t roi_max = 0, zero_lds = 0;          // LDS floats the largest possible RoI of any level needs in one round
    for (int l = 0; l < n_levels; ++l) {
        V3Level &lv = p.lev[l];
        const int Hl = H[l], Wl = (dim == 3) ? W[l] : 1, Dl = (dim == 3) ? D[l] : W[l];
        if (Hl <= 0 || Wl <= 0 || Dl <= 0) return MDT_ERR_INVALID_ARGUMENT;
        if (Hl > 32000 || Wl > 32000 || Dl > 32000) return MDT_ERR_UNSUPPORTED;
        if (Dl % 8 != 0 || (((uintptr_t)outs[l]) & 15) != 0) return MDT_ERR_UNSUPPORTED;     // 16-byte stores, 8-float segments
        const long long vol_floats = (long long)Hl * Wl * Dl;
        if (vol_floats > 0x3fffffLL) return MDT_ERR_UNSUPPORTED;
        lv.out = outs[l]; lv.H = Hl; lv.W = Wl; lv.D = Dl; lv.R = Hl * Wl; lv.L = Dl;
        int seg = 8;
        while (seg < 32 && Dl % (seg * 2) == 0 && ((long long)lv.R * (Dl / seg) > 16384)) seg *= 2;    // keep the bitmap small
        lv.S = seg; lv.S_shift = ilog2_exact(seg); lv.nseg = Dl / seg; lv.useg = seg / 4; lv.useg_shift = ilog2_exact(lv.useg);
        lv.upr = Dl / 4; lv.upr_shift = ilog2_exact(lv.upr);
        const long long nbits = (long long)lv.R * lv.nseg;
        lv.bw = (int)((nbits + 63) / 64);
        lv.merged = (vol_floats * 4 <= V3_MERGE_BYTES) ? 1 : 0;
        lv.zero_parts = 0; lv.rows_per_part = lv.R; lv.zero_base = 0; lv.chunk_rows = 0; lv.chunk_shift = -1; lv.R_shift = -1;
        if (!lv.merged) {
            if (nbits > 65536) return MDT_ERR_UNSUPPORTED;
            const size_t need = (size_t)V3_LEV_BYTES + (size_t)V3_MAXR * 8 * sizeof(short) + 16 + (size_t)B * lv.bw * sizeof(u64);
            if (need > zero_lds) zero_lds = need;
        }
        // a RoI whose index bounding box is the whole map (dense intermediate blocks, see scatter_role's plan)
        const size_t H4 = (size_t)((Hl + 3) & ~3), W4 = (size_t)((Wl + 3) & ~3);
        size_t a = (size_t)Hl * Wl * p.pd;
        if (a < (size_t)p.P4) a = p.P4;
        a = (a + 3) & ~(size_t)3;
        const size_t bsz = ((size_t)Hl * p.pw * p.pd + 3) & ~(size_t)3;
        const size_t wsz = ((size_t)p.ph * H4 + (size_t)p.pw * W4 + (size_t)lv.upr * p.pd * 4 + H4 / 2 + W4 / 2 + 3) & ~(size_t)3;
        if (a + bsz + wsz > roi_max) roi_max = a + bsz + wsz;
    }
    if (zero_lds > (size_t)V3_LDS_CAP) return MDT_ERR_UNSUPPORTED;
    p.n_scatter = (unsigned)nvol;
    // LDS: [levels][pool][hdr][rbox][bb][misc]
    const size_t tail = align16((size_t)V3_GMAX * V3_HDRN * sizeof(int)) + align16((size_t)V3_MAXR * 6 * sizeof(float)) +
                        align16((size_t)V3_MAXR * sizeof(int4)) + 320;
    const size_t lds_cap = (size_t)env_int("MDT_BWD3_LDS_CAP", V3_LDS_CAP);
    if ((size_t)V3_LEV_BYTES + tail + roi_max * sizeof(float) > lds_cap) return MDT_ERR_UNSUPPORTED;
    size_t pool_fl = (lds_cap - (size_t)V3_LEV_BYTES - tail) / sizeof(float);
    pool_fl &= ~(size_t)3;
    p.pool_floats = (int)pool_fl;
    size_t off = (size_t)V3_LEV_BYTES + pool_fl * sizeof(float);
    p.off_hdr = (int)off;  off += align16((size_t)V3_GMAX * V3_HDRN * sizeof(int));
    p.off_rbox = (int)off; off += align16((size_t)V3_MAXR * 6 * sizeof(float));
    p.off_bb = (int)off;   off += align16((size_t)V3_MAXR * sizeof(int4));
    p.off_misc = (int)off; off += 320;
    size_t lds = off;
    if (lds < zero_lds) lds = zero_lds;
    if (lds > lds_cap) return MDT_ERR_UNSUPPORTED;
    {
        static bool optin = false;
        if (!optin) {   // more than 64 KB of dynamic LDS needs the explicit opt-in
            (void)hipFuncSetAttribute((const void *)crop_bwd_gather_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS_CAP);
            (void)hipGetLastError();
            optin = true;
        }
    }
    // zero workgroups: what stays resident beside the scatter workgroups, shared by the big levels in proportion to bytes
    long long big_rows_bytes = 0;
    for (int l = 0; l < n_levels; ++l) if (!p.lev[l].merged) big_rows_bytes += (long long)p.lev[l].R * p.lev[l].L;
    unsigned zrun = 0;
    if (big_rows_bytes > 0 && !accum) {
        const int cus = cu_count();
        long long budget = (long long)resident_per_cu(lds) * cus - (long long)p.n_scatter;
        if (budget < cus / 2) budget = cus;
        const int forced = env_int("MDT_BWD3_ZERO_WGS", 0);
        if (forced > 0) budget = forced;
        for (int l = 0; l < n_levels; ++l) {
            V3Level &lv = p.lev[l];
            if (lv.merged) continue;
            long long z = (long long)((double)budget * (double)((long long)lv.R * lv.L) / (double)big_rows_bytes);
            const long long total_rows = nvol * lv.R;
            if (z < 16) z = 16;
            if (z > total_rows) z = total_rows;
            lv.rows_per_part = (int)((total_rows + z - 1) / z);
            lv.zero_parts = (int)((total_rows + lv.rows_per_part - 1) / lv.rows_per_part);
            const int cr = env_int("MDT_BWD3_ZERO_CHUNK_ROWS", V3_ZERO_CHUNK_ROWS);
            lv.R_shift = ilog2_exact(lv.R);
            lv.chunk_shift = ilog2_exact(cr);
            if (cr > 0 && cr < lv.rows_per_part && lv.chunk_shift >= 0 && lv.R_shift >= 0 && lv.R % cr == 0 && lv.upr_shift >= 0 && lv.useg_shift >= 0 &&
                total_rows < 0x7fffffffLL / lv.upr)
                lv.chunk_rows = cr;
            lv.zero_base = zrun;
            zrun += (unsigned)lv.zero_parts;
        }
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(crop_bwd_gather_kernel, dim3(p.n_scatter + zrun), dim3(V3_NT), lds, s, p);
    return check_launch();
}

}  // namespace mdt_ra
