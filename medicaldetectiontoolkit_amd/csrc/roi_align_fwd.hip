// roi_align_fwd.hip -- 3D RoIAlign ("crop and resize") FORWARD for gfx950, channel-quad form (round 5).
//
// Spec: the reference CUDA kernel cuda_functions/roi_align_3D/roi_align/src/cuda/crop_and_resize_kernel.cu:12-151 (one thread
// per output, 8 scattered 4-byte loads each).  Per-output arithmetic here is that kernel's, term for term (x-lerp, y-lerp, z-lerp;
// this file is compiled with -ffp-contract=off): results are BIT-EXACT against the CPU oracle (tests/test_hip_gpu.py).
//
// Why a third design (rounds 1 and 4 are in roi_align.hip): the direct kernel is bound by cache-line look-ups (64 lanes x 8
// scattered loads), the wave-staged kernel of round 4 by per-wave latency -- one global round trip per (RoI, channel) task with
// nothing in flight behind it, 8 ds_read_b32 + ~60 VALU per output (3.6-10 % of HBM at the inference call sizes).  Here:
//
//   * the sample positions and weights of a RoI are the same for every channel, so a task is (RoI, group of channel QUADS) and the
//     LDS image of the RoI's source box is CHANNEL-INTERLEAVED: [voxel][4 channels] fp32.  A lane owns one output position (y, x, z);
//     ONE ds_read_b128 per corner brings that corner for four channels -- 8 LDS instructions and 7 x 3 VALU per channel give FOUR
//     outputs, the address arithmetic is paid once per four outputs;
//   * staging = per lane one voxel-quad (4 consecutive z, the contiguous axis): 4 x 16-byte global loads (one per channel of the
//     quad: 4 channel planes) -> a 4 x 4 register tile -> 4 x ds_write_b128 ([voxel][c] rows): the transposition costs nothing;
//   * a workgroup walks its stages (output-row chunk x channel quad) through TWO LDS buffers with the next stage's global loads
//     issued into registers BEFORE the current stage is interpolated: the HBM / L2 latency of stage s + 1 hides behind the
//     arithmetic of stage s, one barrier per stage;
//   * the box extents come from the two end samples of every axis (sample coordinates are monotone in the sample index), computed
//     redundantly by every lane right after the box is read: the first stage's loads leave one global round trip after the launch
//     and the sample-table build overlaps them;
//   * source boxes larger than a buffer (24 KB = 1536 voxels x 4 channels) are cut along the OUTPUT rows; a shape for which not even
//     one output row fits (or a map whose z extent is not a multiple of 4) takes the direct path inside the same launch.
//
// HBM-bound gather: no MFMA.  Algorithmic bytes: 4 N C P written + every touched input voxel once + 28 N.
#include <type_traits>
#include "roi_align_common.h"

using namespace mdt_ra;

namespace {

constexpr int CQ_THREADS = 256;
constexpr int CQ_CAP_VQ = 384;                                            // voxel-quads per stage buffer (1536 voxels x 16 B = 24 KB)
constexpr int CQ_ROUNDS = (CQ_CAP_VQ + CQ_THREADS - 1) / CQ_THREADS;      // staging rounds a lane keeps in registers
constexpr int CQ_TAB_MAX = 192;                                           // ch + cw + cd


struct CqParams {
    PyramidMaps maps;
    const float *boxes;
    const int *box_ind;
    const int *level;       // null: every RoI on maps level 0
    float *crops;
    long long *stamps;      // tuning hook (mdt_debug_fwd_stamps): 4 wall-clock stamps per workgroup, or null
    int B, ch, cw, cd, C, nq, groups, qpg;
};

#ifdef MDT_TUNING_HOOKS        // libmdt_hip_tuning.so only (csrc/Makefile): the product library holds no mutable process state
long long *g_fwd_stamps = nullptr;
#endif

// a / d for 0 <= a < 2^24, d >= 1 with a precomputed reciprocal: the float product is off by at most one, fixed up with the remainder
__device__ __forceinline__ int fast_divmod(int a, int d, float rcp, int &rem)
{
    int q = (int)((float)a * rcp);
    int r = a - q * d;
    if (r < 0) { q -= 1; r += d; }
    if (r >= d) { q += 1; r -= d; }
    rem = r;
    return q;
}

struct StageGeom {
    int y0, y1, r0, nrows, fits;
};

template <typename TIN>
__global__ __launch_bounds__(CQ_THREADS) void crop_fwd_cq_kernel(CqParams p)
{
    __shared__ v4f buf[2][CQ_CAP_VQ * 4];
    __shared__ AxisEntry tab[CQ_TAB_MAX];
    // byte offsets of a sample's floor / ceil voxel inside the LDS image (y: relative to the box's first row, x: to its first column,
    // z: to zmin4), so that a corner's address is three additions: .x = floor, .y = ceil
    __shared__ int2 offs[CQ_TAB_MAX];

    const int tid = threadIdx.x;
    if (p.stamps && tid == 0) p.stamps[4 * (long long)blockIdx.x] = wall_clock64();
    const int n = blockIdx.x / p.groups, g = blockIdx.x - n * p.groups;
    // everything the task needs from memory, issued at once (one round trip): level, batch index, box
    const float *bx = p.boxes + (long long)n * 6;
    int l = p.level ? p.level[n] : 0;
    int b_in = p.box_ind[n];
    const float b0 = bx[0], b1 = bx[1], b2 = bx[2], b3 = bx[3], b4 = bx[4], b5 = bx[5];
    const int ch = p.ch, cw = p.cw, cd = p.cd, C = p.C;
    const int P = ch * cw * cd;
    const int q_begin = g * p.qpg, q_end = min(p.nq, q_begin + p.qpg);
    if (q_begin >= q_end) return;
    if (l < 0 || l >= p.maps.n_levels) { l = 0; b_in = -1; }      // no level: the row is zero-filled like a skipped RoI
    const int c_begin = 4 * q_begin, c_end = min(C, 4 * q_end);
    float *out = p.crops + ((long long)n * C + c_begin) * P;
    if (b_in < 0 || b_in >= p.B) {      // skipped RoI: the reference leaves its caller's zero-fill (crop_and_resize_kernel.cu:43-47)
        const int tot = (c_end - c_begin) * P;
        for (int e = tid; e < tot; e += CQ_THREADS) out[e] = 0.0f;
        return;
    }
    const int H = p.maps.H[l], W = p.maps.W[l], D = p.maps.D[l];
    const long long vol = (long long)H * W * D;
    const TIN *image = reinterpret_cast<const TIN *>(p.maps.image[l]) + (long long)b_in * C * vol;

    // the sample table: one lane per sample (the only place the double-precision coordinate arithmetic runs)
    for (int t = tid; t < ch + cw + cd; t += CQ_THREADS) {
        AxisEntry e;
        if (t < ch) e = axis_entry(b0, b2, H, ch, t);
        else if (t < ch + cw) e = axis_entry(b1, b3, W, cw, t - ch);
        else e = axis_entry(b4, b5, D, cd, t - ch - cw);
        tab[t] = e;
    }
    __syncthreads();
    // extents of the touched voxel box: the sample coordinate is monotone in the sample index (either direction: a box may be
    // inverted), so the extremes sit at the two end samples
    const AxisEntry ya = tab[0], yb = tab[ch - 1], xa = tab[ch], xb = tab[ch + cw - 1], za = tab[ch + cw], zb = tab[ch + cw + cd - 1];
    const int ymin = min(ya.lo, yb.lo), ny_all = max(entry_hi(ya), entry_hi(yb)) - ymin + 1;
    const int xmin = min(xa.lo, xb.lo), nx = max(entry_hi(xa), entry_hi(xb)) - xmin + 1;
    const int zlo = min(za.lo, zb.lo), zmin4 = zlo & ~3, nz4 = (max(entry_hi(za), entry_hi(zb)) - zmin4) / 4 + 1;
    const int rowvq = nx * nz4;                       // voxel-quads per source row (one y)
    const int rowf4 = nz4 * 4;                        // v4f slots per (y, x) line of the LDS image
    const int max_rows = CQ_CAP_VQ / rowvq;
    const bool z4 = (D & 3) == 0;
    // staging pays when the samples are dense in the source box: the box is read whole against 8 P scattered loads of the direct form --
    // a (7,7,3) pool over a 16^3-voxel box needs 1176 of its 4096+ voxels and is faster direct (round 4: N = 600 on P2 43.8 us
    // staged-always vs 34.1 us direct; the map is far larger than L2, the staged form then pays fabric bandwidth for voxels nobody reads)
    const bool dense = (long long)ny_all * rowvq * 4 <= 4LL * P;
    // output rows per stage: the k rows [y0, y0 + k) touch at most ceil((k - 1) |sy|) + 3 source rows (sy = source rows per output row)
    int k = 0;
    if (z4 && dense && max_rows >= 2) {
        const float sy = fabsf((b2 - b0) * (float)H / (float)ch);
        k = 1;
        if (max_rows >= 3) {
            const float kk = (float)(max_rows - 3) / fmaxf(sy, 1e-6f);
            k = 1 + (int)fminf(kk, (float)ch);
        }
        if (k > ch) k = ch;
    }

    const int nQ = q_end - q_begin;
    if (k < 1) {
        // ---- direct path for the whole task: 8 scattered loads per output (the round-1 arithmetic)
        const int tot = (c_end - c_begin) * P;
        for (int e = tid; e < tot; e += CQ_THREADS) {
            int idx = e;
            const int z = idx % cd; idx /= cd;
            const int x = idx % cw; idx /= cw;
            const int y = idx % ch;
            const int c = idx / ch;
            const AxisEntry ey = tab[y], ex = tab[ch + x], ez = tab[ch + cw + z];
            const int top = ey.lo, bottom = entry_hi(ey), left = ex.lo, right = entry_hi(ex), front = ez.lo, back = entry_hi(ez);
            const TIN *pc = image + (long long)(c_begin + c) * vol;
            const long long rt_l = (long long)D * (left + (long long)W * top), rt_r = (long long)D * (right + (long long)W * top);
            const long long rb_l = (long long)D * (left + (long long)W * bottom), rb_r = (long long)D * (right + (long long)W * bottom);
            const float tlf = ld(pc, front + rt_l), trf = ld(pc, front + rt_r), blf = ld(pc, front + rb_l), brf = ld(pc, front + rb_r);
            const float tlb = ld(pc, back + rt_l), trb = ld(pc, back + rt_r), blb = ld(pc, back + rb_l), brb = ld(pc, back + rb_r);
            const float top_front = tlf + (trf - tlf) * ex.lerp;
            const float bottom_front = blf + (brf - blf) * ex.lerp;
            const float top_back = tlb + (trb - tlb) * ex.lerp;
            const float bottom_back = blb + (brb - blb) * ex.lerp;
            const float frontv = top_front + (bottom_front - top_front) * ey.lerp;
            const float backv = top_back + (bottom_back - top_back) * ey.lerp;
            out[e] = frontv + (backv - frontv) * ez.lerp;
        }
        if (p.stamps && tid == 0) p.stamps[4 * (long long)blockIdx.x + 3] = wall_clock64();
        return;
    }

    const int nchunks = (ch + k - 1) / k;
    const int rowpos = cw * cd;                         // output positions per output row
    const float rcp_cd = 1.0f / (float)cd, rcp_cw = 1.0f / (float)cw;

    // a STAGE = the source rows of one output-row chunk for `nqs` consecutive channel quads (as many as fit a buffer: small boxes bring
    // several quads per barrier and fill the lanes of the interpolation loop)
    auto geom = [&](int chunk) {
        StageGeom gm;
        gm.y0 = chunk * k;
        gm.y1 = min(ch, gm.y0 + k);
        const AxisEntry ea = tab[gm.y0], eb = tab[gm.y1 - 1];
        gm.r0 = min(ea.lo, eb.lo);
        gm.nrows = max(entry_hi(ea), entry_hi(eb)) - gm.r0 + 1;
        gm.fits = gm.nrows <= max_rows;                  // (guaranteed by the choice of k up to float rounding; guarded)
        return gm;
    };
    auto quads_per_stage = [&](const StageGeom &gm) {
        if (!gm.fits) return 1;
        const int nvq = gm.nrows * rowvq;
        return max(1, min(nQ, CQ_CAP_VQ / nvq));
    };

    // per staging round of this lane: which quad of the stage and which element offset inside a channel plane -- depends on the chunk only
    long long gbase[CQ_ROUNDS];
    int gqs[CQ_ROUNDS];
    auto lane_bases = [&](const StageGeom &gm, int QS) {
        const int nvq = gm.fits ? gm.nrows * rowvq : 0;
#pragma unroll
        for (int r = 0; r < CQ_ROUNDS; ++r) {
            const int sl = tid + r * CQ_THREADS;
            gbase[r] = -1;
            gqs[r] = 0;
            if (sl < QS * nvq) {
                const int qs = sl / nvq, vq = sl - qs * nvq;
                const int row = vq / nz4, zq = vq - row * nz4;
                const int ry = row / nx, rx = row - ry * nx;
                gbase[r] = (long long)D * ((xmin + rx) + (long long)W * (gm.r0 + ry)) + zmin4 + 4 * zq;
                gqs[r] = qs;
            }
        }
    };
    float v[CQ_ROUNDS][4][4];
    auto prefetch = [&](int q, int nqs) {
#pragma unroll
        for (int r = 0; r < CQ_ROUNDS; ++r) {
            if (gbase[r] >= 0 && gqs[r] < nqs) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cj = min(4 * (q + gqs[r]) + j, C - 1);      // (a partial last quad re-reads channel C - 1; never stored)
                    ld4(image + (long long)cj * vol, gbase[r], v[r][j]);
                }
            }
        }
    };
    auto stage_write = [&](int nqs, v4f *dst) {
#pragma unroll
        for (int r = 0; r < CQ_ROUNDS; ++r) {
            if (gbase[r] >= 0 && gqs[r] < nqs) {
                const int sl = tid + r * CQ_THREADS;
#pragma unroll
                for (int zz = 0; zz < 4; ++zz) dst[sl * 4 + zz] = v4f{v[r][0][zz], v[r][1][zz], v[r][2][zz], v[r][3][zz]};
            }
        }
    };

    int chunk = 0, qpos = 0;                       // the stage being interpolated: chunk, first quad (relative to q_begin)
    StageGeom cur = geom(0);
    int QS = quads_per_stage(cur);
    lane_bases(cur, QS);
    if (p.stamps && tid == 0) p.stamps[4 * (long long)blockIdx.x + 1] = wall_clock64();
    if (cur.fits) prefetch(q_begin, min(QS, nQ));
    // the offset table is read by the interpolation only, i.e. behind the first barrier of the loop below
    for (int t = tid; t < ch + cw + cd; t += CQ_THREADS) {
        const AxisEntry e = tab[t];
        int2 o;
        if (t < ch) { o.x = (e.lo - ymin) * nx * rowf4 * 16; o.y = (entry_hi(e) - ymin) * nx * rowf4 * 16; }
        else if (t < ch + cw) { o.x = (e.lo - xmin) * rowf4 * 16; o.y = (entry_hi(e) - xmin) * rowf4 * 16; }
        else { o.x = (e.lo - zmin4) * 16; o.y = (entry_hi(e) - zmin4) * 16; }
        offs[t] = o;
    }
    // Two LDS buffers: stage s is interpolated from buf[s & 1] while the loads of stage s + 1 are in flight into registers; one barrier
    // per stage.  (Tried instead, tools/fwd_stamp_probe.py: ONE buffer, no register prefetch, five workgroups per CU -- the launch got
    // slower, 30 -> 49 us at N = 240: the time from "loads issued" to "stage in LDS" grows from 1.9 to 8 us as soon as more workgroups
    // gather at once; the kernel is bound by the memory pipeline's rate of partial-line requests, not by unhidden latency.)
    for (int s = 0;; ++s) {
        const int nqs = min(QS, nQ - qpos);
        const int q = q_begin + qpos;
        v4f *src = buf[s & 1];
        if (cur.fits) stage_write(nqs, src);
        __syncthreads();
        if (p.stamps && tid == 0 && s == 0) p.stamps[4 * (long long)blockIdx.x + 2] = wall_clock64();
        // the next stage: same chunk, next quads -- or the next chunk
        int chunk1 = chunk, qpos1 = qpos + nqs;
        StageGeom nxt = cur;
        int QS1 = QS;
        if (qpos1 >= nQ) { chunk1 = chunk + 1; qpos1 = 0; }
        const bool more = chunk1 < nchunks;
        if (more) {
            if (chunk1 != chunk) { nxt = geom(chunk1); QS1 = quads_per_stage(nxt); lane_bases(nxt, QS1); }
            if (nxt.fits) prefetch(q_begin + qpos1, min(QS1, nQ - qpos1));
        }
        // ---- interpolate the stage: positions (quad of the stage, output rows [y0, y1)), four channels each
        const int pc = (cur.y1 - cur.y0) * rowpos;          // positions per quad
        const int e0 = cur.y0 * rowpos;
        const int nvq = cur.nrows * rowvq;
        if (cur.fits) {
            // the chunk's image starts at source row r0: fold that into the buffer address once
            const char *img0 = reinterpret_cast<const char *>(src) - (long long)(cur.r0 - ymin) * nx * rowf4 * 16;
            const float rcp_pc = 1.0f / (float)pc;
            for (int pe = tid; pe < nqs * pc; pe += CQ_THREADS) {
                int el, z, x;
                const int qs = fast_divmod(pe, pc, rcp_pc, el);
                const int t = fast_divmod(el, cd, rcp_cd, z);
                const int y = cur.y0 + fast_divmod(t, cw, rcp_cw, x);
                const int e = e0 + el;
                const float ly = tab[y].lerp, lx = tab[ch + x].lerp, lz = tab[ch + cw + z].lerp;
                const int2 oy = offs[y], ox = offs[ch + x], oz = offs[ch + cw + z];
                const char *img = img0 + (long long)qs * nvq * 64;
                const char *rt = img + oy.x, *rb = img + oy.y;
                const char *a = rt + ox.x, *bq = rt + ox.y, *cq = rb + ox.x, *dq = rb + ox.y;
                const v4f tlf = *reinterpret_cast<const v4f *>(a + oz.x), trf = *reinterpret_cast<const v4f *>(bq + oz.x);
                const v4f blf = *reinterpret_cast<const v4f *>(cq + oz.x), brf = *reinterpret_cast<const v4f *>(dq + oz.x);
                const v4f tlb = *reinterpret_cast<const v4f *>(a + oz.y), trb = *reinterpret_cast<const v4f *>(bq + oz.y);
                const v4f blb = *reinterpret_cast<const v4f *>(cq + oz.y), brb = *reinterpret_cast<const v4f *>(dq + oz.y);
                // the reference's order, on four channels at a time (element-wise IEEE fp32 operations; no contraction in this file)
                const v4f top_front = tlf + (trf - tlf) * lx;
                const v4f bottom_front = blf + (brf - blf) * lx;
                const v4f top_back = tlb + (trb - tlb) * lx;
                const v4f bottom_back = blb + (brb - blb) * lx;
                const v4f frontv = top_front + (bottom_front - top_front) * ly;
                const v4f backv = top_back + (bottom_back - top_back) * ly;
                const v4f res = frontv + (backv - frontv) * lz;
                const int cq0 = 4 * (q + qs);
                float *oq = out + (long long)(cq0 - c_begin) * P + e;
                const int nvalid = C - cq0;
                oq[0] = res.x;
                if (nvalid > 1) oq[P] = res.y;
                if (nvalid > 2) oq[2LL * P] = res.z;
                if (nvalid > 3) oq[3LL * P] = res.w;
            }
        } else {
            for (int pe = tid; pe < pc; pe += CQ_THREADS) {      // (a chunk that does not fit is walked one quad per stage)
                int z, x;
                const int t = fast_divmod(pe, cd, rcp_cd, z);
                const int y = cur.y0 + fast_divmod(t, cw, rcp_cw, x);
                const int e = e0 + pe;
                const AxisEntry ey = tab[y], ex = tab[ch + x], ez = tab[ch + cw + z];
                const int top = ey.lo, bottom = entry_hi(ey), left = ex.lo, right = entry_hi(ex), front = ez.lo, back = entry_hi(ez);
                const long long rt_l = (long long)D * (left + (long long)W * top), rt_r = (long long)D * (right + (long long)W * top);
                const long long rb_l = (long long)D * (left + (long long)W * bottom), rb_r = (long long)D * (right + (long long)W * bottom);
                const int nvalid = min(4, C - 4 * q);
                for (int j = 0; j < nvalid; ++j) {
                    const TIN *pcn = image + (long long)(4 * q + j) * vol;
                    const float tlf = ld(pcn, front + rt_l), trf = ld(pcn, front + rt_r), blf = ld(pcn, front + rb_l), brf = ld(pcn, front + rb_r);
                    const float tlb = ld(pcn, back + rt_l), trb = ld(pcn, back + rt_r), blb = ld(pcn, back + rb_l), brb = ld(pcn, back + rb_r);
                    const float top_front = tlf + (trf - tlf) * ex.lerp;
                    const float bottom_front = blf + (brf - blf) * ex.lerp;
                    const float top_back = tlb + (trb - tlb) * ex.lerp;
                    const float bottom_back = blb + (brb - blb) * ex.lerp;
                    const float frontv = top_front + (bottom_front - top_front) * ey.lerp;
                    const float backv = top_back + (bottom_back - top_back) * ey.lerp;
                    out[(long long)(4 * q + j - c_begin) * P + e] = frontv + (backv - frontv) * ez.lerp;
                }
            }
        }
        if (!more) break;
        cur = nxt; QS = QS1; chunk = chunk1; qpos = qpos1;
    }
    if (p.stamps && tid == 0) p.stamps[4 * (long long)blockIdx.x + 3] = wall_clock64();
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Forward on CHANNELS-LAST maps [B, Y, X, Z, C] (round 5): the layout the convolution path produces.  A corner voxel is C contiguous floats
// (144 bytes at C = 36) serving every channel, so a lane owns (output position, channel quad): 8 corner loads of 16 bytes straight from
// global memory -- the 9 quad-lanes of a position read one contiguous 144-byte run, a wave touches ~7-14 lines per load instead of 64 -- no
// LDS image at all.  Removes the row-major copy of the pyramid per forward and its twin in the backward (2 x 110 us on P2 in the training
// step).  Output in the reference layout [N, C, ch, cw, cd]; arithmetic per output = the reference's (bit-equal to the row-major kernels).
struct ClParams {
    PyramidMaps maps;
    const float *boxes;
    const int *box_ind;
    const int *level;
    float *crops;
    int B, ch, cw, cd, C, Q, ppb, nsplit;
};

struct bf16x4raw { unsigned short v[4]; };
__device__ __forceinline__ v4f ldq(const float *p, long long i) { return *reinterpret_cast<const v4f *>(p + i); }
__device__ __forceinline__ v4f ldq(const bf16raw *p, long long i)
{
    const uint2 q = *reinterpret_cast<const uint2 *>(p + i);
    return v4f{__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u)};
}

constexpr int CL_PASSES = 4;            // passes of ppb positions a workgroup gathers in LDS before it stores them channel by channel
constexpr int CL_TILE_MAX = 8192;       // floats of the output tile (C x positions per chunk, +1 padding per channel row)

template <typename TIN>
__global__ __launch_bounds__(CQ_THREADS) void crop_fwd_cl_kernel(ClParams p)
{
    __shared__ AxisEntry tab[CQ_TAB_MAX];
    __shared__ float tile[CL_TILE_MAX];
    const int tid = threadIdx.x;
    const int n = blockIdx.x;
    const float *bx = p.boxes + (long long)n * 6;
    int l = p.level ? p.level[n] : 0;
    int b_in = p.box_ind[n];
    const float b0 = bx[0], b1 = bx[1], b2 = bx[2], b3 = bx[3], b4 = bx[4], b5 = bx[5];
    const int ch = p.ch, cw = p.cw, cd = p.cd, C = p.C, Q = p.Q;
    const int P = ch * cw * cd;
    if (l < 0 || l >= p.maps.n_levels) { l = 0; b_in = -1; }
    float *out = p.crops + (long long)n * C * P;
    if (b_in < 0 || b_in >= p.B) {
        for (int e = blockIdx.y * CQ_THREADS + tid; e < C * P; e += p.nsplit * CQ_THREADS) out[e] = 0.0f;
        return;
    }
    const int H = p.maps.H[l], W = p.maps.W[l], D = p.maps.D[l];
    const TIN *image = reinterpret_cast<const TIN *>(p.maps.image[l]) + (long long)b_in * H * W * D * C;
    for (int t = tid; t < ch + cw + cd; t += CQ_THREADS) {
        AxisEntry e;
        if (t < ch) e = axis_entry(b0, b2, H, ch, t);
        else if (t < ch + cw) e = axis_entry(b1, b3, W, cw, t - ch);
        else e = axis_entry(b4, b5, D, cd, t - ch - cw);
        tab[t] = e;
    }
    __syncthreads();
    const int pl = tid / Q, q = tid - pl * Q;
    const float rcp_cd = 1.0f / (float)cd, rcp_cw = 1.0f / (float)cw;
    const int chunk_pos = p.ppb * CL_PASSES;                 // positions per chunk
    const int trow = chunk_pos + 1;                          // tile row stride (odd: the transposed writes spread over the banks)
    const int nchunks = (P + chunk_pos - 1) / chunk_pos;
    for (int ck = blockIdx.y; ck < nchunks; ck += p.nsplit) {
        const int pos0 = ck * chunk_pos;
        const int npos = min(chunk_pos, P - pos0);
        if (pl < p.ppb) {
#pragma unroll
            for (int ps = 0; ps < CL_PASSES; ++ps) {
                const int lp = ps * p.ppb + pl;              // position inside the chunk
                if (lp >= npos) break;
                const int pos = pos0 + lp;
                int z, x;
                const int t = fast_divmod(pos, cd, rcp_cd, z);
                const int y = fast_divmod(t, cw, rcp_cw, x);
                const AxisEntry ey = tab[y], ex = tab[ch + x], ez = tab[ch + cw + z];
                const int top = ey.lo, bottom = entry_hi(ey), left = ex.lo, right = entry_hi(ex), front = ez.lo, back = entry_hi(ez);
                const long long rt_l = ((long long)top * W + left) * D, rt_r = ((long long)top * W + right) * D;
                const long long rb_l = ((long long)bottom * W + left) * D, rb_r = ((long long)bottom * W + right) * D;
                const int c4 = 4 * q;
                const v4f tlf = ldq(image, (rt_l + front) * C + c4), trf = ldq(image, (rt_r + front) * C + c4);
                const v4f blf = ldq(image, (rb_l + front) * C + c4), brf = ldq(image, (rb_r + front) * C + c4);
                const v4f tlb = ldq(image, (rt_l + back) * C + c4), trb = ldq(image, (rt_r + back) * C + c4);
                const v4f blb = ldq(image, (rb_l + back) * C + c4), brb = ldq(image, (rb_r + back) * C + c4);
                const float lx = ex.lerp, ly = ey.lerp, lz = ez.lerp;
                const v4f top_front = tlf + (trf - tlf) * lx;
                const v4f bottom_front = blf + (brf - blf) * lx;
                const v4f top_back = tlb + (trb - tlb) * lx;
                const v4f bottom_back = blb + (brb - blb) * lx;
                const v4f frontv = top_front + (bottom_front - top_front) * ly;
                const v4f backv = top_back + (bottom_back - top_back) * ly;
                const v4f res = frontv + (backv - frontv) * lz;
                float *tq = tile + c4 * trow + lp;
                tq[0] = res.x; tq[trow] = res.y; tq[2 * trow] = res.z; tq[3 * trow] = res.w;
            }
        }
        __syncthreads();
        // the chunk leaves channel by channel: runs of `npos` consecutive positions of out[n][c][pos0 ..]
        for (int e = tid; e < C * npos; e += CQ_THREADS) {
            const int c = e / npos, lp = e - c * npos;
            out[(long long)c * P + pos0 + lp] = tile[c * trow + lp];
        }
        __syncthreads();
    }
}

}  // namespace

namespace mdt_ra {

template <typename TIN>
int launch_fwd_cq(const PyramidMaps &maps, const float *boxes, const int *box_ind, const int *level, int N, int B,
                  int ch, int cw, int cd, int C, float *crops, hipStream_t s)
{
    if (N <= 0) return MDT_OK;
    if (ch + cw + cd > CQ_TAB_MAX || (long long)C * ch * cw * cd > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    for (int l = 0; l < maps.n_levels; ++l) {
        if ((reinterpret_cast<uintptr_t>(maps.image[l]) & 15) != 0) return MDT_ERR_UNSUPPORTED;
        if ((long long)maps.H[l] * maps.W[l] * maps.D[l] > 0x3fffffffLL) return MDT_ERR_UNSUPPORTED;
    }
    CqParams p;
    p.maps = maps; p.boxes = boxes; p.box_ind = box_ind; p.level = level; p.crops = crops;
#ifdef MDT_TUNING_HOOKS
    p.stamps = g_fwd_stamps;
#else
    p.stamps = nullptr;
#endif
    p.B = B; p.ch = ch; p.cw = cw; p.cd = cd; p.C = C;
    p.nq = (C + 3) / 4;
    // enough workgroups to give every CU three at a time (the pipeline inside a workgroup hides its own latency, neighbours hide the
    // prologue) -- and, for the large pools, SMALL tasks: a task's cost grows with the RoI's source box (more output-row chunks), and the
    // longest task bounds the launch (tools/fwd_stamp_probe.py: median workgroup 16 us, longest 31 us with three quads per task)
    int groups = (768 + N - 1) / N;
    if ((long long)ch * cw * cd >= 512) groups = (C + 3) / 4;
    if (groups < 1) groups = 1;
    if (groups > p.nq) groups = p.nq;
    p.qpg = (p.nq + groups - 1) / groups;
    p.groups = (p.nq + p.qpg - 1) / p.qpg;
    const long long grid = (long long)N * p.groups;
    if (grid > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    (void)hipGetLastError();
    hipLaunchKernelGGL((crop_fwd_cq_kernel<TIN>), dim3((unsigned)grid), dim3(CQ_THREADS), 0, s, p);
    return check_launch();
}

template <typename TIN>
int launch_fwd_cl(const PyramidMaps &maps, const float *boxes, const int *box_ind, const int *level, int N, int B,
                  int ch, int cw, int cd, int C, float *crops, hipStream_t s)
{
    if (N <= 0) return MDT_OK;
    if (C % 4 != 0 || C / 4 > CQ_THREADS || ch + cw + cd > CQ_TAB_MAX || (long long)C * ch * cw * cd > 0x7fffffffLL) return MDT_ERR_UNSUPPORTED;
    for (int l = 0; l < maps.n_levels; ++l)
        if ((reinterpret_cast<uintptr_t>(maps.image[l]) & 15) != 0) return MDT_ERR_UNSUPPORTED;
    ClParams p;
    p.maps = maps; p.boxes = boxes; p.box_ind = box_ind; p.level = level; p.crops = crops;
    p.B = B; p.ch = ch; p.cw = cw; p.cd = cd; p.C = C; p.Q = C / 4;
    p.ppb = CQ_THREADS / p.Q;
    if ((long long)C * (p.ppb * CL_PASSES + 1) > CL_TILE_MAX) return MDT_ERR_UNSUPPORTED;
    const int P = ch * cw * cd;
    const int npass = (P + p.ppb * CL_PASSES - 1) / (p.ppb * CL_PASSES);
    int nsplit = (4096 + N - 1) / N;             // one chunk per workgroup while the grid stays below ~16 workgroups per CU
    if (nsplit > npass) nsplit = npass;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 65535) nsplit = 65535;
    p.nsplit = nsplit;
    (void)hipGetLastError();
    hipLaunchKernelGGL((crop_fwd_cl_kernel<TIN>), dim3((unsigned)N, (unsigned)nsplit), dim3(CQ_THREADS), 0, s, p);
    return check_launch();
}
template int launch_fwd_cl<float>(const PyramidMaps &, const float *, const int *, const int *, int, int, int, int, int, int, float *, hipStream_t);
template int launch_fwd_cl<bf16raw>(const PyramidMaps &, const float *, const int *, const int *, int, int, int, int, int, int, float *, hipStream_t);

template int launch_fwd_cq<float>(const PyramidMaps &, const float *, const int *, const int *, int, int, int, int, int, int, float *, hipStream_t);
template int launch_fwd_cq<bf16raw>(const PyramidMaps &, const float *, const int *, const int *, int, int, int, int, int, int, float *, hipStream_t);
template int launch_fwd_cq<u8raw>(const PyramidMaps &, const float *, const int *, const int *, int, int, int, int, int, int, float *, hipStream_t);

}  // namespace mdt_ra

#ifdef MDT_TUNING_HOOKS
extern "C" void mdt_debug_fwd_stamps(long long *dev_buf) { g_fwd_stamps = dev_buf; }
#endif

extern "C" int mdt_pyramid_roi_align_forward_cl(int n_levels, const void *const *images, int bf16, const int *H, const int *W, const int *D,
                                                const float *boxes, const int *batch_ix, const int *level, int num_boxes, int batch, int depth,
                                                int ch, int cw, int cd, float *crops, void *stream)
{
    if (n_levels < 1 || n_levels > mdt_ra::PYR_MAX_LEVELS || num_boxes < 0 || batch <= 0 || ch <= 0 || cw <= 0 || cd <= 0 || depth <= 0)
        return MDT_ERR_INVALID_ARGUMENT;
    mdt_ra::PyramidMaps maps;
    maps.n_levels = n_levels;
    for (int l = 0; l < n_levels; ++l) {
        if (H[l] <= 0 || W[l] <= 0 || D[l] <= 0 || images[l] == nullptr) return MDT_ERR_INVALID_ARGUMENT;
        if ((long long)H[l] * W[l] * D[l] * depth > 0x3fffffffLL * 4) return MDT_ERR_UNSUPPORTED;
        maps.image[l] = images[l]; maps.H[l] = H[l]; maps.W[l] = W[l]; maps.D[l] = D[l];
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    return bf16 ? mdt_ra::launch_fwd_cl<mdt_ra::bf16raw>(maps, boxes, batch_ix, level, num_boxes, batch, ch, cw, cd, depth, crops, s)
                : mdt_ra::launch_fwd_cl<float>(maps, boxes, batch_ix, level, num_boxes, batch, ch, cw, cd, depth, crops, s);
}
