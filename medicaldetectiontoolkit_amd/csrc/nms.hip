// nms.hip -- 2D/3D greedy IoU-NMS, fully device resident, for gfx950.
//
// Semantics follow the reference (paths relative to the reference checkout):
//   IoU (+1 pixel convention)  cuda_functions/nms_3D/src/cuda/nms_kernel.cu:16-28, nms_2D/...:16-24
//   pairwise mask              cuda_functions/nms_3D/src/cuda/nms_kernel.cu:30-78
//   greedy scan                cuda_functions/nms_3D/src/nms_cuda.c:47-61   (runs on the HOST there,
//                              after a D2H copy of the whole mask, :33-34)
// Design here:
//   * mask kernel: one wavefront per 64x64 block of (row, col) box pairs, four
//     blocks per workgroup.  The 64 lanes hold the 64 column boxes in registers;
//     the row box is broadcast with v_readlane and one __ballot per row IS the
//     u64 mask word.  Only blocks on/above the diagonal are computed.
//   * scan kernel: one workgroup per problem.  Wave 0 resolves a 64-box block
//     with a parallel, ballot-based greedy pass over the block's symmetric
//     overlap words; all waves then OR the kept rows' remaining mask words
//     (prefetched one block ahead) into the `removed` bitmap held in LDS.  An
//     optional max_keep stops the scan early (the RPN keeps only the first
//     75-500 boxes, models/mrcnn.py:348).
//   Nothing is copied to the host and no memory is allocated.
// Compiled with -ffp-contract=off so the IoU rounds like the uncontracted oracle
// (the denominator Sa + Sb - interS would otherwise be contracted).

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "mdt_hip.h"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ float bcast(float v, int src_lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// STRIDE == 7: 3D rows (c0,c1,c2,c3,c4,c5,score); STRIDE == 5: 2D rows (c0,c1,c2,c3,score)
template <int STRIDE>
struct Box {
    float c[STRIDE - 1];
};

template <int STRIDE>
__device__ __forceinline__ float box_iou(const Box<STRIDE> &a, const Box<STRIDE> &b)
{
    const float left = fmaxf(a.c[0], b.c[0]), right = fminf(a.c[2], b.c[2]);
    const float top = fmaxf(a.c[1], b.c[1]), bottom = fminf(a.c[3], b.c[3]);
    const float width = fmaxf(right - left + 1.0f, 0.f);
    const float height = fmaxf(bottom - top + 1.0f, 0.f);
    if (STRIDE == 7) {
        const float front = fmaxf(a.c[4], b.c[4]), back = fminf(a.c[5], b.c[5]);
        const float depth = fmaxf(back - front + 1.0f, 0.f);
        const float interS = width * height * depth;
        const float Sa = (a.c[2] - a.c[0] + 1.0f) * (a.c[3] - a.c[1] + 1.0f) * (a.c[5] - a.c[4] + 1.0f);
        const float Sb = (b.c[2] - b.c[0] + 1.0f) * (b.c[3] - b.c[1] + 1.0f) * (b.c[5] - b.c[4] + 1.0f);
        return interS / (Sa + Sb - interS);
    } else {
        const float interS = width * height;
        const float Sa = (a.c[2] - a.c[0] + 1.0f) * (a.c[3] - a.c[1] + 1.0f);
        const float Sb = (b.c[2] - b.c[0] + 1.0f) * (b.c[3] - b.c[1] + 1.0f);
        return interS / (Sa + Sb - interS);
    }
}

constexpr int MASK_WAVES = 4;

// grid: (ceil(col_blocks / MASK_WAVES), col_blocks, batch); block: 64 * MASK_WAVES
template <int STRIDE>
__global__ __launch_bounds__(64 * MASK_WAVES) void nms_mask_kernel(
    const float *__restrict__ dets, int n, float thresh, int rule, int fill_lower, int block_major,
    u64 *__restrict__ mask, int row_block0, const int *__restrict__ done, long long mask_stride_bytes)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int col_blocks = (n + 63) / 64;
    const int col_start = blockIdx.x * MASK_WAVES + wave;
    const int row_start = row_block0 + blockIdx.y;          // (lazy form: the launch covers the row blocks [row_block0, row_block0 + gridDim.y))
    if (col_start >= col_blocks) return;
    if (done != nullptr && done[(mask_stride_bytes >> 2) * blockIdx.z] != 0) return;    // this problem's scan already has its max_keep boxes
    dets += (long long)blockIdx.z * n * STRIDE;
    // block_major (internal workspace): word of row 64*R + l for column block Cb sits at
    // ((R * col_blocks + Cb) * 64 + l): a wave reads/writes 512 contiguous bytes.
    if (mask_stride_bytes > 0) mask = reinterpret_cast<u64 *>(reinterpret_cast<char *>(mask) + (long long)blockIdx.z * mask_stride_bytes);
    else mask += (long long)blockIdx.z * (block_major ? (long long)col_blocks * col_blocks * 64 : (long long)n * col_blocks);

    const int row_idx = row_start * 64 + lane;
    // fill_lower: 0 = leave the blocks below the diagonal untouched, 1 = write them as zero, 2 = compute them like the
    // reference kernel does (its early-out is commented out, nms_kernel.cu:35; its host scan never reads them, nms_cuda.c:54)
    const bool lower_block = row_start > col_start;
    if (lower_block && fill_lower != 2) {
        if (fill_lower && row_idx < n) mask[(long long)row_idx * col_blocks + col_start] = 0ULL;
        return;
    }
    const int col_idx = col_start * 64 + lane;
    const int row_size = min(n - row_start * 64, 64);

    Box<STRIDE> colb, rowb;
#pragma unroll
    for (int q = 0; q < STRIDE - 1; ++q) {
        colb.c[q] = (col_idx < n) ? dets[(long long)col_idx * STRIDE + q] : 0.0f;
        rowb.c[q] = (row_idx < n) ? dets[(long long)row_idx * STRIDE + q] : 0.0f;
    }

    u64 word = 0ULL;
    for (int i = 0; i < row_size; ++i) {
        Box<STRIDE> a;
#pragma unroll
        for (int q = 0; q < STRIDE - 1; ++q) a.c[q] = bcast(rowb.c[q], i);
        const float v = box_iou<STRIDE>(a, colb);  // a = current (row) box, b = column box
        bool pred = (rule == MDT_NMS_RULE_GT) ? (v > thresh) : (v >= thresh);
        // public (row-major) mask: strictly-upper bits only, like the reference.  Internal block-major mask:
        // diagonal blocks hold the SYMMETRIC relation (the IoU is bitwise symmetric), which lets the scan kernel
        // resolve a block with ballots instead of a 64-step serial loop.
        const int row_g = row_start * 64 + i;
        pred = pred && (col_idx < n) && (block_major ? (col_idx != row_g) : (lower_block || col_idx > row_g));
        const u64 bal = __ballot(pred);
        if (lane == i) word = bal;
    }
    if (block_major) mask[((long long)row_start * col_blocks + col_start) * 64 + lane] = (row_idx < n) ? word : 0ULL;
    else if (row_idx < n) mask[(long long)row_idx * col_blocks + col_start] = word;
}

constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_WAVES = SCAN_THREADS / 64;
constexpr int SCAN_PF = 6;   // column words prefetched per lane before the resolve (covers n <= 6208 in one round)


// grid: batch; block: SCAN_THREADS; dynamic LDS: col_blocks * 8 bytes.  mask is block-major with symmetric
// diagonal blocks.
//
// Per 64-box row block k:
//   * the words this workgroup will need (diagonal + the later column blocks of row block k) were prefetched
//     one iteration earlier -- they depend on nothing -- so their latency hides behind block k-1;
//   * wave 0 resolves the block with a PARALLEL greedy pass: lane i holds the set of earlier boxes of the block
//     that overlap it; each round, a still-undecided box is removed if a kept earlier box overlaps it, kept if no
//     undecided earlier box overlaps it (two ballots per round; the fixed point is exactly the sequential greedy
//     result, reached in as many rounds as the longest dependency chain, typically 2-6);
//   * all 16 waves OR the kept rows' later words into the LDS `removed` bitmap (ds_or_b64).
struct ScanRow {
    u64 pre[SCAN_PF];
    u64 diag;
};

__device__ __forceinline__ ScanRow scan_load(const u64 *__restrict__ mask, int k, int col_blocks, int wave, int lane)
{
    ScanRow r;
    const u64 *rowblk = mask + (long long)k * col_blocks * 64;
#pragma unroll
    for (int q = 0; q < SCAN_PF; ++q) {
        const int j = k + 1 + wave + SCAN_WAVES * q;
        r.pre[q] = (j < col_blocks) ? rowblk[(long long)j * 64 + lane] : 0ULL;
    }
    r.diag = (wave == 0) ? rowblk[(long long)k * 64 + lane] : 0ULL;
    return r;
}

// Lazy form (kb, ke, state != null): the scan covers the row blocks [kb, ke) only and keeps its state -- `removed` bitmap, number kept, a
// done flag -- in global memory between launches: the RPN wants the first 75 boxes of 6000 (mrcnn.py:348), which the scan usually has
// after a few hundred rows, so the mask is built and scanned in growing chunks and the later launches return at once.
// state (ints): [0] done, [1] nkept, [2..3] pad, then col_blocks u64 of `removed`.
__global__ __launch_bounds__(SCAN_THREADS) void nms_scan_kernel(
    const u64 *__restrict__ mask, int n, int max_keep,
    long long *__restrict__ keep, int keep_stride, int *__restrict__ num_out, int kb, int ke, int *__restrict__ state, long long ws_stride_bytes)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    u64 *remv = reinterpret_cast<u64 *>(smem_raw);
    __shared__ u64 s_kept;
    __shared__ int s_nkept;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int col_blocks = (n + 63) / 64;
    if (state != nullptr) {
        mask = reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(mask) + (long long)blockIdx.x * ws_stride_bytes);
        state = reinterpret_cast<int *>(reinterpret_cast<char *>(state) + (long long)blockIdx.x * ws_stride_bytes);
        if (kb > 0 && state[0] != 0) return;           // finished in an earlier chunk
    } else {
        mask += (long long)blockIdx.x * col_blocks * col_blocks * 64;
        kb = 0; ke = col_blocks;
    }
    keep += (long long)blockIdx.x * keep_stride;
    const int limit = (max_keep > 0) ? min(max_keep, keep_stride) : keep_stride;
    u64 *g_remv = state ? reinterpret_cast<u64 *>(state + 4) : nullptr;

    int nkept = 0;
    if (state != nullptr && kb > 0) {
        for (int j = tid; j < col_blocks; j += SCAN_THREADS) remv[j] = g_remv[j];
        nkept = state[1];
    } else {
        for (int j = tid; j < col_blocks; j += SCAN_THREADS) remv[j] = 0ULL;
    }
    if (tid == 0) { s_kept = 0ULL; s_nkept = nkept; }
    ScanRow cur = scan_load(mask, kb, col_blocks, wave, lane);
    __syncthreads();

    bool full = false;
    for (int k = kb; k < ke; ++k) {
        ScanRow nxt;
        // prefetch the next row block unless it is ALREADY fully removed (bits only ever get set, so it stays dead)
        bool want_next = false;
        if (k + 1 < ke) {
            const int rows_next = min(n - (k + 1) * 64, 64);
            const u64 valid_next = (rows_next == 64) ? ~0ULL : ((1ULL << rows_next) - 1ULL);
            want_next = (~remv[k + 1] & valid_next) != 0ULL;
        }
        if (want_next) nxt = scan_load(mask, k + 1, col_blocks, wave, lane);
        else {
#pragma unroll
            for (int q = 0; q < SCAN_PF; ++q) nxt.pre[q] = 0ULL;
            nxt.diag = 0ULL;
        }
        const int rows_here = min(n - k * 64, 64);
        const u64 valid = (rows_here == 64) ? ~0ULL : ((1ULL << rows_here) - 1ULL);
        const u64 r = remv[k];
        if ((~r & valid) == 0ULL) { cur = nxt; continue; }   // block already fully removed (uniform)

        if (wave == 0) {
            const u64 lower = cur.diag & ((1ULL << lane) - 1ULL);   // earlier boxes of this block overlapping box `lane`
            u64 undecided = ~r & valid;
            u64 kept = 0ULL;
            while (undecided) {
                const bool mine = (undecided >> lane) & 1ULL;
                const bool killed = mine && (lower & kept) != 0ULL;
                const bool safe = mine && !killed && (lower & undecided) == 0ULL;
                const u64 k_new = __ballot(safe);
                const u64 r_new = __ballot(killed);
                kept |= k_new;
                undecided &= ~(k_new | r_new);
            }
            if ((kept >> lane) & 1ULL) {
                const int pos = nkept + __popcll(kept & ((1ULL << lane) - 1ULL));
                if (pos < limit) keep[pos] = (long long)(k * 64 + lane);
            }
            if (lane == 0) { s_kept = kept; s_nkept = nkept + __popcll(kept); }
        }
        __syncthreads();
        const u64 kept = s_kept;
        nkept = s_nkept;
        if (max_keep > 0 && nkept >= max_keep) { full = true; break; }

        const bool mine = (kept >> lane) & 1ULL;
#pragma unroll
        for (int q = 0; q < SCAN_PF; ++q) {
            const int j = k + 1 + wave + SCAN_WAVES * q;
            if (mine && cur.pre[q] != 0ULL) atomicOr(&remv[j], cur.pre[q]);
        }
        // columns beyond the prefetch window (large n), 4 loads in flight per lane
        const u64 *rowblk = mask + (long long)k * col_blocks * 64;
        for (int j0 = k + 1 + SCAN_WAVES * SCAN_PF + wave; j0 < col_blocks; j0 += 4 * SCAN_WAVES) {
            u64 w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = j0 + q * SCAN_WAVES;
                w[q] = (mine && j < col_blocks) ? rowblk[(long long)j * 64 + lane] : 0ULL;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (w[q] != 0ULL) atomicOr(&remv[j0 + q * SCAN_WAVES], w[q]);
        }
        __syncthreads();
        cur = nxt;
    }

    if (state != nullptr && !full && ke < col_blocks) {      // not finished: hand the state to the next chunk's launch
        __syncthreads();
        for (int j = tid; j < col_blocks; j += SCAN_THREADS) g_remv[j] = remv[j];
        if (tid == 0) { state[0] = 0; state[1] = nkept; }
        return;
    }
    const int nout = min(nkept, limit);
    if (tid == 0) { num_out[blockIdx.x] = nout; if (state != nullptr) state[0] = 1; }
    for (int j = nout + tid; j < keep_stride; j += SCAN_THREADS) keep[j] = -1LL;
}

inline int check_launch()
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

template <int STRIDE>
int launch_mask(const float *dets, int batch, int n, float thresh, int rule, int fill_lower, int block_major,
                u64 *mask, hipStream_t s, int kb = 0, int ke = -1, const int *done = nullptr, long long stride_bytes = 0)
{
    const int col_blocks = (n + 63) / 64;
    if (col_blocks > 65535 || batch > 65535) return MDT_ERR_UNSUPPORTED;
    if (ke < 0) ke = col_blocks;
    dim3 grid((col_blocks + MASK_WAVES - 1) / MASK_WAVES, ke - kb, batch);
    (void)hipGetLastError(); hipLaunchKernelGGL(nms_mask_kernel<STRIDE>, grid, dim3(64 * MASK_WAVES), 0, s,
                       dets, n, thresh, rule, fill_lower, block_major, mask, kb, done, stride_bytes);
    return check_launch();
}

// lazy form: chunk boundaries in 64-row blocks (8 blocks = 512 rows first, then 4x as many, then the rest)
constexpr int NMS_LAZY_MIN_BLOCKS = 24;          // below this many row blocks (n <= 1536) the single pass is as fast
inline size_t nms_state_bytes(int n) { return ((size_t)((n + 63) / 64) * sizeof(u64) + 16 + 255) & ~(size_t)255; }

template <int STRIDE>
int nms_impl(const float *dets, int batch, int n, float thresh, int rule, int max_keep,
             long long *keep, int keep_stride, int *num_out, void *ws, size_t ws_bytes, hipStream_t s)
{
    if (n < 0 || batch < 0 || (rule != MDT_NMS_RULE_GT && rule != MDT_NMS_RULE_GE) || keep_stride < 0)
        return MDT_ERR_INVALID_ARGUMENT;
    if (batch == 0) return MDT_OK;
    if (n == 0) {  // reference would launch a (0,0) grid unchecked (nms_kernel.cu:85-91); return num_out = 0
        if (hipMemsetAsync(num_out, 0, sizeof(int) * batch, s) != hipSuccess) return MDT_ERR_LAUNCH_FAILED;
        if (keep_stride > 0 &&
            hipMemsetAsync(keep, 0xff, sizeof(long long) * (size_t)batch * keep_stride, s) != hipSuccess)
            return MDT_ERR_LAUNCH_FAILED;
        return MDT_OK;
    }
    const size_t need = (size_t)batch * mdt_nms_workspace_bytes(n);
    if (ws == nullptr || ws_bytes < need) return MDT_ERR_WORKSPACE_TOO_SMALL;
    const int col_blocks = (n + 63) / 64;
    const size_t lds = (size_t)col_blocks * sizeof(u64);
    if (lds > 128 * 1024) return MDT_ERR_UNSUPPORTED;  // n <= 1,048,576
    u64 *mask = reinterpret_cast<u64 *>(ws);
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(nms_scan_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MDT_ERR_LAUNCH_FAILED;
    }
    if (max_keep > 0 && max_keep * 8 < n && col_blocks >= NMS_LAZY_MIN_BLOCKS) {
        // the caller wants few boxes of many (RPN: 75 of 6000): mask + scan in growing chunks, later launches return at once when done
        const long long stride = (long long)mdt_nms_workspace_bytes(n);
        int *state = reinterpret_cast<int *>(reinterpret_cast<char *>(ws) + (stride - (long long)nms_state_bytes(n)));
        int kb = 0;
        for (int chunk = 0; kb < col_blocks; ++chunk) {
            int ke = (chunk == 0) ? 8 : (chunk == 1 ? 40 : col_blocks);
            if (ke > col_blocks || col_blocks - ke < 8) ke = col_blocks;
            const int rc = launch_mask<STRIDE>(dets, batch, n, thresh, rule, 0, 1, mask, s, kb, ke, kb > 0 ? state : nullptr, stride);
            if (rc != MDT_OK) return rc;
            (void)hipGetLastError(); hipLaunchKernelGGL(nms_scan_kernel, dim3(batch), dim3(SCAN_THREADS), lds, s,
                               mask, n, max_keep, keep, keep_stride, num_out, kb, ke, state, stride);
            if (check_launch() != MDT_OK) return MDT_ERR_LAUNCH_FAILED;
            kb = ke;
        }
        return MDT_OK;
    }
    int rc = launch_mask<STRIDE>(dets, batch, n, thresh, rule, 0, 1, mask, s);
    if (rc != MDT_OK) return rc;
    (void)hipGetLastError(); hipLaunchKernelGGL(nms_scan_kernel, dim3(batch), dim3(SCAN_THREADS), lds, s,
                       mask, n, max_keep, keep, keep_stride, num_out, 0, col_blocks, (int *)nullptr, 0LL);
    return check_launch();
}

}  // namespace

extern "C" {

size_t mdt_nms_workspace_bytes(int n)
{
    if (n <= 0) return 16;
    const size_t col_blocks = ((size_t)n + 63) / 64;   // block-major mask: col_blocks^2 blocks of 64 words, + the lazy scan's state
    return ((col_blocks * col_blocks * 64 * sizeof(u64) + 255) & ~(size_t)255) + nms_state_bytes(n);
}

int mdt_nms_mask_3d(const float *dets_sorted, int n, float thresh, int rule, unsigned long long *mask, void *stream)
{
    if (n < 0) return MDT_ERR_INVALID_ARGUMENT;
    if (n == 0) return MDT_OK;
    return launch_mask<7>(dets_sorted, 1, n, thresh, rule, 1, 0, mask, (hipStream_t)stream);
}

int mdt_nms_mask_2d(const float *dets_sorted, int n, float thresh, int rule, unsigned long long *mask, void *stream)
{
    if (n < 0) return MDT_ERR_INVALID_ARGUMENT;
    if (n == 0) return MDT_OK;
    return launch_mask<5>(dets_sorted, 1, n, thresh, rule, 1, 0, mask, (hipStream_t)stream);
}

int mdt_nms_mask_full_3d(const float *dets_sorted, int n, float thresh, int rule, unsigned long long *mask, void *stream)
{
    if (n < 0) return MDT_ERR_INVALID_ARGUMENT;
    if (n == 0) return MDT_OK;
    return launch_mask<7>(dets_sorted, 1, n, thresh, rule, 2, 0, mask, (hipStream_t)stream);
}

int mdt_nms_mask_full_2d(const float *dets_sorted, int n, float thresh, int rule, unsigned long long *mask, void *stream)
{
    if (n < 0) return MDT_ERR_INVALID_ARGUMENT;
    if (n == 0) return MDT_OK;
    return launch_mask<5>(dets_sorted, 1, n, thresh, rule, 2, 0, mask, (hipStream_t)stream);
}

int mdt_nms_3d(const float *dets_sorted, int n, float thresh, int rule, int max_keep,
               long long *keep, int *num_out, void *workspace, size_t workspace_bytes, void *stream)
{
    const int stride = (max_keep > 0 && max_keep < n) ? max_keep : n;
    return nms_impl<7>(dets_sorted, 1, n, thresh, rule, max_keep, keep, stride, num_out,
                       workspace, workspace_bytes, (hipStream_t)stream);
}

int mdt_nms_2d(const float *dets_sorted, int n, float thresh, int rule, int max_keep,
               long long *keep, int *num_out, void *workspace, size_t workspace_bytes, void *stream)
{
    const int stride = (max_keep > 0 && max_keep < n) ? max_keep : n;
    return nms_impl<5>(dets_sorted, 1, n, thresh, rule, max_keep, keep, stride, num_out,
                       workspace, workspace_bytes, (hipStream_t)stream);
}

int mdt_nms_3d_batched(const float *dets_sorted, int batch, int n, float thresh, int rule, int max_keep,
                       long long *keep, int keep_stride, int *num_out,
                       void *workspace, size_t workspace_bytes, void *stream)
{
    return nms_impl<7>(dets_sorted, batch, n, thresh, rule, max_keep, keep, keep_stride, num_out,
                       workspace, workspace_bytes, (hipStream_t)stream);
}

int mdt_nms_2d_batched(const float *dets_sorted, int batch, int n, float thresh, int rule, int max_keep,
                       long long *keep, int keep_stride, int *num_out,
                       void *workspace, size_t workspace_bytes, void *stream)
{
    return nms_impl<5>(dets_sorted, batch, n, thresh, rule, max_keep, keep, keep_stride, num_out,
                       workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
