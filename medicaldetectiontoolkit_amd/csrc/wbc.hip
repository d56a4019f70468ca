// wbc.hip -- weighted box clustering (consolidation of patch / TTA / ensemble
// predictions) for gfx950.  Follows predictor.py:597-706 (float64 numpy; a Python
// while-loop with one numpy sweep per cluster, run in multiprocessing.Pool(6)).
//
// Design: one workgroup per problem (one patient x class).  The greedy cluster
// loop is inherently sequential, so a single resident workgroup walks it: per
// cluster, all threads sweep the still-alive boxes (IoU vs the current top box,
// +1 pixel convention), accumulate the weighted sums in float64 registers, and a
// fixed-shape wave/LDS tree reduces them -> deterministic.  The set of unique
// patch ids in a cluster is counted with a stamp table (one atomicExch per member).
// Summation order differs from numpy's pairwise np.sum, so scores/coords agree
// to ~1e-12 relative, cluster membership exactly.

#include <hip/hip_runtime.h>
#include "mdt_hip.h"

namespace {

constexpr int WBC_THREADS = 1024;
constexpr int NSUM = 10;  // sum_w, sum_sw, sum_novs, count, 6 coord sums

template <int DIM>
__global__ __launch_bounds__(WBC_THREADS) void wbc_kernel(
    const double *__restrict__ dets, const int *__restrict__ patch_ids, int n, int n_patch_ids,
    double thresh, double n_ens, double *__restrict__ out_scores, double *__restrict__ out_coords,
    int *__restrict__ num_out, unsigned char *__restrict__ alive, int *__restrict__ stamp)
{
    constexpr int ROW = 2 * DIM + 3;
    __shared__ double s_red[WBC_THREADS / 64][NSUM];
    __shared__ int s_uniq[WBC_THREADS / 64];
    __shared__ int s_head;
    __shared__ int s_next[WBC_THREADS / 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = tid; j < n; j += WBC_THREADS) alive[j] = 1;
    for (int j = tid; j < n_patch_ids; j += WBC_THREADS) stamp[j] = 0;
    if (tid == 0) s_head = 0;
    __syncthreads();

    int n_clusters = 0;  // stamp value of the current cluster is n_clusters + 1
    int n_kept = 0;
    int head = 0;
    while (true) {
        // ---- find the first alive box at or after head ----
        int found = n;
        for (int base = head; base < n && found == n; base += WBC_THREADS) {
            const int j = base + tid;
            const bool a = (j < n) && alive[j];
            const unsigned long long bal = __ballot(a);
            if (lane == 0) s_next[wave] = bal ? (base + wave * 64 + __ffsll((long long)bal) - 1) : n;
            __syncthreads();
            for (int w = 0; w < WBC_THREADS / 64; ++w) found = min(found, s_next[w]);
            __syncthreads();
        }
        if (found >= n) break;
        head = found;
        const int i = head;
        const double *bi = dets + (long long)i * ROW;
        // areas = (y2 - y1 + 1) * (x2 - x1 + 1) [* (z2 - z1 + 1)]   predictor.py:619-623
        double area_i = (bi[2] - bi[0] + 1.0) * (bi[3] - bi[1] + 1.0);
        if (DIM == 3) area_i = area_i * (bi[5] - bi[4] + 1.0);

        double acc[NSUM];
#pragma unroll
        for (int q = 0; q < NSUM; ++q) acc[q] = 0.0;
        int uniq = 0;
        const int stamp_val = n_clusters + 1;

        for (int j = head + tid; j < n; j += WBC_THREADS) {
            if (!alive[j]) continue;
            const double *bj = dets + (long long)j * ROW;
            // predictor.py:634-650
            const double xx1 = fmax(bi[1], bj[1]);
            const double yy1 = fmax(bi[0], bj[0]);
            const double xx2 = fmin(bi[3], bj[3]);
            const double yy2 = fmin(bi[2], bj[2]);
            const double w = fmax(0.0, xx2 - xx1 + 1.0);
            const double h = fmax(0.0, yy2 - yy1 + 1.0);
            double inter = w * h;
            double area_j = (bj[2] - bj[0] + 1.0) * (bj[3] - bj[1] + 1.0);
            if (DIM == 3) {
                const double zz1 = fmax(bi[4], bj[4]);
                const double zz2 = fmin(bi[5], bj[5]);
                const double d = fmax(0.0, zz2 - zz1 + 1.0);
                inter = inter * d;
                area_j = area_j * (bj[5] - bj[4] + 1.0);
            }
            const double ovr = inter / (area_i + area_j - inter);
            if (ovr > thresh || j == i) {
                const double score = bj[2 * DIM];
                const double pc = bj[2 * DIM + 1];
                const double novs = bj[2 * DIM + 2];
                const double wgt = ovr * area_j * pc;   // match_ov_facts * match_areas * match_pc_facts
                const double sw = score * wgt;          // match_scores *= match_score_weights
                acc[0] += wgt;
                acc[1] += sw;
                acc[2] += novs;
                acc[3] += 1.0;
#pragma unroll
                for (int q = 0; q < 2 * DIM; ++q) acc[4 + q] += bj[q] * sw;
                const int pid = patch_ids[j];
                if (pid >= 0 && pid < n_patch_ids) {
                    if (atomicExch(&stamp[pid], stamp_val) != stamp_val) ++uniq;
                }
                alive[j] = 0;  // inds = np.where(ovr <= thresh): matched boxes leave the pool
            }
        }
        // ---- deterministic tree reduction ----
#pragma unroll
        for (int q = 0; q < NSUM; ++q) {
            double v = acc[q];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0) s_red[wave][q] = v;
        }
        {
            int u = uniq;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) u += __shfl_down(u, off);
            if (lane == 0) s_uniq[wave] = u;
        }
        __syncthreads();
        if (tid == 0) {
            double tot[NSUM];
            for (int q = 0; q < NSUM; ++q) {
                double v = 0.0;
                for (int w = 0; w < WBC_THREADS / 64; ++w) v += s_red[w][q];
                tot[q] = v;
            }
            int u = 0;
            for (int w = 0; w < WBC_THREADS / 64; ++w) u += s_uniq[w];
            const double cnt = tot[3];
            const double n_expected = n_ens * (tot[2] / cnt);           // n_ens * np.mean(match_n_ovs)
            double n_missing = n_expected - (double)u;                  // np.max((0, n_expected - n_unique))
            // numpy's max propagates NaN (a member with an empty overlap slice has n_overlaps = NaN, predictor.py:434):
            // the cluster score is then NaN and fails the `> 0.01` test below, exactly like the reference
            if (n_missing == n_missing && !(n_missing > 0.0)) n_missing = 0.0;
            const double denom = tot[0] + n_missing * (tot[0] / cnt);   // + n_missing * np.mean(weights)
            const double avg_score = tot[1] / denom;
            if (avg_score > 0.01) {                                      // predictor.py:697
                out_scores[n_kept] = avg_score;
                for (int q = 0; q < 2 * DIM; ++q) out_coords[(long long)n_kept * 2 * DIM + q] = tot[4 + q] / tot[1];
                s_head = n_kept + 1;
            } else {
                s_head = n_kept;
            }
        }
        __syncthreads();
        n_kept = s_head;
        ++n_clusters;
        ++head;
        __syncthreads();
    }
    if (tid == 0) *num_out = n_kept;
}

}  // namespace

extern "C" {

size_t mdt_wbc_workspace_bytes(int n, int n_patch_ids)
{
    size_t bytes = ((size_t)(n > 0 ? n : 1) + 15) & ~(size_t)15;
    bytes += (size_t)(n_patch_ids > 0 ? n_patch_ids : 1) * sizeof(int);
    return (bytes + 255) & ~(size_t)255;
}

int mdt_weighted_box_clustering(const double *dets_sorted, const int *patch_ids, int n, int dim,
                                int n_patch_ids, double thresh, double n_ens,
                                double *out_scores, double *out_coords, int *num_out,
                                void *workspace, size_t workspace_bytes, void *stream)
{
    (void)hipGetLastError();   // drop stale error state of earlier runtime calls on this thread
    if (n < 0 || (dim != 2 && dim != 3) || n_patch_ids < 0 || !num_out) return MDT_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        return hipMemsetAsync(num_out, 0, sizeof(int), s) == hipSuccess ? MDT_OK : MDT_ERR_LAUNCH_FAILED;
    }
    if (!workspace || workspace_bytes < mdt_wbc_workspace_bytes(n, n_patch_ids)) return MDT_ERR_WORKSPACE_TOO_SMALL;
    unsigned char *alive = reinterpret_cast<unsigned char *>(workspace);
    int *stamp = reinterpret_cast<int *>(alive + (((size_t)n + 15) & ~(size_t)15));
    if (dim == 3) hipLaunchKernelGGL(wbc_kernel<3>, dim3(1), dim3(WBC_THREADS), 0, s, dets_sorted, patch_ids, n, n_patch_ids,
                           thresh, n_ens, out_scores, out_coords, num_out, alive, stamp);
    else hipLaunchKernelGGL(wbc_kernel<2>, dim3(1), dim3(WBC_THREADS), 0, s, dets_sorted, patch_ids, n, n_patch_ids,
                           thresh, n_ens, out_scores, out_coords, num_out, alive, stamp);
    return hipGetLastError() == hipSuccess ? MDT_OK : MDT_ERR_LAUNCH_FAILED;
}

}  // extern "C"
