// nms2to3d.hip -- merge of per-slice 2D detections into 3D cubes, for gfx950.
// Follows predictor.py:710-773 (nms_2to3D, float64 numpy, Python while-loop): greedy clusters by 2D IoU (+1 pixel
// convention) as in NMS, but a cluster only absorbs the matches whose slice ids are CONNECTED to the core slice
// (no "hole" in between); the others stay in the pool.  One resident workgroup walks the sequential cluster loop;
// per cluster all threads sweep the alive boxes, mark matches, and set the matched slice ids in an LDS bitmap from
// which the first hole above / below the core slice is found.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "mdt_hip.h"

namespace {

typedef unsigned long long u64;
constexpr int M_THREADS = 1024;

__global__ __launch_bounds__(M_THREADS) void nms2to3d_kernel(
    const double *__restrict__ dets, int n, int n_slices, double thresh,
    long long *__restrict__ keep, double *__restrict__ keep_z, int *__restrict__ num_out,
    unsigned char *__restrict__ state)      // 0 dead, 1 alive, 2 alive + match of the current cluster
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    u64 *present = reinterpret_cast<u64 *>(smem_raw);          // [ceil(n_slices / 64)]
    __shared__ int s_next[M_THREADS / 64];
    __shared__ int s_lo, s_hi;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwords = (n_slices + 63) / 64;
    for (int j = tid; j < n; j += M_THREADS) state[j] = 1;
    __syncthreads();

    int head = 0, n_kept = 0;
    while (true) {
        int found = n;
        for (int base = head; base < n && found == n; base += M_THREADS) {
            const int j = base + tid;
            const bool a = (j < n) && state[j];
            const u64 bal = __ballot(a);
            if (lane == 0) s_next[wave] = bal ? (base + wave * 64 + __ffsll((long long)bal) - 1) : n;
            __syncthreads();
            for (int w = 0; w < M_THREADS / 64; ++w) found = min(found, s_next[w]);
            __syncthreads();
        }
        if (found >= n) break;
        head = found;
        const int i = head;
        const double *bi = dets + (long long)i * 6;
        const double area_i = (bi[3] - bi[1] + 1.0) * (bi[2] - bi[0] + 1.0);   // (x2-x1+1)*(y2-y1+1), predictor.py:735
        const int core = (int)bi[5];
        for (int w = tid; w < nwords; w += M_THREADS) present[w] = 0ULL;
        __syncthreads();
        for (int j = head + tid; j < n; j += M_THREADS) {
            if (!state[j]) continue;
            const double *bj = dets + (long long)j * 6;
            const double xx1 = fmax(bi[1], bj[1]), yy1 = fmax(bi[0], bj[0]);
            const double xx2 = fmin(bi[3], bj[3]), yy2 = fmin(bi[2], bj[2]);
            const double w_ = fmax(0.0, xx2 - xx1 + 1.0), h_ = fmax(0.0, yy2 - yy1 + 1.0);
            const double inter = w_ * h_;
            const double area_j = (bj[3] - bj[1] + 1.0) * (bj[2] - bj[0] + 1.0);
            const double ovr = inter / (area_i + area_j - inter);
            if (ovr > thresh || j == i) {
                state[j] = 2;
                const int sl = (int)bj[5];
                if (sl >= 0 && sl < n_slices) atomicOr(&present[sl >> 6], 1ULL << (sl & 63));
            }
        }
        __syncthreads();
        if (tid == 0) {
            // connected range around the core slice: up to (excluding) the first hole on either side (:751-755)
            int hi = core, lo = core;
            while (hi + 1 < n_slices && ((present[(hi + 1) >> 6] >> ((hi + 1) & 63)) & 1ULL)) ++hi;
            while (lo - 1 >= 0 && ((present[(lo - 1) >> 6] >> ((lo - 1) & 63)) & 1ULL)) --lo;
            s_lo = lo; s_hi = hi;
            keep[n_kept] = i;
            keep_z[2 * n_kept] = (double)lo - 1.0;       // z1 = min(connected slice ids) - 1, z2 = max + 1 (:757-758)
            keep_z[2 * n_kept + 1] = (double)hi + 1.0;
        }
        __syncthreads();
        const int lo = s_lo, hi = s_hi;
        for (int j = head + tid; j < n; j += M_THREADS) {
            if (state[j] != 2) continue;
            const int sl = (int)dets[(long long)j * 6 + 5];
            state[j] = (sl >= lo && sl <= hi) ? 0 : 1;   // only connected matches leave the pool (:761)
        }
        ++n_kept;
        ++head;
        __syncthreads();
    }
    if (tid == 0) *num_out = n_kept;
}

}  // namespace

extern "C" {

size_t mdt_nms_2to3d_workspace_bytes(int n) { return (((size_t)(n > 0 ? n : 1)) + 255) & ~(size_t)255; }

int mdt_nms_2to3d(const double *dets_sorted, int n, int n_slices, double thresh,
                  long long *keep, double *keep_z, int *num_out,
                  void *workspace, size_t workspace_bytes, void *stream)
{
    (void)hipGetLastError();
    if (n < 0 || n_slices <= 0 || !num_out) return MDT_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return hipMemsetAsync(num_out, 0, sizeof(int), s) == hipSuccess ? MDT_OK : MDT_ERR_LAUNCH_FAILED;
    if (!workspace || workspace_bytes < mdt_nms_2to3d_workspace_bytes(n)) return MDT_ERR_WORKSPACE_TOO_SMALL;
    const size_t lds = (size_t)((n_slices + 63) / 64) * sizeof(u64);
    if (lds > 48 * 1024) return MDT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(nms2to3d_kernel, dim3(1), dim3(M_THREADS), lds, s, dets_sorted, n, n_slices, thresh, keep, keep_z, num_out,
                       reinterpret_cast<unsigned char *>(workspace));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return e == hipSuccess ? MDT_OK : MDT_ERR_LAUNCH_FAILED;
}

}  // extern "C"
