// adam.hip -- Adam (exec.py:39: torch.optim.Adam(lr, weight_decay)) over FLAT fp32 buffers: parameters, gradients and both moment
// estimates of the whole model (4.94 M values, 19.75 MB each) live in one buffer each (training.FlatAdam), so the update is ONE
// launch over 5 x 19.75 MB instead of torch's ~20 multi-tensor launches (0.3 ms of GPU time and 3.4 ms of host time per step).
// Same arithmetic as torch's single-tensor rule (torch/optim/adam.py _single_tensor_adam), in fp32, no FMA contraction:
//     g' = g + wd * p;  m += (g' - m) * (1 - b1);  v = b2 * v + (1 - b2) * g' * g';  p += -step_size * (m / (sqrt(v) / bc2_sqrt + eps))
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mdt_hip.h"

namespace {

struct AdamScalars {
    float one_minus_b1, b2, one_minus_b2, eps, wd, neg_step_size, bc2_sqrt, grad_div;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamScalars &a)
{
    if (a.grad_div != 1.0f) g = g / a.grad_div;      // the averaging of the gradient all-reduce (sum over ranks / world), same IEEE division as flat.div_(world)
    if (a.wd != 0.0f) g = g + a.wd * p;
    m = m + (g - m) * a.one_minus_b1;
    v = v * a.b2 + a.one_minus_b2 * (g * g);
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p + a.neg_step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_flat_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                         float *__restrict__ v, long long n4, long long n, AdamScalars a)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4 *>(p)[i], mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
        const float4 gg = reinterpret_cast<const float4 *>(g)[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, a);
        adam_one(pp.y, gg.y, mm.y, vv.y, a);
        adam_one(pp.z, gg.z, mm.z, vv.z, a);
        adam_one(pp.w, gg.w, mm.w, vv.w, a);
        reinterpret_cast<float4 *>(p)[i] = pp;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
    }
    // tail (n not a multiple of 4): the first few threads of the grid
    const long long t = 4 * n4 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) adam_one(p[t], g[t], m[t], v[t], a);
}

// ---- per-segment form (round 5): torch.optim.Adam's per-PARAMETER semantics over the same flat buffers ------------------------------
// torch.optim.Adam keeps one step counter per parameter and SKIPS a parameter that has no gradient in a step (p.grad is None: no moment
// decay, no weight decay, no update, counter unchanged).  In the reference that happens whenever a loss term is a constant -- no positive
// RoI: compute_mrcnn_bbox_loss / compute_mrcnn_mask_loss return FloatTensor([0]) (mrcnn.py:266-268, 287-288), no positive anchor:
// compute_rpn_bbox_loss (:233-234) -- so the mask head, linear_bbox, conv_bbox are not touched by exec.py:74 in such a step.  This repo's
// fixed-size masked step produces an exact ZERO gradient for them instead, so whether a segment "has a gradient" is decided here, on the
// device, from (a) a host-known flag per segment (p.grad was None) and (b) a condition value the step wrote (count of positives > 0).
struct SegScalars {
    float neg_step_size, bc2_sqrt;
    int active, pad;
};

// one thread per segment: decide, count, derive the bias-correction scalars of the segment's OWN step count (double, as torch's python side)
__global__ void adam_segments_prepare_kernel(int nseg, int *__restrict__ seg_step, const unsigned char *__restrict__ present,
                                             const int *__restrict__ cond_id, const float *__restrict__ cond, int policy,
                                             double lr, double beta1, double beta2, SegScalars *__restrict__ scal)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    bool active = present == nullptr || present[s] != 0;
    if (active && cond_id != nullptr && cond != nullptr && cond_id[s] >= 0) active = cond[cond_id[s]] > 0.0f;
    int t = seg_step[s];
    if (policy == 1 && t > 0) active = true;     // torch 0.4.1 zero_grad(): after its first gradient a parameter always has one (a zero tensor)
    SegScalars o;
    o.active = active ? 1 : 0;
    o.pad = 0;
    o.neg_step_size = 0.0f;
    o.bc2_sqrt = 1.0f;
    if (active) {
        t += 1;
        seg_step[s] = t;
        const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
        o.neg_step_size = (float)(-(lr / bc1));
        o.bc2_sqrt = (float)sqrt(bc2);
    }
    scal[s] = o;
}

// ARITH 0: every operation rounded on its own (the arithmetic of adam_one above).  ARITH 1: the contraction pattern of torch's foreach
// kernels as hipcc compiles them (a + alpha * b -> fma): lerp, addcmul and addcdiv each end in one fma -- bit-equal to
// torch.optim.Adam(foreach=True) on this stack (tests/test_flat_adam_gpu.py).
template <int ARITH>
__device__ __forceinline__ void adam_seg_one(float &p, float g, float &m, float &v, const SegScalars &sc, const AdamScalars &a)
{
    if (a.grad_div != 1.0f) g = g / a.grad_div;
    if (ARITH == 0) {
        if (a.wd != 0.0f) g = g + a.wd * p;
        m = m + (g - m) * a.one_minus_b1;
        v = v * a.b2 + a.one_minus_b2 * (g * g);
        const float denom = sqrtf(v) / sc.bc2_sqrt + a.eps;
        p = p + sc.neg_step_size * (m / denom);
    } else {
        if (a.wd != 0.0f) g = __builtin_fmaf(a.wd, p, g);                 // _foreach_add(grads, params, alpha=wd)
        m = __builtin_fmaf(a.one_minus_b1, g - m, m);                     // _foreach_lerp_(exp_avgs, grads, 1 - beta1)
        v = v * a.b2;                                                     // _foreach_mul_(exp_avg_sqs, beta2)
        v = __builtin_fmaf(a.one_minus_b2, g * g, v);                     // _foreach_addcmul_(exp_avg_sqs, grads, grads, 1 - beta2): a + s * (b * c)
        const float denom = sqrtf(v) / sc.bc2_sqrt + a.eps;               // _foreach_sqrt, _foreach_div_, _foreach_add_
        p = __builtin_fmaf(sc.neg_step_size, m / denom, p);               // _foreach_addcdiv_(params, exp_avgs, denom, step_size)
    }
}

constexpr int SEG_LDS_MAX = 2048;

template <int ARITH>
__global__ __launch_bounds__(256) void adam_flat_segments_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                                  float *__restrict__ v, long long n, const long long *__restrict__ seg_off,
                                                                  int nseg, const SegScalars *__restrict__ scal, AdamScalars a, int aligned)
{
    __shared__ long long s_off[SEG_LDS_MAX + 1];
    const bool lds = nseg <= SEG_LDS_MAX;
    if (lds) {
        for (int i = threadIdx.x; i <= nseg; i += blockDim.x) s_off[i] = seg_off[i];
        __syncthreads();
    }
    const long long *off = lds ? s_off : seg_off;
    const long long n4 = (n + 3) / 4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
        const long long i0 = 4 * q;
        int lo = 0, hi = nseg - 1;                      // segment of element i0: largest s with off[s] <= i0
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (off[mid] <= i0) lo = mid; else hi = mid - 1;
        }
        int seg = lo;
        const int cnt = (int)((n - i0 < 4) ? (n - i0) : 4);
        // all of the quad inside one INACTIVE segment: nothing to load
        if (i0 + cnt <= off[seg + 1] && !scal[seg].active) continue;
        float pp[4], gg[4], mm[4], vv[4];
        if (aligned && cnt == 4) {
            const float4 a4 = reinterpret_cast<const float4 *>(p)[q], b4 = reinterpret_cast<const float4 *>(g)[q];
            const float4 c4 = reinterpret_cast<const float4 *>(m)[q], d4 = reinterpret_cast<const float4 *>(v)[q];
            pp[0] = a4.x; pp[1] = a4.y; pp[2] = a4.z; pp[3] = a4.w;
            gg[0] = b4.x; gg[1] = b4.y; gg[2] = b4.z; gg[3] = b4.w;
            mm[0] = c4.x; mm[1] = c4.y; mm[2] = c4.z; mm[3] = c4.w;
            vv[0] = d4.x; vv[1] = d4.y; vv[2] = d4.z; vv[3] = d4.w;
        } else {
            for (int k = 0; k < 4; ++k) {
                const bool in = k < cnt;
                pp[k] = in ? p[i0 + k] : 0.f; gg[k] = in ? g[i0 + k] : 0.f; mm[k] = in ? m[i0 + k] : 0.f; vv[k] = in ? v[i0 + k] : 0.f;
            }
        }
        bool any = false;
        SegScalars sc = scal[seg];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= cnt) break;
            while (i0 + k >= off[seg + 1]) { ++seg; sc = scal[seg]; }     // (zero-length segments are skipped over)
            if (sc.active) { adam_seg_one<ARITH>(pp[k], gg[k], mm[k], vv[k], sc, a); any = true; }
        }
        if (!any) continue;
        if (aligned && cnt == 4) {
            reinterpret_cast<float4 *>(p)[q] = make_float4(pp[0], pp[1], pp[2], pp[3]);
            reinterpret_cast<float4 *>(m)[q] = make_float4(mm[0], mm[1], mm[2], mm[3]);
            reinterpret_cast<float4 *>(v)[q] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
            for (int k = 0; k < cnt; ++k) { p[i0 + k] = pp[k]; m[i0 + k] = mm[k]; v[i0 + k] = vv[k]; }
        }
    }
}

}  // namespace

extern "C" size_t mdt_adam_flat_segments_workspace_bytes(int nseg)
{
    return (size_t)(nseg > 0 ? nseg : 1) * sizeof(SegScalars);
}

extern "C" int mdt_adam_flat_segments(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n,
                                      const long long *seg_off, int nseg, int *seg_step, const unsigned char *present, const int *cond_id,
                                      const float *cond, int policy, int arith, double lr, double beta1, double beta2, double eps,
                                      double weight_decay, double grad_div, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || !seg_off || !seg_step || n < 0 || nseg < 1 || !(grad_div > 0.0) || policy < 0 || policy > 1 ||
        arith < 0 || arith > 1)
        return MDT_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < mdt_adam_flat_segments_workspace_bytes(nseg) || (reinterpret_cast<uintptr_t>(workspace) & 15))
        return MDT_ERR_WORKSPACE_TOO_SMALL;
    if (n == 0) return MDT_OK;
    SegScalars *scal = reinterpret_cast<SegScalars *>(workspace);
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    hipLaunchKernelGGL(adam_segments_prepare_kernel, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, s, nseg, seg_step, present, cond_id, cond,
                       policy, lr, beta1, beta2, scal);
    AdamScalars a;
    a.one_minus_b1 = (float)(1.0 - beta1);
    a.b2 = (float)beta2;
    a.one_minus_b2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    a.wd = (float)weight_decay;
    a.neg_step_size = 0.0f;
    a.bc2_sqrt = 1.0f;
    a.grad_div = (float)grad_div;
    const uintptr_t al = reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
                         reinterpret_cast<uintptr_t>(exp_avg_sq);
    const long long n4 = (n + 3) / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (arith == 0)
        hipLaunchKernelGGL(adam_flat_segments_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n, seg_off, nseg,
                           scal, a, (int)((al & 15) == 0));
    else
        hipLaunchKernelGGL(adam_flat_segments_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n, seg_off, nseg,
                           scal, a, (int)((al & 15) == 0));
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

extern "C" int mdt_adam_flat(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, double lr, double beta1, double beta2,
                             double eps, double weight_decay, long long step, double grad_div, void *stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1 || !(grad_div > 0.0)) return MDT_ERR_INVALID_ARGUMENT;
    if (n == 0) return MDT_OK;
    const uintptr_t al = reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
                         reinterpret_cast<uintptr_t>(exp_avg_sq);
    const long long n4 = (al & 15) == 0 ? n / 4 : 0;
    // hyper-parameters arrive as doubles (python floats) and every derived scalar -- 1 - beta, the bias corrections 1 - beta ** step,
    // the step size -- is formed in double and rounded to fp32 once, as torch does (1 - 0.999f would be off by 1.3e-5)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamScalars a;
    a.one_minus_b1 = (float)(1.0 - beta1);
    a.b2 = (float)beta2;
    a.one_minus_b2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    a.wd = (float)weight_decay;
    a.neg_step_size = (float)(-lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.grad_div = (float)grad_div;
    long long work = n4 > 0 ? n4 : n;
    if (n - 4 * n4 > work) work = n - 4 * n4;
    long long blocks = (work + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    // the tail indexes threads from 0: it needs at least (n - 4 n4) threads in the grid
    if (blocks * 256 < n - 4 * n4) blocks = (n - 4 * n4 + 255) / 256;
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n4, n, a);
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}
