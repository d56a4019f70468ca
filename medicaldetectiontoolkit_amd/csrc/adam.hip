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

}  // namespace

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
