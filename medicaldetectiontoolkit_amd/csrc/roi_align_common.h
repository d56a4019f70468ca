// roi_align_common.h -- device helpers shared by the RoIAlign translation units (roi_align.hip, roi_align_fwd.hip, roi_align_bwd_v3.hip; ab/*.hip).
// Sample-coordinate arithmetic follows the reference CUDA kernels
// (cuda_functions/roi_align_3D/roi_align/src/cuda/crop_and_resize_kernel.cu:51-75); every translation unit that
// includes this file is compiled with -ffp-contract=off so the results round like the uncontracted CPU oracle.
#ifndef MDT_ROI_ALIGN_COMMON_H
#define MDT_ROI_ALIGN_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mdt_hip.h"

namespace mdt_ra {

typedef float v4f __attribute__((ext_vector_type(4)));

struct AxisEntry {
    int lo;      // floorf(in)
    float lerp;  // in - lo;  ceilf(in) == lo + (lerp > 0)
};

// crop_and_resize_kernel.cu:51-75 -- see oracle/mdt_oracle.c sample_coord for the
// type analysis (the 0.5 literals are double).
__device__ __forceinline__ float sample_coord(float a1, float a2, int L, int P, int p)
{
    float in;
    if (P > 1) {
        const float scale = (a2 - a1) * (float)L / (float)P;
        const float t = a1 * (float)L + (float)p * scale + scale / 2.0f;
        in = (float)((double)t - 0.5);
    } else {
        in = (float)(0.5 * (double)(a1 + a2) * (double)L);
    }
    if (in > (float)(L - 1)) in = (float)(L - 1);
    if (in < 0.0f) in = 0.0f;
    return in;
}

__device__ __forceinline__ AxisEntry axis_entry(float a1, float a2, int L, int P, int p)
{
    const float in = sample_coord(a1, a2, L, P, p);
    AxisEntry e;
    e.lo = (int)floorf(in);
    e.lerp = in - (float)e.lo;
    return e;
}

__device__ __forceinline__ int entry_hi(const AxisEntry &e) { return e.lo + (e.lerp > 0.0f ? 1 : 0); }

// weight of sample entry e towards voxel index idx (sum of the floor and ceil contributions)
__device__ __forceinline__ float axis_weight(const AxisEntry &e, int idx)
{
    float w = 0.0f;
    if (e.lo == idx) w = 1.0f - e.lerp;
    if (entry_hi(e) == idx) w = w + e.lerp;   // lo == hi happens only with lerp == 0
    return w;
}

// input element of the forward kernels: fp32, or bf16 (raw 16-bit pattern, widened exactly to fp32 on load -- the interpolation itself is
// always fp32; used by the autocast inference path, where it halves the gathered bytes), or uint8 (round 4: the GT masks of a training
// batch travel and stay as uint8; the mask-target crop of detection_target_layer, mrcnn.py:551-563, reads them as they are)
struct bf16raw { unsigned short v; };
struct u8raw { unsigned char v; };
__device__ __forceinline__ float ld(const float *p, long long i) { return p[i]; }
__device__ __forceinline__ float ld(const bf16raw *p, long long i) { return __uint_as_float(((unsigned int)p[i].v) << 16); }
__device__ __forceinline__ float ld(const u8raw *p, long long i) { return (float)p[i].v; }
// four consecutive elements (16 / 8 / 4 bytes), widened exactly
__device__ __forceinline__ void ld4(const float *p, long long i, float o[4]) { const v4f q = *reinterpret_cast<const v4f *>(p + i); o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w; }
__device__ __forceinline__ void ld4(const bf16raw *p, long long i, float o[4])
{
    const uint2 q = *reinterpret_cast<const uint2 *>(p + i);
    o[0] = __uint_as_float(q.x << 16); o[1] = __uint_as_float(q.x & 0xffff0000u); o[2] = __uint_as_float(q.y << 16); o[3] = __uint_as_float(q.y & 0xffff0000u);
}
__device__ __forceinline__ void ld4(const u8raw *p, long long i, float o[4])
{
    const unsigned int q = *reinterpret_cast<const unsigned int *>(p + i);
    o[0] = (float)(q & 255u); o[1] = (float)((q >> 8) & 255u); o[2] = (float)((q >> 16) & 255u); o[3] = (float)(q >> 24);
}

// the maps of a launch: one level, or all pyramid levels (mrcnn.py:373-457 pools every RoI on exactly one level)
constexpr int PYR_MAX_LEVELS = 5;
struct PyramidMaps {
    const void *image[PYR_MAX_LEVELS];
    int H[PYR_MAX_LEVELS], W[PYR_MAX_LEVELS], D[PYR_MAX_LEVELS];
    int n_levels;
};

// roi_align_fwd.hip: 3D forward, channel-quad form (round 5).  MDT_ERR_UNSUPPORTED: outside its budgets -> the caller falls back to the
// direct kernel of roi_align.hip.  `level` may be null (every RoI on level 0).
template <typename TIN>
int launch_fwd_cq(const PyramidMaps &maps, const float *boxes, const int *box_ind, const int *level, int N, int B,
                  int ch, int cw, int cd, int C, float *crops, hipStream_t s);

inline int check_launch()
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

// roi_align_bwd_v3.hip: round-3 default backward (gather form; one launch for one map or for all pyramid levels; 2D maps as
// the W = 1 case).  `level` may be null (every RoI on level 0).  MDT_ERR_UNSUPPORTED: outside its budgets -> the caller
// falls back to the forms above.
int launch_bwd_gather(int dim, int n_levels, const float *grads, const float *boxes, const int *batch_ix, const int *level,
                      int N, int B, int C, const int *H, const int *W, const int *D, int ph, int pw, int pd,
                      float *const *outs, hipStream_t s);
int launch_bwd_gather_acc(int dim, int n_levels, const float *grads, const float *boxes, const int *batch_ix, const int *level,
                      int N, int B, int C, const int *H, const int *W, const int *D, int ph, int pw, int pd,
                      float *const *outs, hipStream_t s, int accumulate);

}  // namespace mdt_ra

#endif
