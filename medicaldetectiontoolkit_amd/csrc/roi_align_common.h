// roi_align_common.h -- device helpers shared by the RoIAlign translation units (roi_align.hip, roi_align_bwd.hip).
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

inline int check_launch()
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MDT_OK;
    if (getenv("MDT_VERBOSE")) fprintf(stderr, "libmdt_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return MDT_ERR_LAUNCH_FAILED;
}

// roi_align_bwd.hip: default backward (single launch, RoI-territory form).  Returns MDT_ERR_UNSUPPORTED when the
// shape does not fit its LDS budgets; the caller then falls back to the two-kernel / ordered forms.
bool bwd_territory_supported(int dim, int N, int B, int H, int W, int D, int ph, int pw, int pd, int C);
int launch_bwd_territory(int dim, const float *grads, const float *boxes, const int *box_ind, int N, int B,
                         int H, int W, int D, int ph, int pw, int pd, int C, float *out, hipStream_t s);

int launch_bwd_territory_multi(int dim, int n_levels, const float *grads, const float *boxes, const int *batch_ix, const int *level,
                               int N, int B, int C, const int *H, const int *W, const int *D, int ph, int pw, int pd,
                               float *const *outs, hipStream_t s);

// roi_align_bwd_v3.hip: round-3 default backward (gather form; one launch for one map or for all pyramid levels; 2D maps as
// the W = 1 case).  `level` may be null (every RoI on level 0).  MDT_ERR_UNSUPPORTED: outside its budgets -> the caller
// falls back to the forms above.
int launch_bwd_gather(int dim, int n_levels, const float *grads, const float *boxes, const int *batch_ix, const int *level,
                      int N, int B, int C, const int *H, const int *W, const int *D, int ph, int pw, int pd,
                      float *const *outs, hipStream_t s);

}  // namespace mdt_ra

#endif
