// declarations shared by the two A/B translation units (libmdt_hip_ab.so only)
#ifndef MDT_ROI_ALIGN_AB_COMMON_H
#define MDT_ROI_ALIGN_AB_COMMON_H
#include "../roi_align_common.h"

namespace mdt_ra {

// ab/roi_align_bwd.hip: the round-2 backward (single launch, RoI-territory form).  Returns MDT_ERR_UNSUPPORTED when the
// shape does not fit its LDS budgets; the caller then falls back to the two-kernel / ordered forms.
bool bwd_territory_supported(int dim, int N, int B, int H, int W, int D, int ph, int pw, int pd, int C);
int launch_bwd_territory(int dim, const float *grads, const float *boxes, const int *box_ind, int N, int B,
                         int H, int W, int D, int ph, int pw, int pd, int C, float *out, hipStream_t s);

int launch_bwd_territory_multi(int dim, int n_levels, const float *grads, const float *boxes, const int *batch_ix, const int *level,
                               int N, int B, int C, const int *H, const int *W, const int *D, int ph, int pw, int pd,
                               float *const *outs, hipStream_t s);

}  // namespace mdt_ra

#endif
