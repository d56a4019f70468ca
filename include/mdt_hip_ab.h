/*
 * mdt_hip_ab.h -- C ABI of libmdt_hip_ab.so: SUPERSEDED kernel generations of the RoIAlign backward, kept as A/B baselines and test
 * subjects (tools/microbench.py, tests/test_hip_gpu.py).  Not part of the product library libmdt_hip.so and never loaded by the
 * package's ops (medicaldetectiontoolkit_amd/_lib.py: ab_lib() is called by tests and tools only).  Conventions as in mdt_hip.h.
 */
#ifndef MDT_HIP_AB_H
#define MDT_HIP_AB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Two-kernel separable form (round-1 default, kept as A/B baseline and fallback): kernel 1 = per-(RoI, channel)
 * expansion into compact blocks in `workspace` running beside a zero-fill role; kernel 2 patches the touched voxels.
 * workspace: mdt_crop_and_resize_backward_twophase_workspace_bytes(...), 16-byte aligned.  Same numerics contract. */
size_t mdt_crop_and_resize_backward_twophase_workspace_bytes(int dim, int num_boxes, int depth,
                                                            int image_height, int image_width, int image_zdepth,
                                                            int crop_height, int crop_width, int crop_zdepth);

int mdt_crop_and_resize_3d_backward_twophase(
    const float *grads, const float *boxes, const int *box_ind,
    int num_boxes, int batch, int image_height, int image_width, int image_zdepth,
    int crop_height, int crop_width, int crop_zdepth, int depth,
    float *grads_image, void *workspace, size_t workspace_bytes, void *stream);

int mdt_crop_and_resize_2d_backward_twophase(
    const float *grads, const float *boxes, const int *box_ind,
    int num_boxes, int batch, int image_height, int image_width,
    int crop_height, int crop_width, int depth,
    float *grads_image, void *workspace, size_t workspace_bytes, void *stream);

/* Tuning hook of the default backward (tools/bwd_stage_probe.py, tools/bwd_trace_probe.py): when set to a device buffer of
 * >= 64 + 4 * grid int64 entries the kernel records per-stage / per-workgroup wall-clock stamps there; NULL (default)
 * turns it off.  Not part of the reference's interface. */
void mdt_debug_bwd_timestamps(long long *dev_buf);

/* A/B variant: vectorised zero-fill kernel followed by an fp32 global-atomic scatter
 * (the reference's algorithm, order-nondeterministic). */
int mdt_crop_and_resize_3d_backward_atomic(
    const float *grads, const float *boxes, const int *box_ind,
    int num_boxes, int batch, int image_height, int image_width, int image_zdepth,
    int crop_height, int crop_width, int crop_zdepth, int depth,
    float *grads_image, void *stream);

/* the round-2 single-launch "territory" kernel (csrc/ab/roi_align_bwd.hip), one map, dim 2 or 3 (D, cd ignored for dim 2); MDT_ERR_UNSUPPORTED (-4)
 * outside its LDS budgets */
int mdt_ab_crop_and_resize_backward_territory(int dim, const float *grads, const float *boxes, const int *box_ind, int num_boxes, int batch,
                                              int H, int W, int D, int ch, int cw, int cd, int depth, float *grads_image, void *stream);

/* ---- libmdt_hip_tuning.so (csrc/Makefile: the product sources compiled with -DMDT_TUNING_HOOKS) --------------------------------------
 * The product library libmdt_hip.so holds NO mutable process state and exports neither of the two setters below; the tuning build is the
 * same code plus these hooks, loaded by tools/bwd3_probe.py, tools/fwd_stamp_probe.py and tools/profile_case.py only
 * (medicaldetectiontoolkit_amd/_lib.py: use_tuning_build()).
 * mdt_debug_bwd3: gather-form backward (csrc/roi_align_bwd_v3.hip): dev_buf >= 16 int64 or NULL; dbg bit0 / bit1 make the scatter / zero
 * role return at once (role-by-role timing); wg = traced scatter workgroup.
 * mdt_debug_fwd_stamps: channel-quad forward (csrc/roi_align_fwd.hip): dev_buf >= 4 * grid int64 (grid = num_boxes x channel groups) or
 * NULL; per workgroup the 100 MHz wall clock at its start, after its box / extents, when its first stage has landed in LDS, at its end. */
void mdt_debug_bwd3(long long *dev_buf, int dbg, int wg);
void mdt_debug_fwd_stamps(long long *dev_buf);

#ifdef __cplusplus
}
#endif
#endif
