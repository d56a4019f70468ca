/*
 * Plain-pointer entry point around the reference's cpu_nms (compiled from
 * /root/reference/cuda_functions/nms_{2D,3D}/src/nms.c where it lies, with the
 * TH shim in this directory).  TEST INFRASTRUCTURE ONLY.
 */
#include <TH/TH.h>

int cpu_nms(THLongTensor *keep_out, THLongTensor *num_out, THFloatTensor *boxes,
            THLongTensor *order, THFloatTensor *areas, float nms_overlap_thresh);

int ref_cpu_nms(long *keep, long *num_out, float *boxes, long n, long dim,
                long *order, float *areas, float thresh)
{
    mdt_th_tensor t_keep = {keep, {n, 0, 0, 0}, 1};
    mdt_th_tensor t_num = {num_out, {1, 0, 0, 0}, 1};
    mdt_th_tensor t_boxes = {boxes, {n, dim, 0, 0}, 2};
    mdt_th_tensor t_order = {order, {n, 0, 0, 0}, 1};
    mdt_th_tensor t_areas = {areas, {n, 0, 0, 0}, 1};
    return cpu_nms(&t_keep, &t_num, &t_boxes, &t_order, &t_areas, thresh);
}
