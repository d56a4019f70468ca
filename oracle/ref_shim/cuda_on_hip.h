/*
 * Force-included (-include) when the reference's CUDA kernel sources are
 * compiled, where they lie, by hipcc for gfx950 (oracle/Makefile target _ref_gpu).
 * It only renames the five CUDA runtime identifiers those files use.
 * TEST INFRASTRUCTURE ONLY: the resulting oracle/_ref/libref_gpu_*.so are the
 * reference kernels themselves, used as a second checker and as the
 * "reference kernel on the same GPU" timing -- never by the product path.
 */
#ifndef MDT_ORACLE_CUDA_ON_HIP_H
#define MDT_ORACLE_CUDA_ON_HIP_H
#include <hip/hip_runtime.h>
#define cudaStream_t hipStream_t
#define cudaError_t hipError_t
#define cudaGetLastError hipGetLastError
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#endif
