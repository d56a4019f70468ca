/*
 * Minimal stand-in for the TH tensor C API (torch 0.4's <TH/TH.h>), just large
 * enough to compile the reference's own cuda_functions/nms_{2D,3D}/src/nms.c
 * where it lies (see ../../Makefile, target _ref).  TEST INFRASTRUCTURE ONLY.
 *
 * A "tensor" here is a borrowed pointer plus sizes; nothing is reference code.
 */
#ifndef MDT_ORACLE_TH_SHIM_H
#define MDT_ORACLE_TH_SHIM_H
#include <stdlib.h>
#include <string.h>

typedef struct mdt_th_tensor {
    void *data;
    long size[4];
    int ndim;
} mdt_th_tensor;

typedef mdt_th_tensor THLongTensor;
typedef mdt_th_tensor THFloatTensor;
typedef mdt_th_tensor THByteTensor;

#define THArgCheck(cond, argn, msg) ((void)(cond))

static inline int THLongTensor_isContiguous(const mdt_th_tensor *t) { (void)t; return 1; }
static inline int THFloatTensor_isContiguous(const mdt_th_tensor *t) { (void)t; return 1; }
static inline long THFloatTensor_size(const mdt_th_tensor *t, int dim) { return t->size[dim]; }
static inline long THLongTensor_size(const mdt_th_tensor *t, int dim) { return t->size[dim]; }
static inline long *THLongTensor_data(const mdt_th_tensor *t) { return (long *)t->data; }
static inline float *THFloatTensor_data(const mdt_th_tensor *t) { return (float *)t->data; }
static inline unsigned char *THByteTensor_data(const mdt_th_tensor *t) { return (unsigned char *)t->data; }

static inline THByteTensor *THByteTensor_newWithSize1d(long n)
{
    THByteTensor *t = (THByteTensor *)malloc(sizeof(THByteTensor));
    t->data = malloc((size_t)(n > 0 ? n : 1));
    t->size[0] = n; t->ndim = 1;
    return t;
}
static inline void THByteTensor_fill(THByteTensor *t, unsigned char v) { memset(t->data, v, (size_t)t->size[0]); }
static inline void THByteTensor_free(THByteTensor *t) { free(t->data); free(t); }

#endif
