/* TEST INFRASTRUCTURE: drives every entry point of the CPU oracle (mdt_oracle.c, compiled into this binary) on seeded random and
 * edge-case inputs under AddressSanitizer + UndefinedBehaviorSanitizer (the reference ships neither tests nor
 * sanitizer runs; its C glue reads raw pointers unchecked, crop_and_resize_gpu.c).  Exact-size heap buffers: any read or write past an
 * array -- a clamped coordinate that is not, an off-by-one in the mask's column blocks -- stops the run.
 *   make -C oracle sanitize   ->   build/oracle_sanitize   (exit code 0 = clean; run by tests/test_oracle_cpu.py) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void oracle_crop_and_resize_3d_forward(const float *, const float *, const int *, int, int, int, int, int, int, int, int, int, float, float *);
void oracle_crop_and_resize_3d_backward(const float *, const float *, const int *, int, int, int, int, int, int, int, int, int, float *);
void oracle_crop_and_resize_2d_forward(const float *, const float *, const int *, int, int, int, int, int, int, int, float, float *);
void oracle_crop_and_resize_2d_backward(const float *, const float *, const int *, int, int, int, int, int, int, int, float *);
void oracle_nms_mask_3d(const float *, int, float, uint64_t *);
void oracle_nms_mask_2d(const float *, int, float, uint64_t *);
int oracle_gpu_nms_3d(const float *, int, float, int, int64_t *, int64_t *);
int oracle_gpu_nms_2d(const float *, int, float, int, int64_t *, int64_t *);
int oracle_cpu_nms_3d(const float *, int64_t, int64_t, const int64_t *, const float *, float, int64_t *, int64_t *);
int oracle_cpu_nms_2d(const float *, int64_t, int64_t, const int64_t *, const float *, float, int64_t *, int64_t *);

static uint64_t rng_state = 0x9E3779B97F4A7C15ULL;
static double urand(void)
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}

static float *rand_floats(size_t n, double lo, double hi)
{
    float *p = (float *)malloc((n ? n : 1) * sizeof(float));
    for (size_t i = 0; i < n; ++i) p[i] = (float)(lo + (hi - lo) * urand());
    return p;
}

/* boxes: ordinary, spilling outside [0, 1], inverted, degenerate (zero extent), far outside */
static float *make_boxes(int n, int dim)
{
    float *b = (float *)malloc((size_t)(n ? n : 1) * 2 * dim * sizeof(float));
    for (int i = 0; i < n; ++i) {
        float *r = b + (size_t)i * 2 * dim;
        for (int a = 0; a < dim; ++a) {
            double c = urand(), h = 0.02 + 0.3 * urand();
            double lo = c - h, hi = c + h;
            switch (i % 6) {
            case 1: lo -= 0.4; hi += 0.4; break;            /* spills */
            case 2: { double t = lo; lo = hi; hi = t; } break; /* inverted */
            case 3: hi = lo; break;                            /* degenerate */
            case 4: lo += 3.0; hi += 3.0; break;               /* far outside: everything clamps */
            case 5: lo -= 3.0; hi -= 3.0; break;
            default: break;
            }
            const int i1 = (dim == 3) ? (a == 0 ? 0 : (a == 1 ? 1 : 4)) : a;
            const int i2 = (dim == 3) ? (a == 0 ? 2 : (a == 1 ? 3 : 5)) : a + 2;
            r[i1] = (float)lo; r[i2] = (float)hi;
        }
    }
    return b;
}

static int cmp_desc(const void *a, const void *b)
{
    const float fa = ((const float *)a)[0], fb = ((const float *)b)[0];
    return (fa < fb) - (fa > fb);
}

int main(void)
{
    /* ---- RoIAlign 3D / 2D: several shapes incl. P == 1, one-voxel axes, out-of-range box_ind */
    const int shapes3[][9] = {{2, 3, 5, 4, 7, 9, 3, 2, 2}, {1, 1, 1, 1, 1, 4, 1, 1, 1}, {3, 2, 8, 8, 16, 12, 7, 7, 3}, {2, 2, 6, 5, 4, 8, 1, 1, 1}, {1, 2, 3, 3, 3, 0, 2, 2, 2}};
    for (unsigned s = 0; s < sizeof(shapes3) / sizeof(shapes3[0]); ++s) {
        const int B = shapes3[s][0], C = shapes3[s][1], H = shapes3[s][2], W = shapes3[s][3], D = shapes3[s][4], N = shapes3[s][5];
        const int ch = shapes3[s][6], cw = shapes3[s][7], cd = shapes3[s][8];
        float *img = rand_floats((size_t)B * C * H * W * D, -1, 1), *bx = make_boxes(N, 3);
        int *ind = (int *)malloc((size_t)(N ? N : 1) * sizeof(int));
        for (int i = 0; i < N; ++i) ind[i] = (i % 5 == 4) ? (i % 2 ? -1 : B) : (int)(urand() * B) % B;
        float *crops = (float *)malloc(((size_t)N * C * ch * cw * cd + 1) * sizeof(float));
        oracle_crop_and_resize_3d_forward(img, bx, ind, N, B, H, W, D, ch, cw, cd, C, 0.0f, crops);
        float *g = rand_floats((size_t)N * C * ch * cw * cd, -1, 1), *gi = (float *)malloc((size_t)B * C * H * W * D * sizeof(float));
        oracle_crop_and_resize_3d_backward(g, bx, ind, N, B, H, W, D, ch, cw, cd, C, gi);
        free(img); free(bx); free(ind); free(crops); free(g); free(gi);
    }
    const int shapes2[][7] = {{2, 3, 9, 7, 10, 3, 5}, {1, 1, 1, 1, 3, 1, 1}, {4, 4, 24, 20, 16, 7, 7}, {1, 2, 5, 5, 0, 2, 2}};
    for (unsigned s = 0; s < sizeof(shapes2) / sizeof(shapes2[0]); ++s) {
        const int B = shapes2[s][0], C = shapes2[s][1], H = shapes2[s][2], W = shapes2[s][3], N = shapes2[s][4], ch = shapes2[s][5], cw = shapes2[s][6];
        float *img = rand_floats((size_t)B * C * H * W, -1, 1), *bx = make_boxes(N, 2);
        int *ind = (int *)malloc((size_t)(N ? N : 1) * sizeof(int));
        for (int i = 0; i < N; ++i) ind[i] = (i % 7 == 6) ? B + 3 : (int)(urand() * B) % B;
        float *crops = (float *)malloc(((size_t)N * C * ch * cw + 1) * sizeof(float));
        oracle_crop_and_resize_2d_forward(img, bx, ind, N, B, H, W, ch, cw, C, 0.0f, crops);
        float *g = rand_floats((size_t)N * C * ch * cw, -1, 1), *gi = (float *)malloc((size_t)B * C * H * W * sizeof(float));
        oracle_crop_and_resize_2d_backward(g, bx, ind, N, B, H, W, ch, cw, C, gi);
        free(img); free(bx); free(ind); free(crops); free(g); free(gi);
    }
    /* ---- NMS: n around the 64-row block edges, both rules, exact-size mask / keep buffers */
    const int ns[] = {0, 1, 2, 63, 64, 65, 127, 128, 129, 500};
    for (unsigned k = 0; k < sizeof(ns) / sizeof(ns[0]); ++k) {
        for (int dim = 2; dim <= 3; ++dim) {
            const int n = ns[k], st = 2 * dim + 1;
            float *rows = (float *)malloc((size_t)(n ? n : 1) * (st + 1) * sizeof(float));      /* [score, coords..., score] for the sort */
            for (int i = 0; i < n; ++i) {
                float *r = rows + (size_t)i * (st + 1);
                r[0] = (float)urand();
                for (int a = 0; a < dim; ++a) {
                    const double c = 10 + 80 * urand(), h = 2 + 20 * urand();
                    const int i1 = (dim == 3) ? (a == 0 ? 0 : (a == 1 ? 1 : 4)) : a;
                    const int i2 = (dim == 3) ? (a == 0 ? 2 : (a == 1 ? 3 : 5)) : a + 2;
                    r[1 + i1] = (float)(c - h); r[1 + i2] = (float)(c + h);
                }
                r[st] = r[0];
            }
            qsort(rows, (size_t)n, (size_t)(st + 1) * sizeof(float), cmp_desc);
            float *dets = (float *)malloc((size_t)(n ? n : 1) * st * sizeof(float));
            for (int i = 0; i < n; ++i) memcpy(dets + (size_t)i * st, rows + (size_t)i * (st + 1) + 1, (size_t)st * sizeof(float));
            const int cb = (n + 63) / 64;
            uint64_t *mask = (uint64_t *)malloc(((size_t)n * cb + 1) * sizeof(uint64_t));
            if (n > 0) { if (dim == 3) oracle_nms_mask_3d(dets, n, 0.5f, mask); else oracle_nms_mask_2d(dets, n, 0.5f, mask); }
            int64_t *keep = (int64_t *)malloc((size_t)(n ? n : 1) * sizeof(int64_t)), num = -1;
            for (int strict = 0; strict <= 1; ++strict) {
                const int rc = (dim == 3) ? oracle_gpu_nms_3d(dets, n, 0.3f, strict, keep, &num) : oracle_gpu_nms_2d(dets, n, 0.3f, strict, keep, &num);
                if (rc != 0 || num < 0 || num > n) { fprintf(stderr, "gpu_nms dim %d n %d: rc %d num %lld\n", dim, n, rc, (long long)num); return 2; }
            }
            /* cpu rule: original order + order array + areas (+1 convention, pth_nms.py:29-31) */
            int64_t *order = (int64_t *)malloc((size_t)(n ? n : 1) * sizeof(int64_t));
            float *areas = (float *)malloc((size_t)(n ? n : 1) * sizeof(float));
            for (int i = 0; i < n; ++i) {
                order[i] = i;
                const float *d = dets + (size_t)i * st;
                areas[i] = (d[2] - d[0] + 1) * (d[3] - d[1] + 1) * (dim == 3 ? (d[5] - d[4] + 1) : 1.0f);
            }
            num = -1;
            if (dim == 3) oracle_cpu_nms_3d(dets, n, st, order, areas, 0.3f, keep, &num); else oracle_cpu_nms_2d(dets, n, st, order, areas, 0.3f, keep, &num);
            if (num < 0 || num > n) { fprintf(stderr, "cpu_nms dim %d n %d: num %lld\n", dim, n, (long long)num); return 3; }
            free(rows); free(dets); free(mask); free(keep); free(order); free(areas);
        }
    }
    printf("oracle sanitize run clean\n");
    return 0;
}
