/* san_oracle.c -- oracle/ls_oracle.c under AddressSanitizer / UBSan and, for its threaded tracers, ThreadSanitizer (CPU build
 * only).  A small relief grid under a 32 x 150 raster: brute force and the threaded binned-SAH BVH tracer must agree bit for
 * bit; the build's parallel top-level splits are forced by the size. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct lso_bvh lso_bvh;
void lso_ray_dirs(const float *vertical_deg, uint32_t V, float begin, float end, uint32_t count, float *dirs);
void lso_trace_bruteforce(const float *dirs, uint32_t nrays, const float *verts, const uint32_t *tris, uint32_t ntris, float *t, uint32_t *gid, int nthreads);
lso_bvh *lso_bvh_build(const float *verts, const uint32_t *tris, uint32_t ntris, int nthreads);
void lso_bvh_trace(const lso_bvh *b, const float *dirs, uint32_t nrays, float *t, uint32_t *gid, int nthreads, uint64_t *stats);
void lso_bvh_free(lso_bvh *b);

int main(int argc, char **argv)
{
    const int cx = argc > 1 ? atoi(argv[1]) : 300, cy = argc > 2 ? atoi(argv[2]) : 120;   /* 72 000 triangles: above PAR_MIN_PRIMS */
    const uint32_t nv = (uint32_t)(cx + 1) * (uint32_t)(cy + 1), nt = 2u * (uint32_t)cx * (uint32_t)cy;
    float *v = (float *)malloc(sizeof(float) * 3 * nv);
    uint32_t *tr = (uint32_t *)malloc(sizeof(uint32_t) * 3 * nt);
    for (int j = 0; j <= cy; ++j)
        for (int i = 0; i <= cx; ++i) {
            const float x = -40.f + 80.f * (float)i / (float)cx, y = -40.f + 80.f * (float)j / (float)cy;
            float *p = v + 3 * ((size_t)j * (cx + 1) + i);
            p[0] = x; p[1] = y; p[2] = -2.0f + 0.25f * sinf(0.35f * x) * cosf(0.27f * y);
        }
    uint32_t k = 0;
    for (int j = 0; j < cy; ++j)
        for (int i = 0; i < cx; ++i) {
            const uint32_t a = (uint32_t)j * (cx + 1) + i, b = a + 1, c = a + cx + 1, d = c + 1;
            tr[3 * k] = a; tr[3 * k + 1] = b; tr[3 * k + 2] = d; ++k;
            tr[3 * k] = a; tr[3 * k + 1] = d; tr[3 * k + 2] = c; ++k;
        }
    const uint32_t V = 32, H = 150, n = V * H;
    float vertical[32];
    for (uint32_t i = 0; i < V; ++i) vertical[i] = 15.0f - (float)i;
    float *dirs = (float *)malloc(sizeof(float) * 3 * n);
    lso_ray_dirs(vertical, V, 0.0f, 360.0f, H, dirs);
    float *t0 = (float *)malloc(4 * n), *t1 = (float *)malloc(4 * n);
    uint32_t *g0 = (uint32_t *)malloc(4 * n), *g1 = (uint32_t *)malloc(4 * n);
    lso_trace_bruteforce(dirs, n, v, tr, nt, t0, g0, 4);
    lso_bvh *b = lso_bvh_build(v, tr, nt, 4);
    uint64_t stats[4] = {0, 0, 0, 0};
    lso_bvh_trace(b, dirs, n, t1, g1, 4, stats);
    lso_bvh_free(b);
    uint32_t hits = 0, diff = 0;
    for (uint32_t r = 0; r < n; ++r) {
        hits += g0[r] != 0xFFFFFFFFu;
        diff += g0[r] != g1[r] || memcmp(&t0[r], &t1[r], 4) != 0;
    }
    printf("oracle: %u triangles, %u rays, %u hits, %u rays differ between brute force and the BVH tracer\n", nt, n, hits, diff);
    free(v); free(tr); free(dirs); free(t0); free(t1); free(g0); free(g1);
    return diff || !hits ? 1 : 0;
}
