/*
 * ls_oracle.c -- CPU ORACLE for the lidarshooter tracer hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (lidarshooter_amd/)
 * never links, imports or calls anything in this directory.
 *
 * It restates, in plain scalar C, the arithmetic of the reference's per-frame path
 * (paths relative to /root/reference/ros_ws/src/lidarshooter/src/):
 *   - sensor pose            LidarDevice.cpp:758-822 (quaternion -> R, Rinv via Eigen 3.4)
 *   - ray tables/directions  LidarDevice.cpp:294-342, :587-618, :824-845
 *   - vertex transform       MeshTransformer.cpp:142-205, :467-477; LidarDevice.cpp:383-391
 *   - closest hit            EmbreeTracer.cpp:297-367, :472-480 (rtcIntersect16)
 *   - hit -> 32-byte point   EmbreeTracer.cpp:338-352; XYZIRBytes.cpp:24-40
 *
 * Third-party arithmetic that is NOT under /root/reference: the BVH build, traversal and
 * ray/triangle test live in Embree 3.13.4 (upstream binary tarball, .devcontainer/Dockerfile:24-27).
 * Embree is absent from this image, so its *published* triangle test is restated here from the
 * Embree 3 sources as recalled (kernels/geometry/triangle_intersector_moeller.h,
 * MoellerTrumboreIntersectorK; kernels/geometry/trianglev.h / triangle.h for e1,e2,Ng):
 *     e1 = v0-v1, e2 = v2-v0, Ng = cross(e2,e1)
 *     C = v0-org, R = cross(C,dir), den = dot(Ng,dir), absDen = |den|, sgn = signbit(den)
 *     U = dot(R,e2)^sgn, V = dot(R,e1)^sgn, T = dot(Ng,C)^sgn
 *     hit  <=>  den != 0, U >= 0, V >= 0, U+V <= absDen, absDen*tnear < T <= absDen*tfar
 *     t = T/absDen      (Embree uses a Newton-refined rcp; an exact IEEE divide is used here)
 *   with Embree's AVX2 vector helpers  cross(a,b) = (msub(a.y,b.z,a.z*b.y), ...)  and
 *   dot(a,b) = madd(a.x,b.x,madd(a.y,b.y,a.z*b.z)), i.e. fused multiply-adds.
 * Closest hit: minimum t over all triangles of all geometries; equal-t ties are broken by the
 * lowest (geomID, primID) -- Embree's winner there is traversal-order dependent and cannot be
 * known offline, so the rule is ours.  Independent of any BVH.
 *
 * PARITY PINNING: the reference's own tests pin only counts (98/162 mesh, 4800 rays,
 * 1668 / 1781 / 0 hit points); tests/test_oracle.py checks this oracle against all of them.
 * Hit t / XYZ / triangle ids are NOT pinned by any reference-owned vector ("parity unpinned"
 * beyond hit counts).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma).  All float arithmetic below is
 * written so that every operation rounds exactly once (explicit fmaf where Embree fuses);
 * the HIP kernels use the same operation sequence, which is what makes bit-exact id parity
 * testable.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>

static double lso_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define LSO_INVALID 0xFFFFFFFFu

/* ------------------------------------------------------------------------------------------
 * a-1  Sensor pose.  LidarDevice.cpp:800-813: q = (qw,qx,qy,qz) is used as given (NOT
 * normalised); R = q.toRotationMatrix(); Rinv = R.inverse() (Eigen general 3x3 inverse,
 * cofactor form, float).  Matrices are row-major float[9].
 * ------------------------------------------------------------------------------------------ */
void lso_pose_from_quat(float qw, float qx, float qy, float qz, float *R, float *Rinv)
{
    /* Eigen/src/Geometry/Quaternion.h, QuaternionBase::toRotationMatrix() */
    const float tx = 2.0f * qx, ty = 2.0f * qy, tz = 2.0f * qz;
    const float twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const float txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const float tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
    R[3] = txy + twz;          R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = 1.0f - (txx + tyy);

    /* Eigen/src/LU/InverseImpl.h, compute_inverse<Matrix3f>: cofactor expansion */
#define M(i, j) R[3 * (i) + (j)]
#define COF(i, j) (M(((i) + 1) % 3, ((j) + 1) % 3) * M(((i) + 2) % 3, ((j) + 2) % 3) - \
                   M(((i) + 1) % 3, ((j) + 2) % 3) * M(((i) + 2) % 3, ((j) + 1) % 3))
    const float c00 = COF(0, 0), c10 = COF(1, 0), c20 = COF(2, 0);
    const float det = (c00 * M(0, 0) + c10 * M(1, 0)) + c20 * M(2, 0);
    const float invdet = 1.0f / det;
    Rinv[0] = c00 * invdet;       Rinv[1] = c10 * invdet;       Rinv[2] = c20 * invdet;
    Rinv[3] = COF(0, 1) * invdet; Rinv[4] = COF(1, 1) * invdet; Rinv[5] = COF(2, 1) * invdet;
    Rinv[6] = COF(0, 2) * invdet; Rinv[7] = COF(1, 2) * invdet; Rinv[8] = COF(2, 2) * invdet;
#undef COF
#undef M
}

/* ------------------------------------------------------------------------------------------
 * a-2  Ray tables.  LidarDevice.cpp:611  step = (end-begin)/(count-1)  (float / unsigned->float)
 * LidarDevice.cpp:306-316 per ray (channel-major, r = v*H + h):
 *     prePhi = begin + step*float(h)                    (float)
 *     theta  = float((90.0 - chi_v) * M_PI / 180.0)     (double expression, rounded once)
 *     phi    = float(prePhi * M_PI / 180.0)
 *     d      = (sinf(theta)*cosf(phi), sinf(theta)*sinf(phi), cosf(theta))
 * Only V+H distinct angles exist, so the four tables below hold every libm value a frame needs.
 * ------------------------------------------------------------------------------------------ */
float lso_azimuth_step(float begin, float end, uint32_t count)
{
    return (end - begin) / (float)(count - 1u);
}

void lso_ray_tables(const float *vertical_deg, uint32_t V, float begin, float end, uint32_t count,
                    float *sin_theta, float *cos_theta, float *sin_phi, float *cos_phi)
{
    const float step = lso_azimuth_step(begin, end, count);
    for (uint32_t v = 0; v < V; ++v) {
        const float preChi = vertical_deg[v];
        const float theta = (float)((90.0 - (double)preChi) * M_PI / 180.0);
        sin_theta[v] = sinf(theta);
        cos_theta[v] = cosf(theta);
    }
    for (uint32_t h = 0; h < count; ++h) {
        const float prePhi = begin + step * (float)h;
        const float phi = (float)((double)prePhi * M_PI / 180.0);
        sin_phi[h] = sinf(phi);
        cos_phi[h] = cosf(phi);
    }
}

/* dirs: float[3*V*H], channel-major */
void lso_ray_dirs(const float *vertical_deg, uint32_t V, float begin, float end, uint32_t count,
                  float *dirs)
{
    float *st = (float *)malloc(sizeof(float) * (2 * (size_t)V + 2 * (size_t)count));
    float *ct = st + V, *sp = ct + V, *cp = sp + count;
    lso_ray_tables(vertical_deg, V, begin, end, count, st, ct, sp, cp);
    for (uint32_t v = 0; v < V; ++v)
        for (uint32_t h = 0; h < count; ++h) {
            float *d = dirs + 3 * ((size_t)v * count + h);
            d[0] = st[v] * cp[h];
            d[1] = st[v] * sp[h];
            d[2] = ct[v];
        }
    free(st);
}

/* ------------------------------------------------------------------------------------------
 * a-4  Vertex transform.
 * MeshTransformer.cpp:467-477: T = Translation(lin) * Rz(ang.z) * Ry(ang.y) * Rx(ang.x), built
 * by Eigen as ((Tr*Rz)*Ry)*Rx with each AngleAxis turned into a matrix by
 * AngleAxis::toRotationMatrix() (Eigen/src/Geometry/AngleAxis.h).  A[12] is row-major 3x4.
 * ------------------------------------------------------------------------------------------ */
static void angle_axis_unit(float angle, int axis, float *m)
{
    float ax[3] = {0.f, 0.f, 0.f};
    ax[axis] = 1.0f;
    const float s = sinf(angle), c = cosf(angle);
    const float sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
    const float ca[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
    float tmp;
    tmp = ca[0] * ax[1]; m[1] = tmp - sa[2]; m[3] = tmp + sa[2];
    tmp = ca[0] * ax[2]; m[2] = tmp + sa[1]; m[6] = tmp - sa[1];
    tmp = ca[1] * ax[2]; m[5] = tmp - sa[0]; m[7] = tmp + sa[0];
    m[0] = ca[0] * ax[0] + c;
    m[4] = ca[1] * ax[1] + c;
    m[8] = ca[2] * ax[2] + c;
}

static void mat3_mul(const float *a, const float *b, float *o)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            o[3 * i + j] = (a[3 * i + 0] * b[0 + j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}

void lso_affine_from_components(const float *lin, const float *ang, float *A)
{
    float rx[9], ry[9], rz[9], zy[9], zyx[9];
    angle_axis_unit(ang[0], 0, rx);
    angle_axis_unit(ang[1], 1, ry);
    angle_axis_unit(ang[2], 2, rz);
    mat3_mul(rz, ry, zy);
    mat3_mul(zy, rx, zyx);
    for (int i = 0; i < 3; ++i) {
        A[4 * i + 0] = zyx[3 * i + 0];
        A[4 * i + 1] = zyx[3 * i + 1];
        A[4 * i + 2] = zyx[3 * i + 2];
        A[4 * i + 3] = lin[i];
    }
}

/* MeshTransformer.cpp:176-195: p = T*v (linear*v + translation), then
 * LidarDevice.cpp:383-391: p' = Rinv * (p - (tx,ty,tz)).
 * verts: records of `stride` bytes whose first 12 bytes are x,y,z (float LE). */
void lso_transform_vertices(const void *verts, uint32_t stride, uint32_t n, const float *A,
                            const float *Rinv, const float *t, float *out)
{
    const uint8_t *base = (const uint8_t *)verts;
    for (uint32_t j = 0; j < n; ++j) {
        float p[3], q[3];
        memcpy(p, base + (size_t)j * stride, 12);
        for (int i = 0; i < 3; ++i)
            q[i] = ((A[4 * i + 0] * p[0] + A[4 * i + 1] * p[1]) + A[4 * i + 2] * p[2]) + A[4 * i + 3];
        const float a = q[0] - t[0], b = q[1] - t[1], c = q[2] - t[2];
        for (int i = 0; i < 3; ++i)
            out[3 * (size_t)j + i] = (Rinv[3 * i + 0] * a + Rinv[3 * i + 1] * b) + Rinv[3 * i + 2] * c;
    }
}

/* f-4  Sensor -> world cloud transform.  CloudTransformer.cpp:283-318 (applyInverseTransform):
 * ptrans = T * p (Eigen Affine3f * Vector3f: linear*p + translation), then
 * LidarDevice.cpp:393-401 (originToSensorInverse): p_world = R * ptrans + (tx,ty,tz).
 * points: 32-byte XYZIR records (XYZIRPoint.hpp:11-35); only x,y,z (bytes 0..11) change, the
 * other 20 bytes are copied.  in == out is allowed. */
void lso_cloud_to_world(const uint8_t *points_in, uint32_t n, const float *A, const float *R, const float *t,
                        uint8_t *points_out)
{
    for (uint32_t j = 0; j < n; ++j) {
        float p[3], q[3], w[3];
        memcpy(p, points_in + 32 * (size_t)j, 12);
        for (int i = 0; i < 3; ++i)
            q[i] = ((A[4 * i + 0] * p[0] + A[4 * i + 1] * p[1]) + A[4 * i + 2] * p[2]) + A[4 * i + 3];
        for (int i = 0; i < 3; ++i)
            w[i] = ((R[3 * i + 0] * q[0] + R[3 * i + 1] * q[1]) + R[3 * i + 2] * q[2]) + t[i];
        if (points_out != points_in) memcpy(points_out + 32 * (size_t)j, points_in + 32 * (size_t)j, 32);
        memcpy(points_out + 32 * (size_t)j, w, 12);
    }
}

/* ------------------------------------------------------------------------------------------
 * a-6  Ray/triangle test (Embree 3.13.4 Moeller-Trumbore, see header) and closest hit.
 * ------------------------------------------------------------------------------------------ */
typedef struct { float x, y, z; } v3;

static inline v3 v3_sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
/* embree common/math/vec3.h: cross = (msub(a.y,b.z,a.z*b.y), msub(a.z,b.x,a.x*b.z), msub(a.x,b.y,a.y*b.x)) */
static inline v3 v3_cross(v3 a, v3 b)
{
    v3 r = {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
    return r;
}
/* dot = madd(a.x,b.x,madd(a.y,b.y,a.z*b.z)) */
static inline float v3_dot(v3 a, v3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }

static inline float xor_sign(float f, uint32_t sgn)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    u ^= sgn;
    memcpy(&f, &u, 4);
    return f;
}

/* returns 1 and *t_out on a hit with  tnear < t <= tfar */
/* TEST-ONLY SWITCH (tests/test_oracle.py::test_edge_rule_*): which form of the third edge test is used.
 *   0 (default)  U + V <= absDen            -- the single-ray form, MoellerTrumboreIntersector1, as recalled
 *   1            W = absDen - U - V, W >= 0 -- the packet form (MoellerTrumboreIntersectorK, what rtcIntersect16 of
 *                EmbreeTracer.cpp:472-480 runs), as recalled
 * The two round differently for a ray within an ulp of the v1-v2 edge; which one the un-vendored Embree 3.13.4 binary
 * uses cannot be settled offline, so the tests show that the answer does not depend on it for every scene of
 * BASELINE.json's configs.  Never changed by bench.py, smoke() or the parity tests. */
static int g_edge_rule = 0;
void lso_set_edge_rule(int rule) { g_edge_rule = rule ? 1 : 0; }
int lso_get_edge_rule(void) { return g_edge_rule; }

static inline int tri_test(v3 org, v3 dir, v3 v0, v3 v1, v3 v2, float tnear, float tfar, float *t_out)
{
    const v3 e1 = v3_sub(v0, v1);
    const v3 e2 = v3_sub(v2, v0);
    const v3 Ng = v3_cross(e2, e1);
    const v3 C = v3_sub(v0, org);
    const v3 R = v3_cross(C, dir);
    const float den = v3_dot(Ng, dir);
    const float absDen = fabsf(den);
    uint32_t sgn;
    memcpy(&sgn, &den, 4);
    sgn &= 0x80000000u;
    const float U = xor_sign(v3_dot(R, e2), sgn);
    const float V = xor_sign(v3_dot(R, e1), sgn);
    const float T = xor_sign(v3_dot(Ng, C), sgn);
    if (!(den != 0.0f)) return 0;
    if (!(U >= 0.0f)) return 0;
    if (!(V >= 0.0f)) return 0;
    if (g_edge_rule) {
        const float W = (absDen - U) - V;
        if (!(W >= 0.0f)) return 0;
    } else if (!(U + V <= absDen)) return 0;
    if (!(absDen * tnear < T)) return 0;
    if (!(T <= absDen * tfar)) return 0;
    *t_out = T / absDen;
    return 1;
}

int lso_tri_intersect(const float *org, const float *dir, const float *a, const float *b,
                      const float *c, float *t_out)
{
    v3 o = {org[0], org[1], org[2]}, d = {dir[0], dir[1], dir[2]};
    v3 v0 = {a[0], a[1], a[2]}, v1 = {b[0], b[1], b[2]}, v2 = {c[0], c[1], c[2]};
    return tri_test(o, d, v0, v1, v2, 0.0f, INFINITY, t_out);
}

/* Brute-force closest hit of one ray (origin 0, LidarDevice.cpp:320) over triangles [0,ntris).
 * verts: float[3*nverts] (sensor frame); tris: uint32[3*ntris] indices into verts (already
 * rebased so that the global triangle id `gid` orders triangles by (geomID, primID)). */
static void closest_brute(const float *dir, const float *verts, const uint32_t *tris, uint32_t ntris,
                          float *t_best, uint32_t *gid_best)
{
    const v3 o = {0.f, 0.f, 0.f};
    const v3 d = {dir[0], dir[1], dir[2]};
    float best = INFINITY;
    uint32_t bid = LSO_INVALID;
    for (uint32_t k = 0; k < ntris; ++k) {
        const float *a = verts + 3 * (size_t)tris[3 * k + 0];
        const float *b = verts + 3 * (size_t)tris[3 * k + 1];
        const float *c = verts + 3 * (size_t)tris[3 * k + 2];
        v3 v0 = {a[0], a[1], a[2]}, v1 = {b[0], b[1], b[2]}, v2 = {c[0], c[1], c[2]};
        float t;
        if (tri_test(o, d, v0, v1, v2, 0.0f, INFINITY, &t)) {
            if (t < best) { best = t; bid = k; } /* ascending k: equal t keeps the lowest id */
        }
    }
    *t_best = (bid == LSO_INVALID) ? -1.0f : best;
    *gid_best = bid;
}

typedef struct {
    const float *dirs; const float *verts; const uint32_t *tris; uint32_t ntris;
    float *t; uint32_t *gid; uint32_t r0, r1;
} brute_job;

static void *brute_worker(void *p)
{
    brute_job *j = (brute_job *)p;
    for (uint32_t r = j->r0; r < j->r1; ++r)
        closest_brute(j->dirs + 3 * (size_t)r, j->verts, j->tris, j->ntris, j->t + r, j->gid + r);
    return NULL;
}

/* t[r] = hit distance or -1; gid[r] = global triangle id or 0xFFFFFFFF */
void lso_trace_bruteforce(const float *dirs, uint32_t nrays, const float *verts, const uint32_t *tris,
                          uint32_t ntris, float *t, uint32_t *gid, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    brute_job jobs[256];
    for (int i = 0; i < nthreads; ++i) {
        jobs[i] = (brute_job){dirs, verts, tris, ntris, t, gid,
                              (uint32_t)((uint64_t)nrays * i / nthreads),
                              (uint32_t)((uint64_t)nrays * (i + 1) / nthreads)};
        pthread_create(&th[i], NULL, brute_worker, &jobs[i]);
    }
    for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
}

/* ------------------------------------------------------------------------------------------
 * CPU BVH tracer used (a) as the cpu_baseline of bench.py and (b) to check full-size GPU
 * results.  Binned-SAH BVH2, leaves <= 4 triangles, single-ray stack traversal, near child
 * first.  Triangle boxes are padded by 2^-16 of the largest |coordinate| and the box test culls
 * with `tnear_box <= t_best`, so the result equals lso_trace_bruteforce bit for bit (same tri_test,
 * same tie-break); tests/test_oracle.py checks that.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float lo[3], hi[3];
    uint32_t left;  /* internal: index of left child (right = left+1); leaf: first prim slot */
    uint32_t count; /* 0 = internal, else number of prims */
} cpu_node;

typedef struct {
    cpu_node *nodes; uint32_t nnodes;
    uint32_t *prim;     /* permutation: slot -> global triangle id */
    float *tv;          /* 9 floats per slot: v0,v1,v2 */
    uint32_t ntris;
    uint32_t *scratch;  /* build only: partition buffer for the parallel top-level splits */
    int nthreads;       /* build only */
} lso_bvh;

typedef struct { float lo[3], hi[3], c[3]; } prim_info;

static void box_init(float *lo, float *hi) { for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; } }
static void box_grow(float *lo, float *hi, const float *l2, const float *h2)
{
    for (int a = 0; a < 3; ++a) { if (l2[a] < lo[a]) lo[a] = l2[a]; if (h2[a] > hi[a]) hi[a] = h2[a]; }
}
static float box_area(const float *lo, const float *hi)
{
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0.f) return 0.f;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

#define NBINS 16
#define PAR_MIN_PRIMS 65536u  /* nodes at least this large are split by all threads together */
#define PAR_MAX_THREADS 64

typedef struct {
    float lo[3], hi[3], clo[3], chi[3];                       /* phase 0: bounds */
    float blo[3][NBINS][3], bhi[3][NBINS][3]; uint32_t bc[3][NBINS];  /* phase 1: bins per axis */
    uint32_t nleft;                                           /* phase 2: partition counts */
} par_local;

typedef struct {
    const prim_info *pi; uint32_t *idx, *scratch; uint32_t first, n; int P, phase;
    float clo[3], scale[3]; int axis, split; uint32_t left_off[PAR_MAX_THREADS], right_off[PAR_MAX_THREADS];
    par_local loc[PAR_MAX_THREADS];
} par_ctx;

typedef struct { par_ctx *c; int t; } par_arg;

static inline int bin_of(float c, float clo, float scale)
{
    int k = (int)((c - clo) * scale);
    if (k >= NBINS) k = NBINS - 1;
    if (k < 0) k = 0;
    return k;
}

static void *par_worker(void *p)
{
    par_arg *a = (par_arg *)p;
    par_ctx *c = a->c;
    const uint32_t i0 = c->first + (uint32_t)((uint64_t)c->n * a->t / c->P);
    const uint32_t i1 = c->first + (uint32_t)((uint64_t)c->n * (a->t + 1) / c->P);
    par_local *L = &c->loc[a->t];
    if (c->phase == 0) {
        box_init(L->lo, L->hi);
        box_init(L->clo, L->chi);
        for (uint32_t i = i0; i < i1; ++i) {
            const prim_info *q = &c->pi[c->idx[i]];
            box_grow(L->lo, L->hi, q->lo, q->hi);
            box_grow(L->clo, L->chi, q->c, q->c);
        }
    } else if (c->phase == 1) {
        for (int ax = 0; ax < 3; ++ax)
            for (int k = 0; k < NBINS; ++k) { box_init(L->blo[ax][k], L->bhi[ax][k]); L->bc[ax][k] = 0; }
        for (uint32_t i = i0; i < i1; ++i) {
            const prim_info *q = &c->pi[c->idx[i]];
            for (int ax = 0; ax < 3; ++ax) {
                if (!(c->scale[ax] > 0.f)) continue;
                const int k = bin_of(q->c[ax], c->clo[ax], c->scale[ax]);
                box_grow(L->blo[ax][k], L->bhi[ax][k], q->lo, q->hi);
                L->bc[ax][k]++;
            }
        }
    } else if (c->phase == 2) {
        uint32_t nl = 0;
        for (uint32_t i = i0; i < i1; ++i)
            if (bin_of(c->pi[c->idx[i]].c[c->axis], c->clo[c->axis], c->scale[c->axis]) <= c->split) ++nl;
        L->nleft = nl;
    } else if (c->phase == 3) {
        uint32_t l = c->left_off[a->t], r = c->right_off[a->t];
        for (uint32_t i = i0; i < i1; ++i) {
            const uint32_t id = c->idx[i];
            if (bin_of(c->pi[id].c[c->axis], c->clo[c->axis], c->scale[c->axis]) <= c->split) c->scratch[l++] = id;
            else c->scratch[r++] = id;
        }
    } else {
        memcpy(c->idx + i0, c->scratch + i0, sizeof(uint32_t) * (i1 - i0));
    }
    return NULL;
}

static void par_run(par_ctx *c, int phase)
{
    pthread_t th[PAR_MAX_THREADS];
    par_arg args[PAR_MAX_THREADS];
    c->phase = phase;
    for (int t = 0; t < c->P; ++t) { args[t].c = c; args[t].t = t; pthread_create(&th[t], NULL, par_worker, &args[t]); }
    for (int t = 0; t < c->P; ++t) pthread_join(th[t], NULL);
}

typedef struct { lso_bvh *b; const prim_info *pi; uint32_t *idx; uint32_t node, first, n; int par; } build_job;
static void build_rec(lso_bvh *b, const prim_info *pi, uint32_t *idx, uint32_t node, uint32_t first, uint32_t n, int par);
static void *build_thread(void *p)
{
    build_job *j = (build_job *)p;
    build_rec(j->b, j->pi, j->idx, j->node, j->first, j->n, j->par);
    return NULL;
}

/* par = remaining levels at which the two children are built by two threads */
static void build_rec(lso_bvh *b, const prim_info *pi, uint32_t *idx, uint32_t node, uint32_t first, uint32_t n, int par)
{
    cpu_node *nd = &b->nodes[node];
    float clo[3], chi[3];
    box_init(nd->lo, nd->hi);
    box_init(clo, chi);
    /* large nodes (the top of the tree) are split by all threads together; the arithmetic and the
     * resulting tree are the same as the serial path below up to the order of primitives in a side */
    par_ctx *pc = NULL;
    if (n >= PAR_MIN_PRIMS && b->nthreads > 1 && b->scratch) {
        pc = (par_ctx *)malloc(sizeof(par_ctx));
        pc->pi = pi; pc->idx = idx; pc->scratch = b->scratch; pc->first = first; pc->n = n;
        pc->P = b->nthreads < 16 ? b->nthreads : 16;  /* thread creation costs ~0.1 ms: keep the fan-out modest */
        par_run(pc, 0);
        for (int t = 0; t < pc->P; ++t) {
            box_grow(nd->lo, nd->hi, pc->loc[t].lo, pc->loc[t].hi);
            box_grow(clo, chi, pc->loc[t].clo, pc->loc[t].chi);
        }
        for (int a = 0; a < 3; ++a) {
            pc->clo[a] = clo[a];
            pc->scale[a] = (chi[a] - clo[a] > 0.f) ? (float)NBINS / (chi[a] - clo[a]) : 0.f;
        }
        par_run(pc, 1);
    } else {
        for (uint32_t i = first; i < first + n; ++i) {
            const prim_info *p = &pi[idx[i]];
            box_grow(nd->lo, nd->hi, p->lo, p->hi);
            box_grow(clo, chi, p->c, p->c);
        }
    }
    if (n <= 4) { nd->left = first; nd->count = n; return; }
    int bestAxis = -1, bestSplit = -1;
    float bestCost = INFINITY;
    for (int a = 0; a < 3; ++a) {
        const float ext = chi[a] - clo[a];
        if (!(ext > 0.f)) continue;
        float blo[NBINS][3], bhi[NBINS][3];
        uint32_t bc[NBINS];
        for (int k = 0; k < NBINS; ++k) { box_init(blo[k], bhi[k]); bc[k] = 0; }
        const float scale = (float)NBINS / ext;
        if (pc) {
            for (int t = 0; t < pc->P; ++t)
                for (int k = 0; k < NBINS; ++k) {
                    box_grow(blo[k], bhi[k], pc->loc[t].blo[a][k], pc->loc[t].bhi[a][k]);
                    bc[k] += pc->loc[t].bc[a][k];
                }
        } else
        for (uint32_t i = first; i < first + n; ++i) {
            const prim_info *p = &pi[idx[i]];
            int k = (int)((p->c[a] - clo[a]) * scale);
            if (k >= NBINS) k = NBINS - 1;
            if (k < 0) k = 0;
            box_grow(blo[k], bhi[k], p->lo, p->hi);
            bc[k]++;
        }
        float rarea[NBINS];
        uint32_t rcount[NBINS];
        float l2[3], h2[3];
        box_init(l2, h2);
        uint32_t c = 0;
        for (int k = NBINS - 1; k >= 1; --k) {
            box_grow(l2, h2, blo[k], bhi[k]);
            c += bc[k];
            rarea[k] = box_area(l2, h2);
            rcount[k] = c;
        }
        box_init(l2, h2);
        c = 0;
        for (int k = 0; k < NBINS - 1; ++k) {
            box_grow(l2, h2, blo[k], bhi[k]);
            c += bc[k];
            if (c == 0 || rcount[k + 1] == 0) continue;
            const float cost = box_area(l2, h2) * (float)c + rarea[k + 1] * (float)rcount[k + 1];
            if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = k; }
        }
    }
    uint32_t mid;
    if (bestAxis < 0) {
        mid = first + n / 2; /* all centroids equal: median split by position */
    } else if (pc) {
        pc->axis = bestAxis; pc->split = bestSplit;
        par_run(pc, 2);
        uint32_t nl = 0;
        for (int t = 0; t < pc->P; ++t) nl += pc->loc[t].nleft;
        uint32_t l = first, r = first + nl;
        for (int t = 0; t < pc->P; ++t) {
            const uint32_t c0 = (uint32_t)((uint64_t)n * t / pc->P), c1 = (uint32_t)((uint64_t)n * (t + 1) / pc->P);
            pc->left_off[t] = l; pc->right_off[t] = r;
            l += pc->loc[t].nleft; r += (c1 - c0) - pc->loc[t].nleft;
        }
        par_run(pc, 3);
        par_run(pc, 4);
        mid = first + nl;
        if (mid == first || mid == first + n) mid = first + n / 2;
    } else {
        const float ext = chi[bestAxis] - clo[bestAxis];
        const float scale = (float)NBINS / ext;
        uint32_t i = first, j = first + n;
        while (i < j) {
            const prim_info *p = &pi[idx[i]];
            int k = (int)((p->c[bestAxis] - clo[bestAxis]) * scale);
            if (k >= NBINS) k = NBINS - 1;
            if (k < 0) k = 0;
            if (k <= bestSplit) ++i;
            else { --j; uint32_t tmp = idx[i]; idx[i] = idx[j]; idx[j] = tmp; }
        }
        mid = i;
        if (mid == first || mid == first + n) mid = first + n / 2;
    }
    free(pc);
    const uint32_t left = __atomic_fetch_add(&b->nnodes, 2u, __ATOMIC_RELAXED);
    nd->left = left;
    nd->count = 0;
    if (par > 0 && n > 8192) {
        pthread_t th;
        build_job job = {b, pi, idx, left, first, mid - first, par - 1};
        if (pthread_create(&th, NULL, build_thread, &job) == 0) {
            build_rec(b, pi, idx, left + 1, mid, first + n - mid, par - 1);
            pthread_join(th, NULL);
            return;
        }
    }
    build_rec(b, pi, idx, left, first, mid - first, 0);
    build_rec(b, pi, idx, left + 1, mid, first + n - mid, 0);
}

lso_bvh *lso_bvh_build(const float *verts, const uint32_t *tris, uint32_t ntris, int nthreads)
{
    const double t_0 = lso_now();
    int par = 0;
    while ((1 << par) < nthreads && par < 8) ++par;
    lso_bvh *b = (lso_bvh *)calloc(1, sizeof(lso_bvh));
    b->ntris = ntris;
    if (ntris == 0) return b;
    prim_info *pi = (prim_info *)malloc(sizeof(prim_info) * ntris);
    uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t) * ntris);
    for (uint32_t k = 0; k < ntris; ++k) {
        box_init(pi[k].lo, pi[k].hi);
        float m = 0.0f;
        for (int c = 0; c < 3; ++c) {
            const float *p = verts + 3 * (size_t)tris[3 * k + c];
            box_grow(pi[k].lo, pi[k].hi, p, p);
            for (int a = 0; a < 3; ++a) if (fabsf(p[a]) > m) m = fabsf(p[a]);
        }
        for (int a = 0; a < 3; ++a) pi[k].c[a] = 0.5f * (pi[k].lo[a] + pi[k].hi[a]);
        /* pad: tri_test accepts rays that miss the exact triangle by rounding error, so the
         * boxes must be a little fatter than the exact bounds for BVH == brute force */
        const float pad = m * 0x1p-16f;
        for (int a = 0; a < 3; ++a) { pi[k].lo[a] -= pad; pi[k].hi[a] += pad; }
        idx[k] = k;
    }
    const double t_a = lso_now();
    b->nodes = (cpu_node *)malloc(sizeof(cpu_node) * (2 * (size_t)ntris + 1));
    b->nnodes = 1;
    b->nthreads = nthreads;
    b->scratch = (nthreads > 1 && ntris >= PAR_MIN_PRIMS) ? (uint32_t *)malloc(sizeof(uint32_t) * ntris) : NULL;
    build_rec(b, pi, idx, 0, 0, ntris, par);
    free(b->scratch);
    b->scratch = NULL;
    const double t_b = lso_now();
    b->prim = idx;
    b->tv = (float *)malloc(sizeof(float) * 9 * (size_t)ntris);
    for (uint32_t s = 0; s < ntris; ++s)
        for (int c = 0; c < 3; ++c)
            memcpy(b->tv + 9 * (size_t)s + 3 * c, verts + 3 * (size_t)tris[3 * idx[s] + c], 12);
    free(pi);
    if (getenv("LSO_BUILD_TIMING"))
        fprintf(stderr, "lso_bvh_build: setup %.1f ms, recursion %.1f ms, gather %.1f ms\n", (t_a - t_0) * 1e3,
                (t_b - t_a) * 1e3, (lso_now() - t_b) * 1e3);
    return b;
}

void lso_bvh_free(lso_bvh *b)
{
    if (!b) return;
    free(b->nodes); free(b->prim); free(b->tv); free(b);
}

uint32_t lso_bvh_node_count(const lso_bvh *b) { return b->nnodes; }

/* conservative slab test for a ray from the origin; returns entry distance via *t0 */
static inline int box_hit(const float *lo, const float *hi, const float *inv, float tbest, float *t0)
{
    float tn = 0.0f, tf = tbest;
    for (int a = 0; a < 3; ++a) {
        const float t1 = lo[a] * inv[a], t2 = hi[a] * inv[a]; /* inv is finite: no NaN */
        const float mn = fminf(t1, t2);
        const float mx = fmaxf(t1, t2);
        tn = fmaxf(tn, mn);
        tf = fminf(tf, mx);
    }
    *t0 = tn;
    return tn <= tf;
}

static void closest_bvh(const lso_bvh *b, const float *dir, float *t_best, uint32_t *gid_best,
                        uint64_t *nnode, uint64_t *ntri)
{
    float best = INFINITY;
    uint32_t bid = LSO_INVALID;
    if (b->ntris == 0) { *t_best = -1.0f; *gid_best = LSO_INVALID; return; }
    float inv[3];
    for (int a = 0; a < 3; ++a) /* |d| < 1e-30 -> 1e-30 keeps inv finite (azimuth 0 has dy == 0) */
        inv[a] = 1.0f / (fabsf(dir[a]) < 1e-30f ? copysignf(1e-30f, dir[a]) : dir[a]);
    const v3 o = {0.f, 0.f, 0.f}, d = {dir[0], dir[1], dir[2]};
    uint32_t stack[128];
    int sp = 0;
    float t0;
    ++*nnode;
    if (box_hit(b->nodes[0].lo, b->nodes[0].hi, inv, best, &t0)) stack[sp++] = 0;
    while (sp) {
        const cpu_node *nd = &b->nodes[stack[--sp]];
        if (nd->count) {
            for (uint32_t s = nd->left; s < nd->left + nd->count; ++s) {
                const float *tv = b->tv + 9 * (size_t)s;
                v3 v0 = {tv[0], tv[1], tv[2]}, v1 = {tv[3], tv[4], tv[5]}, v2 = {tv[6], tv[7], tv[8]};
                float t;
                ++*ntri;
                if (tri_test(o, d, v0, v1, v2, 0.0f, INFINITY, &t)) {
                    const uint32_t id = b->prim[s];
                    if (t < best || (t == best && id < bid)) { best = t; bid = id; }
                }
            }
            continue;
        }
        float ta, tb;
        const int ha = box_hit(b->nodes[nd->left].lo, b->nodes[nd->left].hi, inv, best, &ta);
        const int hb = box_hit(b->nodes[nd->left + 1].lo, b->nodes[nd->left + 1].hi, inv, best, &tb);
        *nnode += 2;
        if (ha && hb) {
            if (ta <= tb) { stack[sp++] = nd->left + 1; stack[sp++] = nd->left; }
            else { stack[sp++] = nd->left; stack[sp++] = nd->left + 1; }
        } else if (ha) stack[sp++] = nd->left;
        else if (hb) stack[sp++] = nd->left + 1;
    }
    *t_best = (bid == LSO_INVALID) ? -1.0f : best;
    *gid_best = bid;
}

typedef struct {
    const lso_bvh *b; const float *dirs; float *t; uint32_t *gid; uint32_t r0, r1;
    uint64_t nnode, ntri;
} bvh_job;

static void *bvh_worker(void *p)
{
    bvh_job *j = (bvh_job *)p;
    for (uint32_t r = j->r0; r < j->r1; ++r)
        closest_bvh(j->b, j->dirs + 3 * (size_t)r, j->t + r, j->gid + r, &j->nnode, &j->ntri);
    return NULL;
}

/* stats (optional, may be NULL): stats[0] = box tests, stats[1] = triangle tests, summed over rays */
void lso_bvh_trace(const lso_bvh *b, const float *dirs, uint32_t nrays, float *t, uint32_t *gid,
                   int nthreads, uint64_t *stats)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    bvh_job jobs[256];
    /* interleave small blocks so threads get equal shares of hit and miss channels */
    for (int i = 0; i < nthreads; ++i) {
        jobs[i] = (bvh_job){b, dirs, t, gid, (uint32_t)((uint64_t)nrays * i / nthreads),
                            (uint32_t)((uint64_t)nrays * (i + 1) / nthreads), 0, 0};
        pthread_create(&th[i], NULL, bvh_worker, &jobs[i]);
    }
    uint64_t a = 0, c = 0;
    for (int i = 0; i < nthreads; ++i) { pthread_join(th[i], NULL); a += jobs[i].nnode; c += jobs[i].ntri; }
    if (stats) { stats[0] = a; stats[1] = c; }
}

/* ------------------------------------------------------------------------------------------
 * a-9  Hit -> 32-byte point.  EmbreeTracer.cpp:340-346: xyz = tfar*dir (float), intensity 64.0;
 * ring = channel index (LidarDeviceKernels.cu:51; the Embree path's constant 0 is a bug, SURVEY
 * appendix).  XYZIRBytes.cpp:24-40: x@0 y@4 z@8 0@12 intensity@16 ring(int32)@20 0@24..31.
 * Points are emitted in ray-index order (the reference's order is thread-interleaved and
 * nondeterministic).  hits (optional): 4 x uint32 per point {ray index, geomID, primID, t bits};
 * geom_first[g] = first global triangle id of geometry slot g (ascending), ngeom slots,
 * geom_ids[g] = the geomID of slot g.
 * ------------------------------------------------------------------------------------------ */
uint32_t lso_pack_points(const float *t, const uint32_t *gid, const float *dirs, uint32_t nrays,
                         uint32_t H, const uint32_t *geom_first, const uint32_t *geom_ids,
                         uint32_t ngeom, uint8_t *points, uint32_t *hits)
{
    uint32_t n = 0;
    for (uint32_t r = 0; r < nrays; ++r) {
        if (gid[r] == LSO_INVALID) continue;
        const float *d = dirs + 3 * (size_t)r;
        const float xyz[3] = {t[r] * d[0], t[r] * d[1], t[r] * d[2]};
        const float intensity = 64.0f;
        const int32_t ring = (int32_t)(r / H);
        uint8_t *p = points + 32 * (size_t)n;
        memset(p, 0, 32);
        memcpy(p + 0, xyz, 12);
        memcpy(p + 16, &intensity, 4);
        memcpy(p + 20, &ring, 4);
        if (hits) {
            uint32_t g = 0;
            while (g + 1 < ngeom && geom_first[g + 1] <= gid[r]) ++g;
            hits[4 * (size_t)n + 0] = r;
            hits[4 * (size_t)n + 1] = geom_ids ? geom_ids[g] : g;
            hits[4 * (size_t)n + 2] = gid[r] - (ngeom ? geom_first[g] : 0);
            memcpy(&hits[4 * (size_t)n + 3], &t[r], 4);
        }
        ++n;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * Walk a BVH that the HIP library built (downloaded with ls_debug_download_bvh; layout in
 * include/lidarshooter_hip.h) with exactly the kernel's per-ray order -- left child first, pending
 * right children on a stack -- to (a) check that the device traversal of that BVH is what the CPU
 * gets from the same arrays and (b) count the node fetches and triangle tests per ray that
 * bench.py's algorithmic-bytes figure uses.
 *   node i (64 B): float L.lo[3]; uint32 left; float L.hi[3]; uint32 right; float R.lo[3]; uint32 0;
 *                  float R.hi[3]; uint32 0.   child ref: bit 31 = leaf k, else node index.
 *   triangle record (48 B): float v0[3]; uint32 gid; float e1[3]; float NgC; float e2[3]; uint32 pad
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float llo[3]; uint32_t left; float lhi[3]; uint32_t right;
    float rlo[3]; uint32_t pad0; float rhi[3]; uint32_t pad1;
} dev_node;
typedef struct { float v0[3]; uint32_t gid; float e1[3]; float NgC; float e2[3]; uint32_t pad; } dev_tri;

#define DEV_LEAF 0x80000000u

static inline int dev_box_hit(const float *lo, const float *hi, const float *inv, float tbest)
{
    float tn = 0.0f, tf = tbest;
    for (int a = 0; a < 3; ++a) {
        const float t1 = lo[a] * inv[a], t2 = hi[a] * inv[a];
        tn = fmaxf(tn, fminf(t1, t2));
        tf = fminf(tf, fmaxf(t1, t2));
    }
    return tn <= tf;
}

void lso_fat_traverse_stats(const void *nodes_, const void *tris_, uint32_t nleaves, uint32_t leaf_size,
                            uint32_t ntris, const float *dirs, uint32_t nrays, float *t_out, uint32_t *gid_out,
                            uint64_t *stats, uint32_t *per_ray_nodes /* optional */)
{
    const dev_node *nodes = (const dev_node *)nodes_;
    const dev_tri *tris = (const dev_tri *)tris_;
    uint64_t nn = 0, nt = 0;
    uint32_t maxsp = 0;
    for (uint32_t r = 0; r < nrays; ++r) {
        const float *dir = dirs + 3 * (size_t)r;
        float inv[3];
        for (int a = 0; a < 3; ++a)
            inv[a] = 1.0f / (fabsf(dir[a]) < 1e-30f ? copysignf(1e-30f, dir[a]) : dir[a]);
        const v3 d = {dir[0], dir[1], dir[2]};
        float best = INFINITY;
        uint32_t bid = LSO_INVALID;
        uint32_t stack[128];
        uint32_t sp = 0;
        uint32_t cur = nleaves ? (nleaves > 1 ? 0u : DEV_LEAF) : LSO_INVALID;
        const uint64_t nn0 = nn;
        while (cur != LSO_INVALID) {
            if (!(cur & DEV_LEAF)) {
                const dev_node *nd = &nodes[cur];
                ++nn;
                const int hl = dev_box_hit(nd->llo, nd->lhi, inv, best);
                const int hr = dev_box_hit(nd->rlo, nd->rhi, inv, best);
                if (hl) {
                    cur = nd->left;
                    if (hr) { stack[sp++] = nd->right; if (sp > maxsp) maxsp = sp; }
                } else if (hr) {
                    cur = nd->right;
                } else {
                    cur = sp ? stack[--sp] : LSO_INVALID;
                }
            }
            while (cur != LSO_INVALID && (cur & DEV_LEAF)) {
                const uint32_t first = (cur & ~DEV_LEAF) * leaf_size;
                const uint32_t last = first + leaf_size < ntris ? first + leaf_size : ntris;
                for (uint32_t s = first; s < last; ++s) {
                    const dev_tri *tr = &tris[s];
                    const v3 v0 = {tr->v0[0], tr->v0[1], tr->v0[2]};
                    const v3 e1 = {tr->e1[0], tr->e1[1], tr->e1[2]};
                    const v3 e2 = {tr->e2[0], tr->e2[1], tr->e2[2]};
                    ++nt;
                    /* tri_test() with org = 0 and e1, e2, dot(Ng,C) taken from the record */
                    const v3 Ng = v3_cross(e2, e1);
                    const v3 R = v3_cross(v0, d);
                    const float den = v3_dot(Ng, d);
                    const float absDen = fabsf(den);
                    uint32_t sgn;
                    memcpy(&sgn, &den, 4);
                    sgn &= 0x80000000u;
                    const float U = xor_sign(v3_dot(R, e2), sgn);
                    const float V = xor_sign(v3_dot(R, e1), sgn);
                    const float T = xor_sign(tr->NgC, sgn);
                    if (den != 0.0f && U >= 0.0f && V >= 0.0f && U + V <= absDen && 0.0f < T) {
                        const float t = T / absDen;
                        if (t < best || (t == best && tr->gid < bid)) { best = t; bid = tr->gid; }
                    }
                }
                cur = sp ? stack[--sp] : LSO_INVALID;
            }
        }
        t_out[r] = (bid == LSO_INVALID) ? -1.0f : best;
        gid_out[r] = bid;
        if (per_ray_nodes) per_ray_nodes[r] = (uint32_t)(nn - nn0);
    }
    if (stats) { stats[0] = nn; stats[1] = nt; stats[2] = maxsp; }
}
