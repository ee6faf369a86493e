// ls_device.h -- device-side helpers shared by the kernel files: the float operation sequences
// that must stay identical to oracle/ls_oracle.c (cross / dot with explicit fused multiply-adds,
// the Embree 3.13.4 Moeller-Trumbore test).  Compiled with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ls {
namespace {

constexpr int kBlock = 256;

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
// Embree common/math/vec3.h: cross = (msub(a.y,b.z,a.z*b.y), msub(a.z,b.x,a.x*b.z), msub(a.x,b.y,a.y*b.x))
__device__ __forceinline__ V3 cross_fma(V3 a, V3 b)
{
    return {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
// dot = madd(a.x,b.x,madd(a.y,b.y,a.z*b.z))
__device__ __forceinline__ float dot_fma(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ float xor_sign(float f, uint32_t s) { return __uint_as_float(__float_as_uint(f) ^ s); }

__device__ __forceinline__ float safe_inv(float d)
{
    // |d| < 1e-30 -> +-1e-30 keeps 1/d finite (azimuth 0 has dir.y == 0 exactly)
    return 1.0f / (fabsf(d) < 1e-30f ? copysignf(1e-30f, d) : d);
}

// Embree 3.13.4 Moeller-Trumbore test (triangle_intersector_moeller.h) for a ray from the origin,
// against a record holding v0, e1 = v0-v1, e2 = v2-v0 and NgC = dot(cross(e2,e1), v0).
// tnear = 0 (strict), no upper bound here: the caller keeps the closest.
__device__ __forceinline__ bool tri_test(V3 d, V3 v0, V3 e1, V3 e2, float NgC, float &t)
{
    const V3 Ng = cross_fma(e2, e1);
    const V3 R = cross_fma(v0, d);
    const float den = dot_fma(Ng, d);
    const float absDen = fabsf(den);
    const uint32_t sgn = __float_as_uint(den) & 0x80000000u;
    const float U = xor_sign(dot_fma(R, e2), sgn);
    const float V = xor_sign(dot_fma(R, e1), sgn);
    const float T = xor_sign(NgC, sgn);
    bool ok = (den != 0.0f) & (U >= 0.0f) & (V >= 0.0f) & (U + V <= absDen) & (0.0f < T);
    if (ok) {
        t = T / absDen;
        // a quotient that overflowed (coordinates of 1e18 metres: Ng . v0 is past FLT_MAX) is no hit: the oracle keeps a hit
        // only if t < its running closest, which starts at +inf (ls_oracle.c:closest_brute) -- so nothing is ever hit "at
        // infinity" (tests/test_gpu_parity.py::test_non_finite_and_huge_vertices)
        ok = t < INFINITY;
    }
    return ok;
}


// Vertex into the sensor frame: p' = Rinv * ((A * v) - t), the reference's operation order
// (MeshTransformer.cpp:176-195 Eigen Affine3f * Vector3f, then LidarDevice.cpp:383-391).
__device__ __forceinline__ V3 xform_vertex(const Affine &m, const uint8_t *rec)
{
    const float *p = reinterpret_cast<const float *>(rec);
    const float x = p[0], y = p[1], z = p[2];
    float q[3], o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        q[i] = ((m.a[4 * i + 0] * x + m.a[4 * i + 1] * y) + m.a[4 * i + 2] * z) + m.a[4 * i + 3];
    const float a = q[0] - m.t[0], b = q[1] - m.t[1], c = q[2] - m.t[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = (m.rinv[3 * i + 0] * a + m.rinv[3 * i + 1] * b) + m.rinv[3 * i + 2] * c;
    return {o[0], o[1], o[2]};
}

// The same with A == identity exactly: A*v reproduces v (1*x, +0*y, +0*z, +0 are exact; only the sign
// of a zero coordinate can differ, which no comparison or product downstream can see), so the first
// matrix product is skipped.
__device__ __forceinline__ V3 xform_vertex_sensor_only(const Affine &m, const uint8_t *rec)
{
    const float *p = reinterpret_cast<const float *>(rec);
    const float a = p[0] - m.t[0], b = p[1] - m.t[1], c = p[2] - m.t[2];
    float o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = (m.rinv[3 * i + 0] * a + m.rinv[3 * i + 1] * b) + m.rinv[3 * i + 2] * c;
    return {o[0], o[1], o[2]};
}

}  // namespace
}  // namespace ls
