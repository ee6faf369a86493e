// ls_host_pool.cpp -- host-side bulk work of the C ABI: parallel copies, the expansion of 16-byte compact points into
// 32-byte PointCloud2 records.  No HIP call in this file.
#include "ls_internal.h"
#include "ls_host_pool.h"

#include <algorithm>
#include <emmintrin.h>
#include <cstdlib>
#include <functional>

namespace lsi {

namespace {
const size_t kCopyChunk = (size_t)std::max(16, tune_int("LS_COPY_CHUNK_KB", 512)) << 10;
}

void parallel_copy(void *dst, const void *src, size_t bytes)
{
    if (bytes <= kCopyChunk) { std::memcpy(dst, src, bytes); return; }
    const size_t n = (bytes + kCopyChunk - 1) / kCopyChunk;
    HostPool::get().run(n, [&](size_t i) {
        const size_t off = i * kCopyChunk;
        std::memcpy(static_cast<uint8_t *>(dst) + off, static_cast<const uint8_t *>(src) + off, std::min(kCopyChunk, bytes - off));
    });
}

int host_pool_threads() { return HostPool::get().threads(); }

// XYZIRBytes.cpp:24-40: x@0 y@4 z@8 0@12 intensity@16 ring@20 0@24..31; intensity is the constant 64.0 (EmbreeTracer.cpp:343).
// 16-byte records (x, y, z, ring) in, 32-byte records out; the destination is written once and not read again by this
// code, so the stores bypass the cache when both sides are 16-byte aligned (no read-for-ownership of 8 MB per frame)
void expand_points_range(uint8_t *dst_points32, const uint8_t *compact16, size_t cnt)
{
    const float intensity = 64.0f;
    uint32_t ibits;
    std::memcpy(&ibits, &intensity, 4);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(compact16);
    uint32_t *dst = reinterpret_cast<uint32_t *>(dst_points32);
    if ((reinterpret_cast<uintptr_t>(dst_points32) & 15u) == 0 && (reinterpret_cast<uintptr_t>(compact16) & 15u) == 0) {
        const __m128i keep_xyz = _mm_set_epi32(0, -1, -1, -1), keep_ring = _mm_set_epi32(0, 0, -1, 0);
        const __m128i intensity_lane = _mm_set_epi32(0, 0, 0, (int)ibits);
        for (size_t k = 0; k < cnt; ++k) {
            const __m128i v = _mm_load_si128(reinterpret_cast<const __m128i *>(src + 4 * k));          // x y z ring
            const __m128i ring = _mm_and_si128(_mm_shuffle_epi32(v, _MM_SHUFFLE(0, 0, 3, 0)), keep_ring);   // 0 ring 0 0
            _mm_stream_si128(reinterpret_cast<__m128i *>(dst + 8 * k), _mm_and_si128(v, keep_xyz));   // x y z 0
            _mm_stream_si128(reinterpret_cast<__m128i *>(dst + 8 * k + 4), _mm_or_si128(ring, intensity_lane));   // 64.0 ring 0 0
        }
        _mm_sfence();
        return;
    }
    for (size_t k = 0; k < cnt; ++k) {
        dst[8 * k + 0] = src[4 * k + 0];
        dst[8 * k + 1] = src[4 * k + 1];
        dst[8 * k + 2] = src[4 * k + 2];
        dst[8 * k + 3] = 0u;
        dst[8 * k + 4] = ibits;
        dst[8 * k + 5] = src[4 * k + 3];
        dst[8 * k + 6] = 0u;
        dst[8 * k + 7] = 0u;
    }
}

// (ray, t) -> point: LidarDevice.cpp:310-316's direction factors from the tables, EmbreeTracer.cpp:341-345's xyz = t * dir,
// in k_pack's operation order (t * (sin_theta * cos_phi): two roundings; this file is compiled with -ffp-contract=off).
// The records come in ascending ray order, so the channel is tracked, not divided out per point.
void expand_hits_range(uint8_t *dst_points32, const uint8_t *hits8, size_t cnt, const float *sin_theta, const float *cos_theta,
                       const float *cs_phi, uint32_t V, uint32_t H)
{
    if (!cnt) return;
    const float intensity = 64.0f;
    uint32_t ibits;
    std::memcpy(&ibits, &intensity, 4);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(hits8);
    uint32_t *dst = reinterpret_cast<uint32_t *>(dst_points32);
    uint32_t v = std::min(src[0] / H, V - 1u);   // (a ray number outside the raster cannot come from k_pack; it must not leave the tables either)
    uint64_t row_end = ((uint64_t)v + 1u) * H;
    float st = sin_theta[v], ct = cos_theta[v];
    const bool aligned = (reinterpret_cast<uintptr_t>(dst_points32) & 15u) == 0;
    for (size_t k = 0; k < cnt; ++k) {
        const uint32_t ray = src[2 * k];
        while (ray >= row_end && v + 1u < V) { ++v; row_end += H; st = sin_theta[v]; ct = cos_theta[v]; }
        const uint32_t h = std::min(ray - (uint32_t)(row_end - H), H - 1u);
        float t;
        std::memcpy(&t, &src[2 * k + 1], 4);
        const float x = t * (st * cs_phi[2 * (size_t)h]), y = t * (st * cs_phi[2 * (size_t)h + 1]), z = t * ct;
        if (aligned) {
            _mm_stream_si128(reinterpret_cast<__m128i *>(dst + 8 * k), _mm_castps_si128(_mm_set_ps(0.0f, z, y, x)));
            _mm_stream_si128(reinterpret_cast<__m128i *>(dst + 8 * k + 4), _mm_set_epi32(0, 0, (int)v, (int)ibits));
        } else {
            std::memcpy(&dst[8 * k + 0], &x, 4); std::memcpy(&dst[8 * k + 1], &y, 4); std::memcpy(&dst[8 * k + 2], &z, 4);
            dst[8 * k + 3] = 0u; dst[8 * k + 4] = ibits; dst[8 * k + 5] = v; dst[8 * k + 6] = 0u; dst[8 * k + 7] = 0u;
        }
    }
    if (aligned) _mm_sfence();
}

void pool_run(size_t n, const std::function<void(size_t)> &fn) { HostPool::get().run(n, fn); }

}  // namespace lsi

using namespace lsi;

extern "C" {

int ls_expand_points(void *dst_points32, const void *compact16, uint32_t n_points)
{
    if ((!dst_points32 || !compact16) && n_points) return LS_ERR_INVALID_ARGUMENT;
    const size_t n = n_points, items = (n + kExpandItem - 1) / kExpandItem;
    const std::function<void(size_t)> work = [&](size_t i) {
        expand_points_range(static_cast<uint8_t *>(dst_points32) + 32 * i * kExpandItem, static_cast<const uint8_t *>(compact16) + 16 * i * kExpandItem,
                            std::min(kExpandItem, n - i * kExpandItem));
    };
    HostPool::get().run(items, work);
    return LS_OK;
}

int ls_parallel_copy(void *dst, const void *src, uint64_t bytes)
{
    if ((!dst || !src) && bytes) return LS_ERR_INVALID_ARGUMENT;
    parallel_copy(dst, src, (size_t)bytes);
    return LS_OK;
}

}  // extern "C"
