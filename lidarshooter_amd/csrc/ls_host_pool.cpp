// ls_host_pool.cpp -- host-side bulk work of the C ABI: parallel copies, the expansion of 16-byte compact points into
// 32-byte PointCloud2 records.  No HIP call in this file.
#include "ls_internal.h"
#include "ls_host_pool.h"

#include <algorithm>
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

}  // namespace lsi

using namespace lsi;

extern "C" {

int ls_expand_points(void *dst_points32, const void *compact16, uint32_t n_points)
{
    if ((!dst_points32 || !compact16) && n_points) return LS_ERR_INVALID_ARGUMENT;
    // XYZIRBytes.cpp:24-40: x@0 y@4 z@8 0@12 intensity@16 ring@20 0@24..31; intensity is the constant 64.0 (EmbreeTracer.cpp:343)
    constexpr size_t kPer = 16384;   // points per work item: 256 KB read, 512 KB written
    const size_t n = n_points, items = (n + kPer - 1) / kPer;
    const float intensity = 64.0f;
    uint32_t ibits;
    std::memcpy(&ibits, &intensity, 4);
    const std::function<void(size_t)> work = [&](size_t i) {
        const uint32_t *src = static_cast<const uint32_t *>(compact16) + 4 * i * kPer;
        uint32_t *dst = static_cast<uint32_t *>(dst_points32) + 8 * i * kPer;
        const size_t cnt = std::min(kPer, n - i * kPer);
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
