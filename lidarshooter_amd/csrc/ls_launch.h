// ls_launch.h -- every kernel launch of the per-frame path goes through launch_k, which normally IS
// hipLaunchKernelGGL.  When the calling thread has a LaunchSink installed (ls_trace.cpp: a frame whose launches are
// being captured into, or replayed from, a HIP graph -- LS_OPT_FRAME_GRAPH), launches on the sink's stream are
// recorded as (function, grid, block, LDS bytes, argument bytes):
//   kCapture   the launch happens (the stream is capturing: it becomes a kernel node) AND is recorded, so that the node
//              can be matched with its record afterwards;
//   kDescribe  nothing is launched: the records are compared with those of the captured graph, the nodes whose
//              arguments changed (a new pose, another output buffer) are patched with hipGraphExecKernelNodeSetParams,
//              and the whole frame goes out as ONE hipGraphLaunch.
// Host cost per frame of three launches + collective + rebuild (tools/micro/graph_launch.hip, MI355X, ROCm 7.2): 9 runtime
// calls 26 - 28 us, one graph launch 6.4 us, 8.0 us with one node patched.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <type_traits>
#include <vector>

namespace ls {

constexpr uint32_t kMaxLaunchArgs = 20;   // arg_off holds n_args + 1 offsets: a kernel takes at most kMaxLaunchArgs - 1 arguments

struct LaunchRecord {
    const void *func = nullptr;
    dim3 grid, block;
    uint32_t shmem = 0;
    uint32_t n_args = 0;
    uint32_t arg_off[kMaxLaunchArgs] = {};   // byte offset of argument i in blob; arg_off[n_args] = blob size
    std::vector<uint8_t> blob;
};

// two records describe the same launch: same grid, block, LDS bytes and argument bytes (padding between arguments included: it
// is zero-filled by pack_args, so two packings of the same values compare equal).  The function is compared by the caller.
inline bool same_launch(const LaunchRecord &a, const LaunchRecord &b)
{
    return a.grid.x == b.grid.x && a.grid.y == b.grid.y && a.grid.z == b.grid.z && a.block.x == b.block.x && a.block.y == b.block.y &&
           a.block.z == b.block.z && a.shmem == b.shmem && a.n_args == b.n_args && a.blob.size() == b.blob.size() &&
           std::memcmp(a.arg_off, b.arg_off, sizeof(uint32_t) * (a.n_args + 1)) == 0 &&
           (a.blob.empty() || std::memcmp(a.blob.data(), b.blob.data(), a.blob.size()) == 0);
}

// hipKernelNodeParams::kernelParams for a record: one pointer per argument, into the record's own blob (valid until the
// record is packed again or destroyed -- hipGraphExecKernelNodeSetParams copies the values out before it returns)
inline void argument_pointers(LaunchRecord &r, void *argv[kMaxLaunchArgs])
{
    for (uint32_t a = 0; a < r.n_args; ++a) argv[a] = r.blob.data() + r.arg_off[a];
}

struct LaunchSink {
    enum Mode { kOff = 0, kCapture = 1, kDescribe = 2 };
    int mode = kOff;
    hipStream_t stream = nullptr;
    size_t n = 0;                     // records of the frame being built (recs keeps its storage across frames)
    std::vector<LaunchRecord> recs;
    LaunchRecord &next()
    {
        if (n == recs.size()) recs.emplace_back();
        return recs[n++];
    }
};

// the sink of the calling thread (set by the library entry points of a handle with a frame graph open)
LaunchSink *&thread_sink();

namespace detail {
inline void pack_args(LaunchRecord &) {}
template <typename A, typename... Rest>
inline void pack_args(LaunchRecord &r, const A &a, const Rest &...rest)
{
    static_assert(std::is_trivially_copyable<A>::value, "kernel arguments are plain data");
    const size_t align = alignof(A) > 16 ? 16 : alignof(A);   // (offsets only order the arguments inside the blob: nothing reads them at their natural alignment)
    size_t at = (r.blob.size() + align - 1) / align * align;
    r.blob.resize(at + sizeof(A));   // (value-initialises the padding: zero)
    std::memcpy(r.blob.data() + at, &a, sizeof(A));
    r.arg_off[r.n_args++] = (uint32_t)at;
    pack_args(r, rest...);
}
}  // namespace detail

template <typename... KArgs, typename... Args>
inline void launch_k(void (*kernel)(KArgs...), dim3 grid, dim3 block, uint32_t shmem, hipStream_t s, Args &&...args)
{
    static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count");
    static_assert(sizeof...(KArgs) < kMaxLaunchArgs, "LaunchRecord::arg_off holds one offset more than there are arguments");
    LaunchSink *sink = thread_sink();
    if (!sink || sink->mode == LaunchSink::kOff || sink->stream != s) {
        hipLaunchKernelGGL(kernel, grid, block, shmem, s, static_cast<KArgs>(args)...);
        return;
    }
    LaunchRecord &r = sink->next();
    r.func = reinterpret_cast<const void *>(kernel);
    r.grid = grid;
    r.block = block;
    r.shmem = shmem;
    r.n_args = 0;
    r.blob.clear();
    detail::pack_args(r, static_cast<KArgs>(args)...);   // converted to the kernel's own parameter types, as a launch would
    r.arg_off[r.n_args] = (uint32_t)r.blob.size();
    if (sink->mode == LaunchSink::kCapture) hipLaunchKernelGGL(kernel, grid, block, shmem, s, static_cast<KArgs>(args)...);
}

}  // namespace ls
