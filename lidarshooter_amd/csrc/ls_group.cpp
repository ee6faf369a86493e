// ls_group.cpp -- include/lidarshooter_group.h: azimuth-sharded or frame-interleaved LiDAR frames over the GPUs of a
// node, one process per GPU, RCCL for the one collective (loaded with dlopen so that the slot arithmetic below works,
// and is tested, on machines without a GPU or without RCCL).
#include "../../include/lidarshooter_group.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <string>

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load()
    {
        if (lib) return true;
        // the soname first: a process that already holds RCCL (PyTorch brings its own copy) gets that one back
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = "librccl.so.1 not found"; return false; }
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy || !GetErrorString) { err = "librccl lacks an entry point"; return false; }
        return true;
    }
};

Rccl &rccl()
{
    static Rccl r;
    return r;
}

constexpr int kSets = 3;

}  // namespace

struct ls_group {
    uint32_t world = 1, rank = 0, full_turn = 0;
    long pipeline_before = 0;   // the tracer's LS_OPT_PIPELINE as the group found it
    int mode = LS_GROUP_SHARDED;
    ls_tracer *tr = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_collected[kSets] = {};
    bool used[kSets] = {};
    uint32_t capacity = 0, cloud_capacity = 0;
    size_t slot_bytes = 0;
    uint8_t *slot[kSets] = {}, *gathered[kSets] = {}, *local_points = nullptr;
    uint8_t *cloud_points[kSets] = {}, *cloud_hits[kSets] = {};
    uint32_t *cloud_n[kSets] = {};
    std::string err;
};

namespace {

int fail(ls_group *g, int code, const std::string &msg)
{
    g->err = msg;
    return code;
}

#define LSG_HIP(call)                                                                             \
    do {                                                                                          \
        const hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) return fail(g, LS_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define LSG_NCCL(call)                                                                            \
    do {                                                                                          \
        const ncclResult_t r_ = (call);                                                           \
        if (r_ != ncclSuccess) return fail(g, LS_ERR_HIP, std::string(#call) + ": " + rccl().GetErrorString(r_)); \
    } while (0)

}  // namespace

extern "C" {

void ls_group_shard_columns(uint32_t H, uint32_t world, uint32_t rank, uint32_t *first_az, uint32_t *n_az)
{
    if (!world) world = 1;
    const uint32_t base = H / world, rem = H % world;
    if (first_az) *first_az = rank * base + std::min(rank, rem);
    if (n_az) *n_az = base + (rank < rem ? 1u : 0u);
}

uint32_t ls_group_slot_capacity(uint32_t V, uint32_t H, uint32_t world)
{
    uint32_t n = 0;
    ls_group_shard_columns(H, world, 0, nullptr, &n);   // rank 0 has a largest shard
    return V * n;
}

uint64_t ls_group_slot_bytes(uint32_t capacity) { return LS_GROUP_SLOT_HEADER + 16ull * capacity; }

void ls_group_write_slot(void *slot, uint32_t capacity, const ls_hit *hits, uint32_t n)
{
    n = std::min(n, capacity);
    std::memset(slot, 0, LS_GROUP_SLOT_HEADER);
    std::memcpy(slot, &n, 4);
    if (n) std::memcpy(static_cast<uint8_t *>(slot) + LS_GROUP_SLOT_HEADER, hits, 16ull * n);
}

uint32_t ls_group_decode_gathered(const void *gathered, uint32_t world, uint32_t capacity, ls_hit *out_hits)
{
    const uint64_t sb = ls_group_slot_bytes(capacity);
    uint32_t total = 0;
    for (uint32_t r = 0; r < world; ++r) {
        const uint8_t *s = static_cast<const uint8_t *>(gathered) + r * sb;
        uint32_t n;
        std::memcpy(&n, s, 4);
        n = std::min(n, capacity);
        if (n && out_hits) std::memcpy(out_hits + total, s + LS_GROUP_SLOT_HEADER, 16ull * n);
        total += n;
    }
    return total;
}

int ls_group_unique_id(uint8_t id[LS_GROUP_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == LS_GROUP_ID_BYTES, "ncclUniqueId size");
    if (!id) return LS_ERR_INVALID_ARGUMENT;
    if (!rccl().load()) return LS_ERR_NO_DEVICE;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return LS_ERR_HIP;
    std::memcpy(id, &u, sizeof(u));
    return LS_OK;
}

void ls_group_destroy(ls_group *g)
{
    if (!g) return;
    if (g->tr) {
        (void)ls_tracer_synchronize(g->tr);
        (void)ls_tracer_set_output_buffers(g->tr, nullptr, nullptr, nullptr, 0);
        (void)ls_tracer_set_option(g->tr, LS_OPT_PIPELINE, (int)g->pipeline_before);
        (void)ls_tracer_set_stream(g->tr, nullptr);   // back on its own stream before the group's streams go
        if (g->full_turn) (void)ls_tracer_set_shard(g->tr, 0, g->full_turn);
    }
    if (g->comm_stream) (void)hipStreamSynchronize(g->comm_stream);
    if (g->comm && rccl().CommDestroy) (void)rccl().CommDestroy(g->comm);
    for (int i = 0; i < kSets; ++i) {
        if (g->ev_collected[i]) (void)hipEventDestroy(g->ev_collected[i]);
        (void)hipFree(g->slot[i]);
        (void)hipFree(g->gathered[i]);
        (void)hipFree(g->cloud_points[i]);
        (void)hipFree(g->cloud_hits[i]);
        (void)hipFree(g->cloud_n[i]);
    }
    (void)hipFree(g->local_points);
    if (g->comm_stream) (void)hipStreamDestroy(g->comm_stream);
    delete g;
}

int ls_group_create(const uint8_t id[LS_GROUP_ID_BYTES], uint32_t world, uint32_t rank, int mode, ls_tracer *tr, ls_group **out)
{
    if (!out) return LS_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!tr || !world || rank >= world || (mode != LS_GROUP_SHARDED && mode != LS_GROUP_INTERLEAVED)) return LS_ERR_INVALID_ARGUMENT;
    if (mode == LS_GROUP_SHARDED && !id) return LS_ERR_INVALID_ARGUMENT;
    ls_group *g = new ls_group();
    g->world = world;
    g->rank = rank;
    g->mode = mode;
    // Everything that can fail without the tracer comes first (streams, events, RCCL, the communicator, device memory):
    // until `attached` the caller's tracer has not been touched and a failure leaves it exactly as it was.  Afterwards
    // a failure puts it back on its own stream, on the full turn and on its own output buffers before the group's
    // streams are destroyed (ls_group_destroy does that for an attached tracer).
    bool attached = false;
    auto bail = [&](int code) {
        if (!attached) g->tr = nullptr;
        ls_group_destroy(g);
        return code;
    };
    const uint32_t V = ls_total_channels(tr);
    const long h = ls_get_info(tr, LS_INFO_AZIMUTH_COUNT);
    if (h < 2 || !V) return bail(LS_ERR_INVALID_ARGUMENT);
    const uint32_t H = (uint32_t)h;
    g->full_turn = H;
    if (hipStreamCreateWithFlags(&g->comm_stream, hipStreamNonBlocking) != hipSuccess) return bail(LS_ERR_HIP);
    for (int i = 0; i < kSets; ++i)
        if (hipEventCreateWithFlags(&g->ev_collected[i], hipEventDisableTiming) != hipSuccess) return bail(LS_ERR_HIP);
    g->cloud_capacity = V * H;
    uint32_t first = 0, n = H;
    if (mode == LS_GROUP_SHARDED) {
        ls_group_shard_columns(H, world, rank, &first, &n);
        g->capacity = ls_group_slot_capacity(V, H, world);
        g->slot_bytes = (size_t)ls_group_slot_bytes(g->capacity);
        g->cloud_capacity = g->capacity * world;
        if (!rccl().load()) { g->err = rccl().err; return bail(LS_ERR_NO_DEVICE); }
        ncclUniqueId u;
        std::memcpy(&u, id, sizeof(u));
        if (rccl().CommInitRank(&g->comm, (int)world, u, (int)rank) != ncclSuccess) return bail(LS_ERR_HIP);
        if (hipMalloc(reinterpret_cast<void **>(&g->local_points), (size_t)g->capacity * 32) != hipSuccess) return bail(LS_ERR_HIP);
        for (int i = 0; i < kSets; ++i) {
            if (hipMalloc(reinterpret_cast<void **>(&g->slot[i]), g->slot_bytes) != hipSuccess ||
                hipMalloc(reinterpret_cast<void **>(&g->gathered[i]), g->slot_bytes * world) != hipSuccess)
                return bail(LS_ERR_HIP);
            (void)hipMemset(g->slot[i], 0, LS_GROUP_SLOT_HEADER);
        }
    }
    for (int i = 0; i < kSets; ++i)
        if (hipMalloc(reinterpret_cast<void **>(&g->cloud_points[i]), (size_t)g->cloud_capacity * 32) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&g->cloud_hits[i]), (size_t)g->cloud_capacity * 16) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&g->cloud_n[i]), 64) != hipSuccess)
            return bail(LS_ERR_HIP);
    // ---- nothing but the tracer itself can fail from here on
    g->tr = tr;
    attached = true;
    g->pipeline_before = std::max(0l, ls_get_info(tr, LS_INFO_PIPELINE_MODE));
    // (the tracer stays on its own stream: a frame is handed to the collective stream by ls_tracer_order_after_last_frame, and
    // a handle on a caller's stream would order every frame's stream after that stream -- two runtime calls per frame)
    if (ls_tracer_set_stream(tr, nullptr) != LS_OK) { g->err = ls_last_error(tr); return bail(LS_ERR_HIP); }
    // three frames in flight per rank (the library falls back to two on one stream when the device does not give it
    // three concurrent streams): the gather + rebuild of frame f overlap the tracing of f+1 and f+2, per frame the
    // collective stream waits for that frame alone (ls_tracer_order_after_last_frame), never for the tracer as a whole
    if (ls_tracer_set_option(tr, LS_OPT_PIPELINE, 2) != LS_OK) { g->err = ls_last_error(tr); return bail(LS_ERR_HIP); }
    if (ls_tracer_set_shard(tr, first, n) != LS_OK) { g->err = ls_last_error(tr); return bail(LS_ERR_INVALID_ARGUMENT); }
    *out = g;
    return LS_OK;
}

int ls_group_owns_frame(const ls_group *g, uint32_t frame_index)
{
    if (!g) return 0;
    return g->mode == LS_GROUP_SHARDED || frame_index % g->world == g->rank;
}

int ls_group_trace(ls_group *g, uint32_t frame_index)
{
    if (!g) return LS_ERR_INVALID_ARGUMENT;
    if (!ls_group_owns_frame(g, frame_index)) return 1;
    // INTERLEAVED: this rank's own frames rotate over the sets; SHARDED: every frame does
    const uint32_t seq = g->mode == LS_GROUP_SHARDED ? frame_index : frame_index / g->world;
    const int b = (int)(seq % kSets);
    ls_frame f;
    if (g->mode == LS_GROUP_INTERLEAVED) {
        if (ls_tracer_set_output_buffers(g->tr, g->cloud_points[b], g->cloud_hits[b], g->cloud_n[b], g->cloud_capacity) != LS_OK)
            return fail(g, LS_ERR_HIP, ls_last_error(g->tr));
        const int rc = ls_trace_scene_async(g->tr, frame_index, &f);
        if (rc < -1) return fail(g, rc, ls_last_error(g->tr));
        return rc == -1 ? -1 : 0;
    }
    // the set's previous frame (three frames ago) must have left its slot: the gather reads it on the other stream
    if (g->used[b] && ls_tracer_next_frame_waits(g->tr, g->ev_collected[b]) != LS_OK) return fail(g, LS_ERR_HIP, ls_last_error(g->tr));
    if (ls_tracer_set_output_buffers(g->tr, g->local_points, g->slot[b] + LS_GROUP_SLOT_HEADER, reinterpret_cast<uint32_t *>(g->slot[b]),
                                     g->capacity) != LS_OK)
        return fail(g, LS_ERR_HIP, ls_last_error(g->tr));
    const int rc = ls_trace_scene_async(g->tr, frame_index, &f);
    if (rc < -1) return fail(g, rc, ls_last_error(g->tr));
    if (rc == -1) {   // empty scene: nothing was traced, an empty slot travels
        LSG_HIP(hipMemsetAsync(g->slot[b], 0, 4, g->comm_stream));   // (the collective stream is the slot's only reader; its last gather of this slot is behind it there)
    } else if (ls_tracer_order_after_last_frame(g->tr, g->comm_stream) != LS_OK) {   // this frame's slot is complete for the gather; frames in flight go on
        return fail(g, LS_ERR_HIP, ls_last_error(g->tr));
    }
    // the frame's one collective: every rank's slot to every rank (xGMI is fully connected: direct peer writes)
    LSG_NCCL(rccl().AllGather(g->slot[b], g->gathered[b], g->slot_bytes, ncclUint8, g->comm, g->comm_stream));
    if (ls_expand_gathered_hits_on(g->tr, g->comm_stream, g->gathered[b], g->world, g->capacity, g->cloud_points[b], g->cloud_hits[b],
                                   g->cloud_n[b]) != LS_OK)
        return fail(g, LS_ERR_HIP, ls_last_error(g->tr));
    LSG_HIP(hipEventRecord(g->ev_collected[b], g->comm_stream));
    g->used[b] = true;
    return rc == -1 ? -1 : 0;
}

int ls_group_cloud(ls_group *g, uint32_t frame_index, ls_frame *out)
{
    if (!g || !out) return LS_ERR_INVALID_ARGUMENT;
    if (!ls_group_owns_frame(g, frame_index)) return fail(g, LS_ERR_OUT_OF_RANGE, "this rank does not hold that frame");
    const uint32_t seq = g->mode == LS_GROUP_SHARDED ? frame_index : frame_index / g->world;
    const int b = (int)(seq % kSets);
    std::memset(out, 0, sizeof(*out));
    out->frame = frame_index;
    out->n_rays = g->cloud_capacity;
    out->d_points32 = g->cloud_points[b];
    out->d_hits = g->cloud_hits[b];
    out->d_n_points = g->cloud_n[b];
    return LS_OK;
}

long ls_group_download_cloud(ls_group *g, uint32_t frame_index, void *points32, void *hits, uint32_t capacity)
{
    ls_frame f;
    int rc = ls_group_cloud(g, frame_index, &f);
    if (rc != LS_OK) return rc;
    if ((rc = ls_group_synchronize(g)) != LS_OK) return rc;
    uint32_t n = 0;
    LSG_HIP(hipMemcpy(&n, f.d_n_points, 4, hipMemcpyDeviceToHost));
    if (n > capacity) return fail(g, LS_ERR_OUT_OF_RANGE, "host buffers smaller than the cloud");
    if (n && points32) LSG_HIP(hipMemcpy(points32, f.d_points32, (size_t)n * 32, hipMemcpyDeviceToHost));
    if (n && hits) LSG_HIP(hipMemcpy(hits, f.d_hits, (size_t)n * 16, hipMemcpyDeviceToHost));
    return (long)n;
}

int ls_group_synchronize(ls_group *g)
{
    if (!g) return LS_ERR_INVALID_ARGUMENT;
    const int rc = ls_tracer_synchronize(g->tr);
    if (rc != LS_OK) return fail(g, rc, ls_last_error(g->tr));
    LSG_HIP(hipStreamSynchronize(g->comm_stream));
    return LS_OK;
}

const char *ls_group_last_error(const ls_group *g) { return g ? g->err.c_str() : "null group"; }

}  // extern "C"
