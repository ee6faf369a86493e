// ls_group.cpp -- include/lidarshooter_group.h: azimuth-sharded or frame-interleaved LiDAR frames over the GPUs of a
// node, one process per GPU, RCCL for the one collective (loaded with dlopen so that the slot arithmetic below works,
// and is tested, on machines without a GPU or without RCCL).
#include "../../include/lidarshooter_group.h"
#include "../../include/lidarshooter_hip_debug.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    // optional (the group works without them, with less to report / on the round-3 path)
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, void *) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    std::string err;
    bool is_shim = false;   // tests/shim/librccl_shim.so (reported by ls_group_info: a run on it is labelled, never mistaken for RCCL)
    bool load()
    {
        if (lib) return true;
        // LS_GROUP_RCCL_LIBRARY: the collective library a site wants instead of the one the loader finds (a path; a site's own
        // RCCL build -- and tests/shim/librccl_shim.so, which stands in for RCCL where one device must serve two ranks).  Named
        // and not found is an error, never a silent fall-back on another library.
        const char *forced = std::getenv("LS_GROUP_RCCL_LIBRARY");
        if (forced && *forced) {
            lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!lib) { err = std::string("LS_GROUP_RCCL_LIBRARY=") + forced + ": " + dlerror(); return false; }
        }
        // the soname first: a process that already holds RCCL (PyTorch brings its own copy) gets that one back
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (lib) break;
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!lib) { err = "librccl.so.1 not found"; return false; }
        is_shim = dlsym(lib, "ls_rccl_shim_marker") != nullptr;
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy || !GetErrorString) { err = "librccl lacks an entry point"; return false; }
        CommSplit = reinterpret_cast<decltype(CommSplit)>(dlsym(lib, "ncclCommSplit"));
        Broadcast = reinterpret_cast<decltype(Broadcast)>(dlsym(lib, "ncclBroadcast"));
        CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(lib, "ncclCommCount"));
        CommCuDevice = reinterpret_cast<decltype(CommCuDevice)>(dlsym(lib, "ncclCommCuDevice"));
        GetVersion = reinterpret_cast<decltype(GetVersion)>(dlsym(lib, "ncclGetVersion"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(dlsym(lib, "ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(dlsym(lib, "ncclRecv"));
        return true;
    }
};

Rccl &rccl()
{
    static Rccl r;
    return r;
}

constexpr int kSets = 3;

// what ls_expand_gathered_hits_sized leaves in pinned host memory (ls_kernels.h: GatherStat)
struct GatherStat {
    uint32_t max_count, truncated, epoch, pad[13];
};
static_assert(sizeof(GatherStat) == 64, "one line");

}  // namespace

struct ls_group {
    uint32_t world = 1, rank = 0, full_turn = 0;
    long pipeline_before = 0;   // the tracer's LS_OPT_PIPELINE as the group found it
    int mode = LS_GROUP_SHARDED;
    ls_tracer *tr = nullptr;
    ncclComm_t comm = nullptr;
    // per-set mode (the default): one communicator per buffer set -- comm, then two duplicates of it -- so that the whole of
    // set b's frame (trace, gather, rebuild) lives on ONE stream, the tracer's slot stream b, and is one graph launch
    ncclComm_t comm_dup[kSets - 1] = {};
    bool per_set = false;
    uint32_t arrangement_mine = 0, arrangement_common = 0;   // agree_on_arrangement: this rank's answer, the AND over the ranks
    uint32_t flags = 0;
    long frame_graph_before = 0;           // the tracer's LS_OPT_FRAME_GRAPH as the group found it
    int emit_points_before = 1;            // ... and its LS_OPT_EMIT_POINTS
    bool frame_graph_set = false;          // ... and whether the group changed it
    uint32_t set_frame[kSets] = {};        // which frame each set holds
    bool set_valid[kSets] = {};
    hipStream_t loose[kSets] = {};         // per-set mode: a stream with work the tracer's flush does not know of (empty-scene frames)
    // LS_GROUP_FLAG_SIZED_GATHER: the gather moves the front of every slot -- header + gcap records -- sized from what the
    // set's previous tenant needed (the same number on every rank: it comes out of the gathered headers)
    bool sized = false;
    uint32_t gcap = 0, gstep = 0;          // records that travel per slot now; the granularity of that number
    GatherStat *h_stat[kSets] = {};        // pinned: written by the rebuild of the set's frame
    bool stat_pending[kSets] = {};         // the set's frame will write its stat (nobody has read it yet)
    unsigned long long truncated_frames = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_collected[kSets] = {};
    bool used[kSets] = {};
    uint32_t capacity = 0, cloud_capacity = 0;
    size_t slot_bytes = 0;
    uint8_t *slot[kSets] = {}, *gathered[kSets] = {};
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
        (void)ls_group_synchronize(g);
        if (g->frame_graph_set) {   // the cached frame graphs hold this group's collectives: they go before the communicators do
            (void)ls_frame_graph_reset(g->tr);
            (void)ls_tracer_set_option(g->tr, LS_OPT_FRAME_GRAPH, (int)g->frame_graph_before);
        }
        (void)ls_tracer_set_output_buffers(g->tr, nullptr, nullptr, nullptr, 0);
        (void)ls_tracer_set_option(g->tr, LS_OPT_EMIT_POINTS, g->emit_points_before);
        (void)ls_tracer_set_option(g->tr, LS_OPT_PIPELINE, (int)g->pipeline_before);
        (void)ls_tracer_set_stream(g->tr, nullptr);   // back on its own stream before the group's streams go
        if (g->full_turn) (void)ls_tracer_set_shard(g->tr, 0, g->full_turn);
    }
    if (g->comm_stream) (void)hipStreamSynchronize(g->comm_stream);
    for (ncclComm_t &c : g->comm_dup)
        if (c && rccl().CommDestroy) { (void)rccl().CommDestroy(c); c = nullptr; }
    if (g->comm && rccl().CommDestroy) (void)rccl().CommDestroy(g->comm);
    for (int i = 0; i < kSets; ++i) {
        if (g->ev_collected[i]) (void)hipEventDestroy(g->ev_collected[i]);
        (void)hipFree(g->slot[i]);
        (void)hipFree(g->gathered[i]);
        (void)hipFree(g->cloud_points[i]);
        (void)hipFree(g->cloud_hits[i]);
        (void)hipFree(g->cloud_n[i]);
    }
    for (GatherStat *&st : g->h_stat)
        if (st) { (void)hipHostFree(st); st = nullptr; }
    if (g->comm_stream) (void)hipStreamDestroy(g->comm_stream);
    delete g;
}

namespace {
// One 32-bit word from every rank over the first communicator -> the AND over the ranks (negative: the gather failed).
int gather_and(ls_group *g, uint32_t mine)
{
    uint32_t *d = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&d), 4u * (g->world + 1u)) != hipSuccess) return LS_ERR_HIP;
    std::vector<uint32_t> all(g->world, 0u);
    bool ok = hipMemcpyAsync(d + g->world, &mine, 4, hipMemcpyHostToDevice, g->comm_stream) == hipSuccess &&
              rccl().AllGather(d + g->world, d, 4, ncclUint8, g->comm, g->comm_stream) == ncclSuccess &&
              hipMemcpyAsync(all.data(), d, 4u * g->world, hipMemcpyDeviceToHost, g->comm_stream) == hipSuccess &&
              hipStreamSynchronize(g->comm_stream) == hipSuccess;
    (void)hipFree(d);
    if (!ok) return LS_ERR_HIP;
    uint32_t common = 0x7FFFFFFFu;
    for (uint32_t v : all) common &= v;
    return (int)common;
}

// a second communicator over the same ranks from a fresh id that rank 0 makes and the first communicator broadcasts
// (libraries without ncclCommSplit).  Collective; nullptr when it did not work here.
ncclComm_t duplicate_by_id(ls_group *g, hipStream_t s)
{
    Rccl &R = rccl();
    ncclComm_t dup = nullptr;
    ncclUniqueId u;
    std::memset(&u, 0, sizeof(u));
    if (g->rank == 0 && R.GetUniqueId(&u) != ncclSuccess) std::memset(&u, 0, sizeof(u));   // (the others then fail to join, all alike)
    void *d = nullptr;
    if (hipMalloc(&d, sizeof(u)) != hipSuccess) return nullptr;
    bool ok = hipMemcpyAsync(d, &u, sizeof(u), hipMemcpyHostToDevice, s) == hipSuccess &&
              R.Broadcast(d, d, sizeof(u), ncclUint8, 0, g->comm, s) == ncclSuccess &&
              hipMemcpyAsync(&u, d, sizeof(u), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    (void)hipFree(d);
    if (ok) ok = R.CommInitRank(&dup, (int)g->world, u, (int)g->rank) == ncclSuccess;
    return ok ? dup : nullptr;
}

void drop_duplicates(ls_group *g);

// The communicators of the buffer sets 1 .. kSets - 1.  Every rank issues the SAME collectives whatever happens to it
// locally: HOW a duplicate is made -- ncclCommSplit with one colour, or a fresh id broadcast over the first
// communicator -- is the group's choice, never a rank's, and a rank whose duplicate i failed still attempts duplicate i + 1
// with its peers.  (The old code chose per rank and per call: a rank whose split failed locally went into a broadcast its
// peers were not in.)  Stage 1, if every rank's library has ncclCommSplit: all splits, then the AND of "all of mine exist"
// over the ranks; unless everybody holds everything, everybody drops what it holds and the group goes on to stage 2: all
// duplicates by broadcast ids.  A duplicate that failed HERE in stage 2 is nullptr while the peers may hold their half of
// it: agree_on_arrangement's AND (later, once the tracer's half of the arrangement is known) makes every rank drop all of them.
// -> 1 every duplicate exists on this rank, 0 not, negative: the first communicator itself does not work.
int make_duplicates(ls_group *g)
{
    Rccl &R = rccl();
    const int by_split = gather_and(g, R.CommSplit ? 1u : 0u);
    if (by_split < 0) return by_split;
    if (by_split & 1) {
        bool all = true;
        for (int i = 0; i < kSets - 1; ++i) {
            ncclComm_t dup = nullptr;
            if (R.CommSplit(g->comm, 0, (int)g->rank, &dup, nullptr) != ncclSuccess) dup = nullptr;
            g->comm_dup[i] = dup;
            all = all && dup != nullptr;
        }
        const int everybody = gather_and(g, all ? 1u : 0u);
        if (everybody < 0) return everybody;
        if (everybody & 1) return 1;
        drop_duplicates(g);
    }
    const int can_broadcast = gather_and(g, R.Broadcast ? 1u : 0u);
    if (can_broadcast < 0) return can_broadcast;
    if (!(can_broadcast & 1)) return 0;
    bool all = true;
    for (int i = 0; i < kSets - 1; ++i) {
        g->comm_dup[i] = duplicate_by_id(g, g->comm_stream);
        all = all && g->comm_dup[i] != nullptr;
    }
    return all ? 1 : 0;
}

void drop_duplicates(ls_group *g)
{
    for (ncclComm_t &c : g->comm_dup)
        if (c) { (void)rccl().CommDestroy(c); c = nullptr; }
}

// What every rank decided for itself about the group's arrangement -- `mine`: bit 0 "I hold a communicator per buffer set",
// bit 1 "my tracer runs three frames on three streams" -- gathered over the first communicator: -> the AND over the ranks, or
// negative.  A rank alone in an arrangement would issue its collectives on another communicator than its peers and the group
// would hang at its first frame: one of the two inputs is a timing measurement (ensure_slot_streams'
// calibration), the other a library call that can fail on one rank only.
int agree_on_arrangement(ls_group *g, uint32_t mine)
{
    int common = gather_and(g, mine);
    if (common < 0) return common;
    if (g->flags & LS_GROUP_FLAG_DEBUG_PEER_REFUSES) common = 0;
    return common & 3;
}
}  // namespace

int ls_group_create(const uint8_t id[LS_GROUP_ID_BYTES], uint32_t world, uint32_t rank, int mode, ls_tracer *tr, ls_group **out)
{
    return ls_group_create_opts(id, world, rank, mode, 0u, tr, out);
}

int ls_group_create_opts(const uint8_t id[LS_GROUP_ID_BYTES], uint32_t world, uint32_t rank, int mode, uint32_t flags, ls_tracer *tr,
                         ls_group **out)
{
    if (!out) return LS_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!tr || !world || rank >= world || (mode != LS_GROUP_SHARDED && mode != LS_GROUP_INTERLEAVED)) return LS_ERR_INVALID_ARGUMENT;
    if (mode == LS_GROUP_SHARDED && !id) return LS_ERR_INVALID_ARGUMENT;
    ls_group *g = new ls_group();
    g->world = world;
    g->rank = rank;
    g->mode = mode;
    g->flags = flags;
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
        // per-set mode: a communicator per buffer set (collective: every rank takes this branch or none does -- the flags
        // are the caller's and equal on all ranks).  Without the duplicates the group runs round 3's path.
        if (!(flags & LS_GROUP_FLAG_ONE_COMMUNICATOR)) {
            const int made = make_duplicates(g);
            if (made < 0) { g->err = "the first communicator does not carry a 4-byte all-gather"; return bail(LS_ERR_HIP); }
            g->per_set = made == 1;   // (a partial set is kept until the ranks have agreed: its other halves live on the peers)
        }
        // (the shard's own 32-byte points are never written: LS_OPT_EMIT_POINTS = 0 below, hit buffers alone are installed
        // per frame -- ls_tracer_set_hit_buffers -- and every rank rebuilds the whole frame's points from the gathered records)
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
    g->emit_points_before = ls_get_info(tr, LS_INFO_EMIT_POINTS) == 0 ? 0 : 1;
    if (mode == LS_GROUP_SHARDED && ls_tracer_set_option(tr, LS_OPT_EMIT_POINTS, 0) != LS_OK) { g->err = ls_last_error(tr); return bail(LS_ERR_HIP); }
    // (hit buffers alone from the start -- every frame installs its own set's -- so that LS_OPT_EMIT_POINTS = 1 is refused for
    // as long as the group holds the tracer)
    if (mode == LS_GROUP_SHARDED && ls_tracer_set_hit_buffers(tr, g->slot[0] + LS_GROUP_SLOT_HEADER, reinterpret_cast<uint32_t *>(g->slot[0]), g->capacity) != LS_OK) {
        g->err = ls_last_error(tr);
        return bail(LS_ERR_HIP);
    }
    bool three_streams = ls_get_info(tr, LS_INFO_PIPELINE_MODE) == 2;
    if (mode == LS_GROUP_SHARDED && !(flags & LS_GROUP_FLAG_ONE_COMMUNICATOR)) {
        // the arrangement is the GROUP's, not a rank's: every rank takes what all of them can do (agree_on_arrangement)
        const int common = agree_on_arrangement(g, (g->per_set ? 1u : 0u) | (three_streams ? 2u : 0u));
        if (common < 0) { g->err = "the ranks could not agree on the group's arrangement (all-gather over the first communicator failed)"; return bail(LS_ERR_HIP); }
        g->arrangement_mine = (g->per_set ? 1u : 0u) | (three_streams ? 2u : 0u);
        g->arrangement_common = (uint32_t)common;
        if (!(common & 1)) drop_duplicates(g);
        g->per_set = (common & 1) != 0;
        three_streams = three_streams && (common & 2);   // (a rank with three streams among ranks without runs the one-communicator path like them)
    }
    g->per_set = g->per_set && three_streams;
    if (g->per_set && (flags & LS_GROUP_FLAG_SIZED_GATHER)) {
        g->sized = true;
        g->gcap = g->capacity;                                   // until a frame has said what it needs
        g->gstep = std::max(1u, (g->capacity + 15u) / 16u);
        for (int i = 0; i < kSets; ++i) {
            if (hipHostMalloc(reinterpret_cast<void **>(&g->h_stat[i]), sizeof(GatherStat), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
                return bail(LS_ERR_HIP);
            std::memset(g->h_stat[i], 0, sizeof(GatherStat));
        }
    }
    if (three_streams && (g->per_set || mode == LS_GROUP_INTERLEAVED)) {   // frames as graph launches (INTERLEAVED: the three launches of a frame)
        g->frame_graph_before = std::max(0l, ls_get_info(tr, LS_INFO_FRAME_GRAPH_STATE)) ? 1 : 0;
        g->frame_graph_set = true;
        if (ls_tracer_set_option(tr, LS_OPT_FRAME_GRAPH, (flags & LS_GROUP_FLAG_NO_GRAPH) ? 0 : 1) != LS_OK) {
            g->err = ls_last_error(tr);
            return bail(LS_ERR_HIP);
        }
    }
    *out = g;
    return LS_OK;
}

namespace {
// The stat of the frame that holds set b, once its rebuild has written it: -> truncated (1 / 0), *need = the largest rank's
// count; negative on error.  wait: spin for it (the frame is at most three frames old), then the whole group is waited for.
int read_stat(ls_group *g, int b, bool wait, uint32_t *need)
{
    GatherStat *st = g->h_stat[b];
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; __atomic_load_n(&st->epoch, __ATOMIC_ACQUIRE) != 1u; ++spins) {
        if (!wait) return -1000;   // (not there yet)
        __builtin_ia32_pause();
        if ((spins & 0xFFFFu) == 0xFFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
            const int rc = ls_group_synchronize(g);   // a slow but healthy frame: after this its word is final
            if (rc != LS_OK) return rc;
            if (__atomic_load_n(&st->epoch, __ATOMIC_ACQUIRE) != 1u) return fail(g, LS_ERR_HIP, "a frame's gather statistics never reached the host");
        }
    }
    if (need) *need = st->max_count;
    return st->truncated ? 1 : 0;
}

// LS_GROUP_FLAG_SIZED_GATHER, before set b's next frame is enqueued: how many records of every slot travel.  The set's
// previous tenant -- three frames ago -- says how many hits the largest shard had (its rebuild left that in pinned memory;
// the gathered headers it comes from are the same on every rank, so every rank arrives at the same number and the
// collectives' sizes agree); a quarter of headroom on top, in steps of capacity / 16, growing at once and shrinking only
// past two steps (every change of size captures the set's graph anew).  A frame whose hits outgrow the headroom within
// three frames is TRUNCATED on every rank alike: ls_group_download_cloud reports it (LS_ERR_OUT_OF_RANGE), the next frames
// are sized up; it is never delivered as complete.
int size_gather(ls_group *g, int b)
{
    if (!g->stat_pending[b]) return LS_OK;   // (no tenant yet: the full capacity travels)
    uint32_t need = 0;
    const int t = read_stat(g, b, true, &need);
    if (t < 0) return t;
    if (t == 1) ++g->truncated_frames;
    g->stat_pending[b] = false;
    __atomic_store_n(&g->h_stat[b]->epoch, 0u, __ATOMIC_RELEASE);   // (the next tenant's rebuild writes 1 again: a constant, so a replayed graph needs no patch)
    const unsigned long long want = (unsigned long long)need + need / 4u + 1024u;
    uint32_t target = (uint32_t)std::min<unsigned long long>(g->capacity, (want + g->gstep - 1u) / g->gstep * g->gstep);
    target = std::max(target, std::min(g->gstep, g->capacity));
    if (target > g->gcap || target + 2u * g->gstep <= g->gcap) g->gcap = target;
    return LS_OK;
}

// Per-set mode, one SHARDED frame: the set is the tracer's next slot, so that the frame's launches, its gather (this set's
// communicator) and the rebuild of the cloud are consecutive work on ONE stream -- no event, no cross-stream wait -- and,
// with LS_OPT_FRAME_GRAPH, one captured graph per set that every later frame of the set replays with a single
// hipGraphLaunch (the poses that changed are patched into the k_project node first).  Three frames in flight: the
// three sets' streams.  Host cost per frame through a one-rank communicator: 39 us (round 3) -> see DESIGN.md section 8.
int trace_per_set(ls_group *g, uint32_t frame_index)
{
    for (int attempt = 0; attempt < 2; ++attempt) {
        const long nb = ls_get_info(g->tr, LS_INFO_NEXT_SLOT);
        if (nb < 0 || nb >= kSets) return fail(g, LS_ERR_HIP, ls_last_error(g->tr));
        const int b = (int)nb;
        if (g->sized && attempt == 0) {
            const int rc_size = size_gather(g, b);
            if (rc_size != LS_OK) return rc_size;
        }
        if (ls_tracer_set_hit_buffers(g->tr, g->slot[b] + LS_GROUP_SLOT_HEADER, reinterpret_cast<uint32_t *>(g->slot[b]), g->capacity) != LS_OK)
            return fail(g, LS_ERR_HIP, ls_last_error(g->tr));
        // (the tag names what the caller adds to the graph: this group, and how many bytes its collective moves)
        if (ls_frame_graph_begin(g->tr, reinterpret_cast<uintptr_t>(g) + (g->sized ? g->gcap : 0u)) != LS_OK) return fail(g, LS_ERR_HIP, ls_last_error(g->tr));
        ls_frame f;
        const int rc = ls_trace_scene_async(g->tr, frame_index, &f);
        if (rc < -1) {
            (void)ls_frame_graph_end(g->tr);
            return fail(g, rc, ls_last_error(g->tr));
        }
        void *stream_v = nullptr;
        uint32_t slot = 0;
        int mode = LS_FRAME_EAGER;
        if (ls_frame_graph_stream(g->tr, &stream_v, &slot, &mode) != LS_OK || (slot != LS_FRAME_NO_SLOT && (int)slot != b)) {
            (void)ls_frame_graph_end(g->tr);
            return fail(g, LS_ERR_HIP, "the tracer's slot rotation and the group's sets disagree");
        }
        hipStream_t s = static_cast<hipStream_t>(stream_v);
        // (a timing / counting frame runs on the handle's own stream, behind everything in flight: its gather and rebuild
        // follow it there, and the host waits for them below, before set b's stream is used again -- a measurement path)
        const bool off_rotation = slot == LS_FRAME_NO_SLOT;
        if (rc == -1) {   // empty scene: nothing was traced (and no graph opened), an empty slot travels
            LSG_HIP(hipMemsetAsync(g->slot[b], 0, 4, s));
            g->loose[b] = s;
        }
        ncclComm_t comm = b == 0 ? g->comm : g->comm_dup[b - 1];
        // (replaying: the gather is a node of the graph already; only this library's launches are described again)
        const size_t gather_bytes = g->sized ? (size_t)ls_group_slot_bytes(g->gcap) : g->slot_bytes;
        if (mode != LS_FRAME_REPLAYING) {
            const ncclResult_t r = rccl().AllGather(g->slot[b], g->gathered[b], gather_bytes, ncclUint8, comm, s);
            if (r != ncclSuccess) {
                (void)ls_frame_graph_end(g->tr);   // (a capture that holds a failed collective is not worth keeping either)
                (void)ls_frame_graph_reset(g->tr);
                return fail(g, LS_ERR_HIP, std::string("ncclAllGather: ") + rccl().GetErrorString(r));
            }
        }
        const int erc = g->sized ? ls_expand_gathered_hits_sized(g->tr, s, g->gathered[b], g->world, g->gcap, g->cloud_points[b], g->cloud_hits[b],
                                                                g->cloud_n[b], g->h_stat[b], 1u)
                                 : ls_expand_gathered_hits_on(g->tr, s, g->gathered[b], g->world, g->capacity, g->cloud_points[b], g->cloud_hits[b],
                                                              g->cloud_n[b]);
        if (erc != LS_OK) {
            (void)ls_frame_graph_end(g->tr);
            return fail(g, LS_ERR_HIP, ls_last_error(g->tr));
        }
        const int e = ls_frame_graph_end(g->tr);
        if (e == 1) continue;   // the graph was given up before anything was launched: once more (captured anew, or plain)
        if (e != LS_OK) return fail(g, e, ls_last_error(g->tr));
        if (off_rotation) LSG_HIP(hipStreamSynchronize(s));
        g->set_frame[b] = frame_index;
        g->set_valid[b] = true;
        g->stat_pending[b] = g->sized;
        return rc == -1 ? -1 : 0;
    }
    return fail(g, LS_ERR_HIP, "the frame graph was discarded twice in a row");
}
}  // namespace

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
        g->set_frame[b] = frame_index;
        g->set_valid[b] = true;
        if (ls_tracer_set_output_buffers(g->tr, g->cloud_points[b], g->cloud_hits[b], g->cloud_n[b], g->cloud_capacity) != LS_OK)
            return fail(g, LS_ERR_HIP, ls_last_error(g->tr));
        const int rc = ls_trace_scene_async(g->tr, frame_index, &f);
        if (rc < -1) return fail(g, rc, ls_last_error(g->tr));
        return rc == -1 ? -1 : 0;
    }
    if (g->per_set) return trace_per_set(g, frame_index);
    g->set_frame[b] = frame_index;
    g->set_valid[b] = true;
    // the set's previous frame (three frames ago) must have left its slot: the gather reads it on the other stream
    if (g->used[b] && ls_tracer_next_frame_waits(g->tr, g->ev_collected[b]) != LS_OK) return fail(g, LS_ERR_HIP, ls_last_error(g->tr));
    if (ls_tracer_set_hit_buffers(g->tr, g->slot[b] + LS_GROUP_SLOT_HEADER, reinterpret_cast<uint32_t *>(g->slot[b]), g->capacity) != LS_OK)
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
    int b = -1;
    for (int i = 0; i < kSets; ++i)
        if (g->set_valid[i] && g->set_frame[i] == frame_index) b = i;
    if (b < 0) return fail(g, LS_ERR_OUT_OF_RANGE, "that frame's buffers have been reused (three frames are kept)");
    std::memset(out, 0, sizeof(*out));
    out->frame = frame_index;
    out->n_rays = g->cloud_capacity;
    out->d_points32 = g->cloud_points[b];
    out->d_hits = g->cloud_hits[b];
    out->d_n_points = g->cloud_n[b];
    return LS_OK;
}

int ls_group_frame_status(ls_group *g, uint32_t frame_index)
{
    if (!g) return LS_ERR_INVALID_ARGUMENT;
    if (!ls_group_owns_frame(g, frame_index)) return fail(g, LS_ERR_OUT_OF_RANGE, "this rank does not hold that frame");
    for (int i = 0; i < kSets; ++i) {
        if (!g->set_valid[i] || g->set_frame[i] != frame_index) continue;
        if (!g->sized || !g->stat_pending[i]) return LS_OK;
        const int t = read_stat(g, i, false, nullptr);
        if (t == -1000) return fail(g, LS_ERR_NOT_COMMITTED, "that frame's rebuild has not finished: wait for it first (ls_group_synchronize)");
        if (t < 0) return t;
        return t == 1 ? fail(g, LS_ERR_OUT_OF_RANGE, "that frame's hits outgrew the sized gather: its cloud is incomplete (the following frames are sized up)") : LS_OK;
    }
    return fail(g, LS_ERR_OUT_OF_RANGE, "that frame's buffers have been reused (three frames are kept)");
}

long ls_group_download_cloud(ls_group *g, uint32_t frame_index, void *points32, void *hits, uint32_t capacity)
{
    ls_frame f;
    int rc = ls_group_cloud(g, frame_index, &f);
    if (rc != LS_OK) return rc;
    if ((rc = ls_group_synchronize(g)) != LS_OK) return rc;
    if ((rc = ls_group_frame_status(g, frame_index)) != LS_OK) return rc;
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
    for (hipStream_t &s : g->loose)
        if (s) { LSG_HIP(hipStreamSynchronize(s)); s = nullptr; }
    return LS_OK;
}

long ls_group_info(ls_group *g, int what)
{
    if (!g) return LS_ERR_INVALID_ARGUMENT;
    Rccl &R = rccl();
    int v = 0;
    switch (what) {
    case LS_GROUP_INFO_RCCL_VERSION: return (R.lib && R.GetVersion && R.GetVersion(&v) == ncclSuccess) ? v : 0;
    case LS_GROUP_INFO_COMM_RANKS: return (g->comm && R.CommCount && R.CommCount(g->comm, &v) == ncclSuccess) ? v : 0;
    case LS_GROUP_INFO_COMM_DEVICE: return (g->comm && R.CommCuDevice && R.CommCuDevice(g->comm, &v) == ncclSuccess) ? v : -1;
    case LS_GROUP_INFO_COMMUNICATORS: return g->comm ? (g->per_set ? kSets : 1) : 0;
    case LS_GROUP_INFO_PER_SET: return g->per_set ? 1 : 0;
    case LS_GROUP_INFO_FRAME_GRAPH: return g->tr ? ls_get_info(g->tr, LS_INFO_FRAME_GRAPH_STATE) : 0;
    case LS_GROUP_INFO_GATHER_CAPACITY: return g->sized ? (long)g->gcap : (long)g->capacity;
    case LS_GROUP_INFO_TRUNCATED_FRAMES: {
        // those counted when their set was reused + the finished ones among the frames still held (not waited for here)
        unsigned long long n = g->truncated_frames;
        if (g->sized)
            for (int i = 0; i < kSets; ++i)
                if (g->set_valid[i] && g->stat_pending[i] && read_stat(g, i, false, nullptr) == 1) ++n;
        return (long)n;
    }
    case LS_GROUP_INFO_ARRANGEMENT_MINE: return (long)g->arrangement_mine;
    case LS_GROUP_INFO_ARRANGEMENT_COMMON: return (long)g->arrangement_common;
    case LS_GROUP_INFO_COLLECTIVES_ARE_A_SHIM: return R.is_shim ? 1 : 0;
    default: return fail(g, LS_ERR_INVALID_ARGUMENT, "unknown info key");
    }
}

const char *ls_group_last_error(const ls_group *g) { return g ? g->err.c_str() : "null group"; }

}  // extern "C"
