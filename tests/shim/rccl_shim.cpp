// rccl_shim.cpp -- TEST INFRASTRUCTURE, not part of the product: the handful of RCCL entry points that
// lidarshooter_amd/csrc/ls_group.cpp resolves with dlsym, for several PROCESSES ON ONE DEVICE.
//
// Why: RCCL refuses two ranks on one GPU and the build pool grants one GPU, so the world > 1 control flow of
// include/lidarshooter_group.h (per-set communicators, the agreement collective, the sized gather, one all-gather per
// frame) had never met a peer.  With this library in place of librccl.so.1 (LS_GROUP_RCCL_LIBRARY=<path>, read by
// ls_group.cpp's loader) two child processes run that code against each other on the one GPU; tests/test_gpu_group_shim.py
// compares both ranks' clouds with the CPU oracle.  It proves SEMANTICS, not speed, and says nothing about xGMI.
//
// How: a communicator is a POSIX shared-memory segment named by the unique id -- a header of sequence counters and one
// staging region per rank.  A collective is HOST-BLOCKING: wait for the stream, copy the send buffer device -> region,
// publish, wait for the peers, copy every region -> receive buffer, publish.  Stream order is kept (the copies are issued on
// the caller's stream and waited for), the host is not asynchronous the way RCCL's is; every rank must issue the
// collectives of one communicator in the same order (RCCL's own rule).  A collective on a CAPTURING stream is refused
// (ncclInvalidUsage): frame graphs that hold a collective are not what this shim can test.  Every wait has a deadline
// (LS_SHIM_TIMEOUT_S, default 60 s) and fails with ncclSystemError instead of hanging a GPU box.
//
// Fault injection (the tests' "one rank only" cases):
//   LS_SHIM_FAIL_SPLIT_RANK=<r>   ncclCommSplit on rank r takes part in the collective (its peers succeed) and then fails locally
//   LS_SHIM_NO_SPLIT=1            ncclCommSplit fails on every rank at once (the group's unique-id + broadcast path is taken)
//   LS_SHIM_FAIL_INIT_RANK=<r>    every ncclCommInitRank AFTER THE FIRST of the process fails on rank r, after the rendezvous
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kRegionBytes = 8u << 20;   // staging per rank; larger collectives go in pieces
constexpr char kMagic[8] = {'L', 'S', 'S', 'H', 'I', 'M', '1', 0};

struct alignas(64) Line {
    std::atomic<uint64_t> v;
};

struct Header {
    std::atomic<uint32_t> arrived;
    uint32_t pad[15];
    Line in[kMaxRanks];        // ops whose data stand in this rank's region
    Line out[kMaxRanks];       // ops this rank has finished reading
    Line split_seq[kMaxRanks]; // ncclCommSplit: this rank has posted its (colour, key) for split number v
    Line split_done[kMaxRanks];
    std::atomic<int32_t> split_color[kMaxRanks], split_key[kMaxRanks];
};

double timeout_s()
{
    const char *e = std::getenv("LS_SHIM_TIMEOUT_S");
    const double v = e ? std::atof(e) : 0.0;
    return v > 0.0 ? v : 60.0;
}

}  // namespace

struct ncclComm {
    Header *h = nullptr;
    uint8_t *regions = nullptr;
    size_t map_bytes = 0;
    int rank = 0, nranks = 1, device = 0;
    uint64_t op = 0, splits = 0;
    std::string name;
    bool dead = false;
};

namespace {

template <class F>
bool wait_for(ncclComm *c, F &&ready)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; !ready(); ++spins) {
        if (spins < 2000) { __builtin_ia32_pause(); continue; }
        sched_yield();
        if ((spins & 0x3FFu) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) {
            std::fprintf(stderr, "rccl_shim: rank %d of %s gave up waiting for a peer after %.0f s\n", c->rank, c->name.c_str(), timeout_s());
            c->dead = true;
            return false;
        }
    }
    return true;
}

bool all_at_least(ncclComm *c, Line *lines, uint64_t k)
{
    for (int r = 0; r < c->nranks; ++r)
        if (lines[r].v.load(std::memory_order_acquire) < k) return false;
    return true;
}

ncclResult_t attach(ncclComm_t *out, const std::string &name, int nranks, int rank)
{
    *out = nullptr;
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    const int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    const size_t bytes = ((sizeof(Header) + 4095) & ~size_t(4095)) + kRegionBytes * (size_t)nranks;
    if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); return ncclSystemError; }   // (the same size from every rank; new pages are zero)
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    ncclComm *c = new ncclComm();
    c->h = static_cast<Header *>(p);
    c->regions = static_cast<uint8_t *>(p) + ((sizeof(Header) + 4095) & ~size_t(4095));
    c->map_bytes = bytes;
    c->rank = rank;
    c->nranks = nranks;
    c->name = name;
    if (hipGetDevice(&c->device) != hipSuccess) c->device = -1;
    c->h->arrived.fetch_add(1u, std::memory_order_acq_rel);
    const bool met = wait_for(c, [&] { return c->h->arrived.load(std::memory_order_acquire) >= (uint32_t)nranks; });
    if (rank == 0) shm_unlink(name.c_str());   // everybody holds a mapping (or never will): the name can go
    if (!met) {
        munmap(p, bytes);
        delete c;
        return ncclSystemError;
    }
    *out = c;
    return ncclSuccess;
}

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

bool capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

// One piece of a collective: ranks in `senders` put n bytes of `send` into their region, every rank then reads the regions
// of `senders` into recv + r * stride (stride 0: one sender, plain recv).
ncclResult_t piece(ncclComm *c, hipStream_t s, const uint8_t *send, uint8_t *recv, size_t n, size_t stride, int root /* -1: everybody sends */)
{
    const uint64_t k = ++c->op;
    if (!wait_for(c, [&] { return all_at_least(c, c->h->out, k - 1); })) return ncclSystemError;   // my region's last tenant has been read
    if (root < 0 || root == c->rank) {
        if (hipMemcpyAsync(c->regions + kRegionBytes * (size_t)c->rank, send, n, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    c->h->in[c->rank].v.store(k, std::memory_order_release);
    if (!wait_for(c, [&] { return all_at_least(c, c->h->in, k); })) return ncclSystemError;
    for (int r = 0; r < c->nranks; ++r) {
        if (root >= 0 && r != root) continue;
        uint8_t *dst = recv + (root < 0 ? stride * (size_t)r : 0);
        if (hipMemcpyAsync(dst, c->regions + kRegionBytes * (size_t)r, n, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    c->h->out[c->rank].v.store(k, std::memory_order_release);
    return ncclSuccess;
}

}  // namespace

extern "C" {

// tests look for this symbol to make sure the group really ran on the shim
int ls_rccl_shim_marker(void) { return 1; }

ncclResult_t ncclGetVersion(int *version)
{
    if (!version) return ncclInvalidArgument;
    *version = 1;   // no RCCL release has this number: bench.py / the tests label a run with it "shim"
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled HIP error (shim)";
    case ncclSystemError: return "system error (shim: shared memory, or a peer that never came)";
    case ncclInternalError: return "internal error (shim: injected)";
    case ncclInvalidArgument: return "invalid argument (shim)";
    case ncclInvalidUsage: return "invalid usage (shim: a collective on a capturing stream, or after a failure)";
    default: return "error (shim)";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    static std::atomic<uint32_t> counter{0};
    std::memset(id, 0, sizeof(*id));
    std::memcpy(id->internal, kMagic, 8);
    const auto ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
    std::snprintf(id->internal + 8, 100, "/lsshim-%d-%u-%llx", (int)getpid(), counter.fetch_add(1u), (unsigned long long)ns);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm) return ncclInvalidArgument;
    *comm = nullptr;
    if (std::memcmp(id.internal, kMagic, 8) != 0) return ncclInvalidArgument;   // (an all-zero id from a rank 0 that could not make one)
    id.internal[sizeof(id.internal) - 1] = 0;
    static std::atomic<int> inits{0};
    const ncclResult_t rc = attach(comm, std::string(id.internal + 8), nranks, rank);
    const char *fail = std::getenv("LS_SHIM_FAIL_INIT_RANK");
    if (rc == ncclSuccess && inits.fetch_add(1) > 0 && fail && std::atoi(fail) == rank) {   // the peers hold a communicator this rank walked away from
        (void)ncclCommDestroy(*comm);
        *comm = nullptr;
        return ncclInternalError;
    }
    return rc;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclInvalidArgument;
    munmap(c->h, c->map_bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int *n)
{
    if (!c || !n) return ncclInvalidArgument;
    *n = c->nranks;
    return ncclSuccess;
}

ncclResult_t ncclCommCuDevice(const ncclComm_t c, int *d)
{
    if (!c || !d) return ncclInvalidArgument;
    *d = c->device;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t c, int *r)
{
    if (!c || !r) return ncclInvalidArgument;
    *r = c->rank;
    return ncclSuccess;
}

ncclResult_t ncclCommSplit(ncclComm_t c, int color, int key, ncclComm_t *newcomm, ncclConfig_t *)
{
    if (!c || !newcomm) return ncclInvalidArgument;
    *newcomm = nullptr;
    if (c->dead) return ncclInvalidUsage;
    if (std::getenv("LS_SHIM_NO_SPLIT")) return ncclInvalidUsage;   // every rank alike, nothing exchanged
    const uint64_t s = ++c->splits;
    Header *h = c->h;
    h->split_color[c->rank].store(color, std::memory_order_relaxed);
    h->split_key[c->rank].store(key, std::memory_order_relaxed);
    h->split_seq[c->rank].v.store(s, std::memory_order_release);
    if (!wait_for(c, [&] { return all_at_least(c, h->split_seq, s); })) return ncclSystemError;
    std::vector<std::pair<int, int>> members;   // (key, old rank) of my colour
    for (int r = 0; r < c->nranks; ++r)
        if (h->split_color[r].load(std::memory_order_relaxed) == color) members.push_back({h->split_key[r].load(std::memory_order_relaxed), r});
    h->split_done[c->rank].v.store(s, std::memory_order_release);   // (the slots may be written again once everybody has read them)
    if (!wait_for(c, [&] { return all_at_least(c, h->split_done, s); })) return ncclSystemError;
    if (color == NCCL_SPLIT_NOCOLOR) return ncclSuccess;
    std::sort(members.begin(), members.end());
    int new_rank = 0;
    for (size_t i = 0; i < members.size(); ++i)
        if (members[i].second == c->rank) new_rank = (int)i;
    ncclComm_t fresh = nullptr;
    const ncclResult_t rc = attach(&fresh, c->name + "-s" + std::to_string(s) + "c" + std::to_string(color), (int)members.size(), new_rank);
    if (rc != ncclSuccess) return rc;
    const char *fail = std::getenv("LS_SHIM_FAIL_SPLIT_RANK");
    if (fail && std::atoi(fail) == c->rank) {   // the peers hold a communicator this rank walked away from
        (void)ncclCommDestroy(fresh);
        return ncclInternalError;
    }
    *newcomm = fresh;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s)
{
    const size_t bytes = count * type_bytes(t);
    if (!c || !send || !recv || !type_bytes(t)) return ncclInvalidArgument;
    if (c->dead || capturing(s)) return ncclInvalidUsage;
    if (!bytes) return ncclSuccess;
    for (size_t off = 0; off < bytes; off += kRegionBytes) {
        const ncclResult_t rc = piece(c, s, static_cast<const uint8_t *>(send) + off, static_cast<uint8_t *>(recv) + off, std::min(kRegionBytes, bytes - off), bytes, -1);
        if (rc != ncclSuccess) { c->dead = true; return rc; }
    }
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t s)
{
    const size_t bytes = count * type_bytes(t);
    if (!c || !recv || !type_bytes(t) || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    if (c->dead || capturing(s)) return ncclInvalidUsage;
    if (!bytes) return ncclSuccess;
    for (size_t off = 0; off < bytes; off += kRegionBytes) {
        const ncclResult_t rc = piece(c, s, static_cast<const uint8_t *>(send) + off, static_cast<uint8_t *>(recv) + off, std::min(kRegionBytes, bytes - off), 0, root);
        if (rc != ncclSuccess) { c->dead = true; return rc; }
    }
    return ncclSuccess;
}

}  // extern "C"
