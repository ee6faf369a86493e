// graph_launch.hip -- what does the HOST pay per frame for the launch pattern of a sharded group frame
// (csrc/ls_group.cpp:ls_group_trace), and what would one captured HIP graph per buffer set cost instead?
//   A  three kernel launches on one of three rotating streams (a tracer frame: k_project with its 2.4 KB of
//      kernel arguments, k_project_finish, k_pack)
//   B  A + the collective side as round 3 issued it: event record + wait, a device-to-device copy standing in for the
//      all-gather (or RCCL's own ncclAllGather over a one-rank communicator with --rccl), one more kernel, event record,
//      and the slot guard (one wait on the next frame's stream)
//   C  one hipGraphLaunch of [3 kernels] per frame
//   D  one hipGraphLaunch of [3 kernels + copy / all-gather + kernel] per frame, each set on its own stream
//   E  D + hipGraphExecKernelNodeSetParams of the first kernel before every launch (a pose change)
//   F  (round 6) ONE hipGraphLaunch for THREE frames -- captured from s[0], forked to the two other
//      set streams by events and joined again: three parallel chains of [3 kernels] -- and G the same with two frames per
//      chain (six frames per launch), H = G + SetParams of every frame's first kernel: can a launch per K frames get the host
//      under 3 us per frame?
// Kernels are one workgroup each: the loop is host-bound, microseconds per frame = host cost per frame.
// build: hipcc --offload-arch=gfx950 -O3 -o graph_launch graph_launch.hip -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

struct BigArgs { float f[600]; };   // ~ GeomBatch: 2.4 KB by value

__global__ void k_a(BigArgs a, float *out) { if (threadIdx.x == 0) out[blockIdx.x] = a.f[blockIdx.x % 600] + a.f[7]; }
__global__ void k_b(float *out, const float *in) { out[threadIdx.x] = in[threadIdx.x] + 1.0f; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const bool use_rccl = argc > 1 && !strcmp(argv[1], "--rccl");
    constexpr int kSets = 3, kFrames = 20000;
    hipStream_t s[kSets], comm;
    hipEvent_t ev_frame[kSets], ev_coll[kSets];
    for (int i = 0; i < kSets; ++i) {
        CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
        CK(hipEventCreateWithFlags(&ev_frame[i], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ev_coll[i], hipEventDisableTiming));
    }
    CK(hipStreamCreateWithFlags(&comm, hipStreamNonBlocking));
    float *buf[kSets], *slot[kSets], *gath[kSets];
    const size_t slot_bytes = 64 + 16 * 65536;
    for (int i = 0; i < kSets; ++i) {
        CK(hipMalloc(&buf[i], 4096 * 4));
        CK(hipMalloc(&slot[i], slot_bytes));
        CK(hipMalloc(&gath[i], slot_bytes));
    }
    BigArgs big;
    for (int i = 0; i < 600; ++i) big.f[i] = (float)i;

    ncclComm_t nc = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    if (use_rccl) {
        void *lib = dlopen("librccl.so.1", RTLD_NOW);
        if (!lib) { printf("no librccl\n"); return 1; }
        auto GetId = reinterpret_cast<ncclResult_t (*)(ncclUniqueId *)>(dlsym(lib, "ncclGetUniqueId"));
        auto Init = reinterpret_cast<ncclResult_t (*)(ncclComm_t *, int, ncclUniqueId, int)>(dlsym(lib, "ncclCommInitRank"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        ncclUniqueId id;
        if (GetId(&id) != ncclSuccess || Init(&nc, 1, id, 0) != ncclSuccess) { printf("rccl init failed\n"); return 1; }
    }
    auto gather = [&](int b, hipStream_t st) -> int {
        if (use_rccl) return AllGather(slot[b], gath[b], slot_bytes, ncclUint8, nc, st) == ncclSuccess ? 0 : 1;
        return hipMemcpyAsync(gath[b], slot[b], slot_bytes, hipMemcpyDeviceToDevice, st) == hipSuccess ? 0 : 1;
    };
    auto tracer = [&](int b, hipStream_t st) {
        hipLaunchKernelGGL(k_a, dim3(8), dim3(64), 0, st, big, buf[b]);
        hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, st, buf[b] + 64, buf[b]);
        hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, st, slot[b], buf[b] + 64);
    };
    auto report = [&](const char *what, double t0, double t_enq, double t_end) {
        printf("%-78s enqueue %6.2f us/frame, with drain %6.2f us/frame\n", what, (t_enq - t0) / kFrames, (t_end - t0) / kFrames);
    };
    for (int rep = 0; rep < 2; ++rep) {
        // ---- A
        double t0 = now_us();
        for (int f = 0; f < kFrames; ++f) tracer(f % kSets, s[f % kSets]);
        double t1 = now_us();
        CK(hipDeviceSynchronize());
        if (rep) report("A  3 launches per frame, three rotating streams", t0, t1, now_us());
        // ---- B
        bool used[kSets] = {};
        t0 = now_us();
        for (int f = 0; f < kFrames; ++f) {
            const int b = f % kSets;
            if (used[b]) CK(hipStreamWaitEvent(s[b], ev_coll[b], 0));
            tracer(b, s[b]);
            CK(hipEventRecord(ev_frame[b], s[b]));
            CK(hipStreamWaitEvent(comm, ev_frame[b], 0));
            if (gather(b, comm)) return 1;
            hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, comm, buf[b] + 128, gath[b]);
            CK(hipEventRecord(ev_coll[b], comm));
            used[b] = true;
        }
        t1 = now_us();
        CK(hipDeviceSynchronize());
        if (rep) report(use_rccl ? "B  round 3's group frame (ncclAllGather, one rank): 9 calls" : "B  round 3's group frame (copy for the gather): 9 calls", t0, t1, now_us());
    }
    // ---- graphs
    hipGraph_t g3[kSets], g5[kSets];
    hipGraphExec_t x3[kSets], x5[kSets];
    for (int b = 0; b < kSets; ++b) {
        CK(hipStreamBeginCapture(s[b], hipStreamCaptureModeThreadLocal));
        tracer(b, s[b]);
        CK(hipStreamEndCapture(s[b], &g3[b]));
        CK(hipGraphInstantiate(&x3[b], g3[b], nullptr, nullptr, 0));
        CK(hipStreamBeginCapture(s[b], hipStreamCaptureModeThreadLocal));
        tracer(b, s[b]);
        if (gather(b, s[b])) { printf("the gather cannot be captured\n"); return 1; }
        hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s[b], buf[b] + 128, gath[b]);
        CK(hipStreamEndCapture(s[b], &g5[b]));
        CK(hipGraphInstantiate(&x5[b], g5[b], nullptr, nullptr, 0));
    }
    // the first kernel node of g5[b], for E
    hipGraphNode_t first_node[kSets];
    hipKernelNodeParams kp[kSets];
    for (int b = 0; b < kSets; ++b) {
        size_t n = 0;
        CK(hipGraphGetNodes(g5[b], nullptr, &n));
        std::vector<hipGraphNode_t> nodes(n);
        CK(hipGraphGetNodes(g5[b], nodes.data(), &n));
        first_node[b] = nullptr;
        for (auto nd : nodes) {
            hipGraphNodeType ty;
            CK(hipGraphNodeGetType(nd, &ty));
            if (ty != hipGraphNodeTypeKernel) continue;
            hipKernelNodeParams p;
            CK(hipGraphKernelNodeGetParams(nd, &p));
            if (p.func == reinterpret_cast<void *>(k_a)) { first_node[b] = nd; kp[b] = p; }
        }
        if (!first_node[b]) { printf("k_a node not found among %zu nodes\n", n); return 1; }
        if (b == 0) printf("graph D has %zu nodes\n", n);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now_us();
        for (int f = 0; f < kFrames; ++f) CK(hipGraphLaunch(x3[f % kSets], s[f % kSets]));
        double t1 = now_us();
        CK(hipDeviceSynchronize());
        if (rep) report("C  one graph of [3 kernels] per frame", t0, t1, now_us());
        t0 = now_us();
        for (int f = 0; f < kFrames; ++f) CK(hipGraphLaunch(x5[f % kSets], s[f % kSets]));
        t1 = now_us();
        CK(hipDeviceSynchronize());
        if (rep) report(use_rccl ? "D  one graph of [3 kernels + ncclAllGather + kernel] per frame" : "D  one graph of [3 kernels + copy + kernel] per frame", t0, t1, now_us());
        t0 = now_us();
        float *outp;
        for (int f = 0; f < kFrames; ++f) {
            const int b = f % kSets;
            big.f[7] = (float)f;
            outp = buf[b];
            void *args[2] = {&big, &outp};
            hipKernelNodeParams p = kp[b];
            p.kernelParams = args;
            p.extra = nullptr;
            CK(hipGraphExecKernelNodeSetParams(x5[b], first_node[b], &p));
            CK(hipGraphLaunch(x5[b], s[b]));
        }
        t1 = now_us();
        CK(hipDeviceSynchronize());
        if (rep) report("E  D + new arguments for the first kernel every frame (SetParams)", t0, t1, now_us());
    }
    // ---- F / G / H: several frames per graph launch
    for (int per_chain = 1; per_chain <= 2; ++per_chain) {
        hipEvent_t fork, join[kSets];
        CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        for (int b = 0; b < kSets; ++b) CK(hipEventCreateWithFlags(&join[b], hipEventDisableTiming));
        hipGraph_t gm;
        hipGraphExec_t xm;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        CK(hipEventRecord(fork, s[0]));
        for (int b = 1; b < kSets; ++b) CK(hipStreamWaitEvent(s[b], fork, 0));
        for (int b = 0; b < kSets; ++b)
            for (int k = 0; k < per_chain; ++k) tracer(b, s[b]);
        for (int b = 1; b < kSets; ++b) {
            CK(hipEventRecord(join[b], s[b]));
            CK(hipStreamWaitEvent(s[0], join[b], 0));
        }
        CK(hipStreamEndCapture(s[0], &gm));
        CK(hipGraphInstantiate(&xm, gm, nullptr, nullptr, 0));
        size_t n = 0;
        CK(hipGraphGetNodes(gm, nullptr, &n));
        std::vector<hipGraphNode_t> nodes(n), firsts;
        CK(hipGraphGetNodes(gm, nodes.data(), &n));
        hipKernelNodeParams kpm;
        for (auto nd : nodes) {
            hipGraphNodeType ty;
            CK(hipGraphNodeGetType(nd, &ty));
            if (ty != hipGraphNodeTypeKernel) continue;
            hipKernelNodeParams p;
            CK(hipGraphKernelNodeGetParams(nd, &p));
            if (p.func == reinterpret_cast<void *>(k_a)) { firsts.push_back(nd); kpm = p; }
        }
        const int frames_per_launch = kSets * per_chain;
        printf("graph of %d frames has %zu nodes (%zu first kernels)\n", frames_per_launch, n, firsts.size());
        for (int patch = 0; patch <= 1; ++patch)
            for (int rep = 0; rep < 2; ++rep) {
                double t0 = now_us();
                for (int f = 0; f < kFrames; f += frames_per_launch) {
                    if (patch)
                        for (size_t k = 0; k < firsts.size(); ++k) {
                            big.f[7] = (float)(f + (int)k);
                            float *outp = buf[k % kSets];
                            void *args[2] = {&big, &outp};
                            hipKernelNodeParams p = kpm;
                            p.kernelParams = args;
                            p.extra = nullptr;
                            CK(hipGraphExecKernelNodeSetParams(xm, firsts[k], &p));
                        }
                    CK(hipGraphLaunch(xm, s[0]));
                }
                double t1 = now_us();
                CK(hipDeviceSynchronize());
                char what[120];
                snprintf(what, sizeof(what), "%s  one graph of %d frames ([3 kernels] x %d chains x %d)%s", patch ? "H" : (per_chain == 1 ? "F" : "G"), frames_per_launch, kSets, per_chain,
                         patch ? " + SetParams per frame" : "");
                if (rep) report(what, t0, t1, now_us());
            }
    }
    // ---- I: K frames in a ROW per graph (no fork: a chain of 3 K kernels on the set's stream), the three sets' graphs rotating
    for (int K : {2, 4, 8}) {
        hipGraph_t gl[kSets];
        hipGraphExec_t xl[kSets];
        for (int b = 0; b < kSets; ++b) {
            CK(hipStreamBeginCapture(s[b], hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < K; ++k) tracer(b, s[b]);
            CK(hipStreamEndCapture(s[b], &gl[b]));
            CK(hipGraphInstantiate(&xl[b], gl[b], nullptr, nullptr, 0));
        }
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now_us();
            for (int f = 0, l = 0; f < kFrames; f += K, ++l) CK(hipGraphLaunch(xl[l % kSets], s[l % kSets]));
            double t1 = now_us();
            CK(hipDeviceSynchronize());
            char what[120];
            snprintf(what, sizeof(what), "I  one graph of %d frames in a row ([3 kernels] x %d on one stream)", K, K);
            if (rep) report(what, t0, t1, now_us());
        }
    }
    // did E's arguments arrive?  k_a writes a.f[blockIdx % 600] + a.f[7]: block 0 of the last frame of set b
    for (int b = 0; b < kSets; ++b) {
        float v = 0.f;
        CK(hipMemcpy(&v, buf[b], 4, hipMemcpyDeviceToHost));
        printf("set %d: k_a saw f[7] = %.0f (last frame of the set: %d)\n", b, v, ((kFrames - 1) / kSets) * kSets + b - (b > (kFrames - 1) % kSets ? kSets : 0));
    }
    return 0;
}
