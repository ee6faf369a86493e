// san_launch_record.cpp -- the host half of the frame graphs (csrc/ls_launch.h) without a GPU, under AddressSanitizer + UBSan:
// how launch_k packs a kernel's arguments into a LaunchRecord, what same_launch calls "the same launch", and what
// frame_graph_close does with the records (csrc/ls_trace.cpp: describe -> compare -> argument_pointers -> swap with the
// cached record).  VERDICT round 5, item 2: a SIGSEGV under rocprofv3 whose stack ended below frame_graph_close
// (EXPERIMENTS.md E8.2); the questions asked of this code -- is n_args / arg_off right for a kernel with the most arguments
// launch_k admits (19), for arguments aligned above 16, does a blob survive the swap with the cached record and its reuse by
// the next frame -- are answered here, on the CPU, where a wrong answer is an ASan report instead of a fault inside the runtime.
// In kDescribe mode launch_k launches nothing, so plain host functions stand in for the kernels.
#include <cstdio>
#include <cstdlib>
#include <cstring>

// g++ has no hipLaunchKernelGGL (hipcc's); launch_k never reaches it in kDescribe mode -- if it ever did, that is a failure
template <class... A>
inline void hipLaunchKernelGGL(A &&...)
{
    std::puts("FAILED: launch_k tried to launch in kDescribe mode");
    std::abort();
}

#include "../../lidarshooter_amd/csrc/ls_launch.h"

namespace ls {
LaunchSink *&thread_sink()
{
    static thread_local LaunchSink *s = nullptr;
    return s;
}
}  // namespace ls

namespace {

struct alignas(32) Wide {   // aligned above 16: packed at a 16-byte offset, copied whole
    float m[16];
};
struct Odd {                // 13 bytes: the next argument's offset needs padding
    uint8_t b[13];
};
struct Big {                // a geometry table by value, like k_project's (1 KiB)
    uint64_t q[128];
};

void k19(int, Odd, double, Wide, char, uint64_t, float, Big, short, void *, int, int, int, int, int, int, int, int, uint8_t) {}
void k0() {}
void k3(uint32_t, const float *, Wide) {}

int failures = 0;
#define CHECK(c)                                                                  \
    do {                                                                          \
        if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); ++failures; } \
    } while (0)

template <class T>
T read_arg(ls::LaunchRecord &r, uint32_t i)
{
    void *argv[ls::kMaxLaunchArgs];
    ls::argument_pointers(r, argv);
    T v;
    std::memcpy(&v, argv[i], sizeof(T));
    return v;
}

}  // namespace

int main()
{
    ls::LaunchSink sink;
    sink.mode = ls::LaunchSink::kDescribe;
    sink.stream = nullptr;
    ls::thread_sink() = &sink;
    const dim3 grid(7, 2, 1), block(256, 1, 1);
    Wide w;
    for (int i = 0; i < 16; ++i) w.m[i] = 0.5f * (float)i;
    Odd o;
    for (int i = 0; i < 13; ++i) o.b[i] = (uint8_t)(200 + i);
    Big big;
    for (int i = 0; i < 128; ++i) big.q[i] = 0x0101010101010101ull * (uint64_t)i;
    int dummy = 0;

    // ---- 19 arguments: the most launch_k takes (a 20th is a compile error: static_assert against kMaxLaunchArgs)
    sink.n = 0;
    ls::launch_k(k19, grid, block, 128u, nullptr, 1, o, 2.5, w, 'c', 0x1122334455667788ull, 3.5f, big, (short)-7, (void *)&dummy, 10, 11, 12, 13, 14, 15, 16, 17,
                 (uint8_t)255);
    CHECK(sink.n == 1);
    ls::LaunchRecord &r = sink.recs[0];
    CHECK(r.n_args == 19 && r.func == reinterpret_cast<const void *>(k19));
    CHECK(r.arg_off[19] == r.blob.size());
    for (uint32_t i = 0; i < r.n_args; ++i) CHECK(r.arg_off[i] < r.arg_off[i + 1] && r.arg_off[i + 1] <= r.blob.size());
    CHECK(r.arg_off[3] % 16 == 0);                                  // Wide: alignof 32, packed at 16
    CHECK(r.arg_off[2] % alignof(double) == 0 && r.arg_off[5] % 8 == 0 && r.arg_off[7] % 8 == 0);
    CHECK(read_arg<int>(r, 0) == 1 && read_arg<double>(r, 2) == 2.5 && read_arg<char>(r, 4) == 'c');
    CHECK(read_arg<uint64_t>(r, 5) == 0x1122334455667788ull && read_arg<float>(r, 6) == 3.5f && read_arg<short>(r, 8) == -7);
    CHECK(read_arg<void *>(r, 9) == (void *)&dummy && read_arg<int>(r, 17) == 17 && read_arg<uint8_t>(r, 18) == 255);
    CHECK(std::memcmp(read_arg<Odd>(r, 1).b, o.b, 13) == 0);
    {
        const Wide got = read_arg<Wide>(r, 3);
        CHECK(std::memcmp(got.m, w.m, sizeof(w.m)) == 0);
        const Big gb = read_arg<Big>(r, 7);
        CHECK(std::memcmp(gb.q, big.q, sizeof(big.q)) == 0);
    }
    // the last argument's bytes end exactly at the blob's end: a reader of sizeof(argument) bytes never leaves the blob
    CHECK(r.arg_off[18] + sizeof(uint8_t) == r.blob.size());

    // ---- "the same launch": packing the same values twice compares equal (padding is zero-filled), any change does not
    sink.n = 0;
    ls::launch_k(k3, grid, block, 0u, nullptr, 5u, (const float *)w.m, w);
    ls::launch_k(k3, grid, block, 0u, nullptr, 5u, (const float *)w.m, w);
    ls::launch_k(k3, dim3(8, 2, 1), block, 0u, nullptr, 5u, (const float *)w.m, w);   // another grid (the culled launch is sized per frame)
    ls::launch_k(k3, grid, block, 0u, nullptr, 6u, (const float *)w.m, w);            // another value
    ls::launch_k(k0, grid, block, 0u, nullptr);                                       // no arguments at all
    CHECK(sink.n == 5);
    CHECK(ls::same_launch(sink.recs[0], sink.recs[1]));
    CHECK(!ls::same_launch(sink.recs[0], sink.recs[2]) && !ls::same_launch(sink.recs[0], sink.recs[3]));
    CHECK(sink.recs[4].n_args == 0 && sink.recs[4].blob.empty() && sink.recs[4].arg_off[0] == 0);
    CHECK(ls::same_launch(sink.recs[4], sink.recs[4]));

    // ---- what frame_graph_close does, frame after frame: the sink's records are rewritten in place (next() reuses them), a changed
    //      one is swapped with the cached copy, the cached copy's pointers are taken and read
    std::vector<ls::LaunchRecord> cached;
    unsigned patched = 0;
    for (uint32_t frame = 0; frame < 200; ++frame) {
        sink.n = 0;
        Wide pose = w;
        pose.m[3] = (float)(frame / 3);                           // changes every third frame
        ls::launch_k(k3, dim3(7 + frame % 5, 2, 1), block, 0u, nullptr, frame / 7, (const float *)w.m, pose);
        ls::launch_k(k19, grid, block, 128u, nullptr, 1, o, 2.5, w, 'c', (uint64_t)frame, 3.5f, big, (short)-7, (void *)&dummy, 10, 11, 12, 13, 14, 15, 16, 17,
                     (uint8_t)255);
        ls::launch_k(k0, grid, block, 0u, nullptr);
        if (frame == 0) {
            cached.assign(sink.recs.begin(), sink.recs.begin() + (ptrdiff_t)sink.n);
            continue;
        }
        CHECK(sink.n == cached.size());
        for (size_t i = 0; i < sink.n; ++i) {
            ls::LaunchRecord &now = sink.recs[i];
            CHECK(now.func == cached[i].func);
            if (ls::same_launch(now, cached[i])) continue;
            void *argv[ls::kMaxLaunchArgs];
            ls::argument_pointers(now, argv);
            for (uint32_t a = 0; a < now.n_args; ++a) {           // every pointer + its argument's size stays inside the blob
                const uint8_t *p = static_cast<const uint8_t *>(argv[a]);
                CHECK(p >= now.blob.data() && p + (now.arg_off[a + 1] - now.arg_off[a]) <= now.blob.data() + now.blob.size());
            }
            std::swap(cached[i], now);
            ++patched;
        }
        CHECK(read_arg<uint64_t>(cached[1], 5) == frame);         // the cached copy is this frame's
        CHECK(read_arg<Wide>(cached[0], 2).m[3] == (float)(frame / 3));
    }
    CHECK(patched > 200);
    ls::thread_sink() = nullptr;
    std::printf("launch records: %u patches over 200 frames, failures: %d\n", patched, failures);
    return failures ? 1 : 0;
}
