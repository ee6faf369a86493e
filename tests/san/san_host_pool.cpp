// san_host_pool.cpp -- lidarshooter_amd/csrc/ls_host_pool.h under ThreadSanitizer / AddressSanitizer (CPU build only; never on
// the GPU box).  The pool is hand-rolled (a generation counter, an entered / left hand-shake, the caller's stack shared with
// the workers) and serves two tracers at once in a two-sensor process (BASELINE configs[2]; the reference calls its tracer
// from the Qt thread and the ROS spinner thread, mainwindow.cpp:150-154,315-323,335-339): two callers hammer run() with and
// without on_done, jobs of every size from 0 to a few hundred items, pauses long enough for the workers to fall asleep.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <thread>
#include <vector>

#include "../../lidarshooter_amd/csrc/ls_host_pool.h"

static std::atomic<long> g_failures{0};

static void caller(int id, int jobs)
{
    lsi::HostPool &pool = lsi::HostPool::get();
    unsigned seed = 1234u + 77u * (unsigned)id;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    for (int j = 0; j < jobs; ++j) {
        const size_t n = rnd() % 7 == 0 ? rnd() % 3 : rnd() % 300;
        std::vector<unsigned> out(n, 0u);
        std::vector<unsigned> in(n);
        for (size_t i = 0; i < n; ++i) in[i] = rnd();
        const std::function<void(size_t)> fn = [&](size_t i) {
            unsigned acc = in[i];
            for (int k = 0; k < 50 + (int)(in[i] % 200); ++k) acc = acc * 2654435761u + (unsigned)k;
            out[i] = acc;
        };
        if (j % 2 == 0) {
            pool.run(n, fn);
        } else {
            size_t expect = 0;
            bool ordered = true;
            const std::function<void(size_t)> done = [&](size_t i) {
                ordered = ordered && i == expect && out[i] != 0u;   // in index order, on the calling thread, after the item is complete
                ++expect;
            };
            pool.run(n, fn, &done);
            if (!ordered || expect != n) g_failures.fetch_add(1);
        }
        for (size_t i = 0; i < n; ++i) {
            unsigned acc = in[i];
            for (int k = 0; k < 50 + (int)(in[i] % 200); ++k) acc = acc * 2654435761u + (unsigned)k;
            if (out[i] != acc) g_failures.fetch_add(1);
        }
        if (j % 97 == 96) std::this_thread::sleep_for(std::chrono::microseconds(600));   // past kSpinUs: the workers go to sleep
    }
}

int main(int argc, char **argv)
{
    const int jobs = argc > 1 ? std::atoi(argv[1]) : 3000;
    std::printf("host pool: %d threads, two callers x %d jobs\n", lsi::HostPool::get().threads(), jobs);
    std::thread a(caller, 0, jobs), b(caller, 1, jobs);
    a.join();
    b.join();
    std::printf("failures: %ld\n", g_failures.load());
    return g_failures.load() ? 1 : 0;
}
