// ls_host_pool.h -- worker threads for host-side copies (pageable caller memory <-> pinned staging), the expansion of
// compact point records.  One pool per process, created on first use.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "ls_tuning.h"

namespace lsi {

// Half the cores, at most 8 threads (the calling thread works too).
class HostPool {
public:
    static HostPool &get()
    {
        static HostPool pool;
        return pool;
    }
    int threads() const { return (int)workers_.size() + 1; }
    // run fn(0) .. fn(n-1); the calling thread works too.  `on_done(i)` (optional) is called on the CALLING thread,
    // in index order, as soon as item i is complete -- the caller enqueues item i's DMA there while later items copy.
    void run(size_t n, const std::function<void(size_t)> &fn, const std::function<void(size_t)> *on_done = nullptr)
    {
        if (!n) return;
        if (workers_.empty() || n == 1) {
            for (size_t i = 0; i < n; ++i) { fn(i); if (on_done) (*on_done)(i); }
            return;
        }
        std::unique_lock<std::mutex> run_lock(run_mu_);   // one job at a time
        std::vector<std::atomic<uint8_t>> done(n);
        for (auto &d : done) d.store(0, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            done_ = done.data();
            n_ = n;
            next_.store(0);
            active_ = workers_.size();
            ++generation_;
        }
        cv_.notify_all();
        size_t reported = 0;
        auto report = [&]() {
            while (on_done && reported < n && done[reported].load(std::memory_order_acquire)) (*on_done)(reported++);
        };
        if (!on_done) {
            for (size_t i; (i = next_.fetch_add(1)) < n;) { fn(i); done[i].store(1, std::memory_order_release); }
        } else {
            // the caller only copies when nothing is waiting to be reported (its DMA calls are what the device waits for)
            while (reported < n) {
                report();
                if (reported == n) break;
                if (!done[reported].load(std::memory_order_acquire)) {
                    const size_t i = next_.fetch_add(1);
                    if (i < n) { fn(i); done[i].store(1, std::memory_order_release); }
                    else std::this_thread::yield();
                }
            }
        }
        std::unique_lock<std::mutex> lk(mu_);
        idle_cv_.wait(lk, [&] { return active_ == 0; });
        fn_ = nullptr;
    }

private:
    HostPool()
    {
        int n = tune_int("LS_HOST_THREADS", 0);
        if (n <= 0) {
            const unsigned hw = std::thread::hardware_concurrency();
            n = (int)std::min(8u, std::max(1u, hw / 2u));
        }
        for (int i = 1; i < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            const std::function<void(size_t)> *fn = fn_;
            std::atomic<uint8_t> *done = done_;
            const size_t n = n_;
            lk.unlock();
            for (size_t i; (i = next_.fetch_add(1)) < n;) { (*fn)(i); done[i].store(1, std::memory_order_release); }
            lk.lock();
            if (--active_ == 0) idle_cv_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, idle_cv_;
    const std::function<void(size_t)> *fn_ = nullptr;
    std::atomic<uint8_t> *done_ = nullptr;
    size_t n_ = 0, active_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t generation_ = 0;
    bool stop_ = false;
};

}  // namespace lsi
