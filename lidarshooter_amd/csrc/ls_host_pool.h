// ls_host_pool.h -- worker threads for host-side copies (pageable caller memory <-> pinned staging), the expansion of
// compact point records.  One pool per process, created on first use.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "ls_tuning.h"

namespace lsi {

// Half the cores, at most 8 threads (the calling thread works too).  A worker that has finished a job keeps polling for
// the next one for kSpinUs before it goes to sleep on the condition variable, and the caller polls for the job's end: a
// frame hands the pool two or three jobs ~0.1 ms apart (staging copy, expansion of the cloud), and a futex wake-up on each
// side of each of them was 40 of the 62 us an expansion of 256 k points took (MI355X host, 8 threads); between frames of a
// 10 Hz sensor the workers sleep.
class HostPool {
public:
    static constexpr int kSpinUs = 250;
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
            entered_ = 0;
            left_.store(0, std::memory_order_relaxed);
            open_ = true;
            generation_.fetch_add(1, std::memory_order_release);
        }
        if (sleepers_.load(std::memory_order_acquire)) cv_.notify_all();   // (a worker checks the generation under mu_ before it sleeps)
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
        // the job closes: a worker that wakes up from now on stays out; those that entered must have left (fn, done and n
        // are the caller's stack) -- polled: their last items end within microseconds of the caller's own -- with a yield
        // now and then for the case of fewer cores than threads
        size_t entered;
        {
            std::lock_guard<std::mutex> lk(mu_);
            open_ = false;
            fn_ = nullptr;
            entered = entered_;
        }
        for (uint32_t spins = 0; left_.load(std::memory_order_acquire) != entered; ++spins) {
            if ((spins & 1023u) == 1023u) std::this_thread::yield();
            else cpu_relax();
        }
    }

private:
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
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
            stop_.store(true, std::memory_order_release);
        }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            // poll for the next job for kSpinUs, then sleep
            bool ready = false;
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t spins = 0;; ++spins) {
                if (stop_.load(std::memory_order_acquire) || generation_.load(std::memory_order_acquire) != seen) { ready = true; break; }
                if ((spins & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(kSpinUs)) break;
                cpu_relax();
            }
            std::unique_lock<std::mutex> lk(mu_);
            if (!ready) {
                sleepers_.fetch_add(1, std::memory_order_acq_rel);
                cv_.wait(lk, [&] { return stop_.load(std::memory_order_acquire) || generation_.load(std::memory_order_acquire) != seen; });
                sleepers_.fetch_sub(1, std::memory_order_acq_rel);
            }
            if (stop_.load(std::memory_order_acquire)) return;
            seen = generation_.load(std::memory_order_acquire);
            if (!open_) continue;   // (woke up after the caller had finished the job alone)
            ++entered_;
            const std::function<void(size_t)> *fn = fn_;
            std::atomic<uint8_t> *done = done_;
            const size_t n = n_;
            lk.unlock();
            for (size_t i; (i = next_.fetch_add(1)) < n;) { (*fn)(i); done[i].store(1, std::memory_order_release); }
            left_.fetch_add(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_;
    const std::function<void(size_t)> *fn_ = nullptr;
    std::atomic<uint8_t> *done_ = nullptr;
    size_t n_ = 0, entered_ = 0;   // (under mu_)
    bool open_ = false;            // (under mu_)
    std::atomic<size_t> next_{0}, left_{0};
    std::atomic<uint64_t> generation_{0};
    std::atomic<int> sleepers_{0};
    std::atomic<bool> stop_{false};
};

}  // namespace lsi
