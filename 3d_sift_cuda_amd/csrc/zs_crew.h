/*
 * zs_crew.h -- the host threads of the one-process Z-slab driver (zslab_driver.hip): one per rank.  Plain C++11, no HIP in here
 * (tests/crew_check.cpp runs it under ThreadSanitizer on the CPU).
 */
#ifndef SIFT3D_ZS_CREW_H
#define SIFT3D_ZS_CREW_H
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

/* One host thread per rank (round 5).  Rounds 2 - 4 queued the launches of every rank from the calling thread, round-robin: 0.5 ms of
 * HIP calls per rank and extraction, i.e. 4 ms for eight ranks whose devices each have about 2 ms of work -- the enqueueing thread
 * was what an eight-GPU extraction would have waited for (profiles/r05_zslab_host_cost.txt).  The crew keeps the driver's shape --
 * a step of the schedule is "for every rank: ..." -- and runs the ranks of a step side by side: rank 0 on the calling thread, rank r
 * on worker r, the calling thread going on when all are back (what a step reads of a neighbour -- its event, its buffers -- was
 * written in an earlier step).  A worker waits for its next step spinning for some tens of microseconds, then asleep: between the
 * steps of an extraction it stays hot, between extractions it costs nothing. */
struct zs_crew {
    std::vector<std::thread> th; /* workers of ranks 1 .. S-1 */
    std::atomic<uint64_t> gen{0};
    std::atomic<int> left{0};
    std::atomic<int> sleepers{0};
    std::atomic<bool> quit{false};
    int n_active = 0; /* ranks that take part in the current step; written, like the two below, before gen moves on */
    void (*tramp)(void *, int) = nullptr;
    void *ctx = nullptr;
    std::mutex m;
    std::condition_variable cv;
    int spin_limit = 4096; /* pauses before a worker goes to sleep (some tens of microseconds) */

    void worker(int r, std::function<void(int)> on_start)
    {
        if (on_start) on_start(r); /* the driver: hipSetDevice of the rank's device, once */
        uint64_t seen = 0;
        for (;;) {
            int spins = 0;
            while (gen.load() == seen && !quit.load()) {
                if (++spins < spin_limit) {
                    __builtin_ia32_pause();
                    continue;
                }
                std::unique_lock<std::mutex> lk(m);
                sleepers.fetch_add(1);
                cv.wait(lk, [&] { return gen.load() != seen || quit.load(); });
                sleepers.fetch_sub(1);
            }
            if (quit.load()) return;
            seen++; /* every worker answers every step, taking part or not: the calling thread publishes the next one only then */
            if (r < n_active) tramp(ctx, r);
            left.fetch_sub(1);
        }
    }
    void start(int n_workers, std::function<void(int)> on_start)
    {
        for (int r = 1; r <= n_workers; r++) th.emplace_back(&zs_crew::worker, this, r, on_start);
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            quit.store(true);
        }
        cv.notify_all();
        for (std::thread &t : th) t.join();
        th.clear();
    }
    /* f(r) for r = 0 .. n-1, each on its rank's thread; returns when all are done */
    template <class F> void run(int n, F &&f)
    {
        if (n <= 1 || th.empty()) {
            for (int r = 0; r < n; r++) f(r);
            return;
        }
        typedef typename std::remove_reference<F>::type Fn;
        ctx = (void *)&f;
        tramp = [](void *c, int r) { (*static_cast<Fn *>(c))(r); };
        n_active = n;
        left.store((int)th.size());
        gen.fetch_add(1);
        if (sleepers.load() > 0) {
            std::lock_guard<std::mutex> lk(m);
            cv.notify_all();
        }
        f(0);
        for (int spins = 0; left.load() != 0; spins++)
            if (spins < 20000) __builtin_ia32_pause(); else std::this_thread::yield();
    }
};
#endif
