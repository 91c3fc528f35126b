// ts_line_pool.h — the host threads that lay out and merge the report's lines of a damaged transport stream (ts_runtime.cpp).
// No HIP in here: the pool is checked on its own, also under ThreadSanitizer (tests/test_sanitizers.py).
#ifndef TS_LINE_POOL_H
#define TS_LINE_POOL_H

#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// A few host threads for the one part of a scan that is the host's and grows with the stream's damage: laying out and
// merging the report's lines (129 000 of them for one damaged spot per 1000 packets in 10 GiB).  The work comes as a BURST of
// short fork / join rounds (count, scatter, order, merge: 50-100 us each), so the workers are woken once per scan — as soon as
// the scan knows it has many lines, while the lines are still on their way over the link — and SPIN between the rounds (a
// wake-up through a condition variable costs as much as a round); run(n, f) calls f(0) ... f(n - 1), the caller taking its
// share.  Jobs are handed out by one atomic ticket that carries the round's number in its upper half, so that a worker that is
// late for one round can never run a job of it with the next round's function.
struct ts_line_pool {
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable wake;
    bool stop = false;
    std::atomic<bool> burst{false};
    // round << 48 | jobs of that round << 32 | next job: ONE word says which round a ticket belongs to, how many jobs it has and
    // which one this is — a worker never combines a ticket of one round with the job count or the function of another (it did,
    // once: `njobs` and `job` were read after the ticket, the main thread had moved on in between, and a stale ticket below
    // the NEW count ran one of the new round's jobs a second time: holes in the report and a `done` that overshot)
    std::atomic<uint64_t> ticket{0};
    std::atomic<uint32_t> done{0};
    // the function of round r is jobs[r & 1]: written before the round's ticket is, and not again before round r + 2 — by
    // which time every job of round r has long been counted in `done`
    const std::function<void(int)> *jobs[2] = {nullptr, nullptr};
    uint32_t round = 0;
    static constexpr uint32_t kMaxJobs = 0xFFFFu;
    static uint32_t jobs_of(uint64_t t) { return (uint32_t)(t >> 32) & 0xFFFFu; }
    static void relax() { __builtin_ia32_pause(); }
    void worker()
    {
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m);
                wake.wait(lk, [&] { return stop || burst.load(); });
                if (stop)
                    return;
            }
            while (burst.load(std::memory_order_acquire)) {
                uint64_t t = ticket.load(std::memory_order_acquire);
                if ((uint32_t)t >= jobs_of(t)) {
                    relax();
                    continue;  // (nothing left of this round)
                }
                t = ticket.fetch_add(1, std::memory_order_acq_rel);
                const uint32_t idx = (uint32_t)t;
                if (idx < jobs_of(t)) {
                    (*jobs[(t >> 48) & 1u])((int)idx);
                    done.fetch_add(1, std::memory_order_release);
                }
            }
        }
    }
    // The workers run on the cores that share the caller's L3 (one CCD): the rounds hand cache lines from thread to thread —
    // events counted by one are scattered by it, lines scattered by eight are merged by others — and across CCDs or sockets
    // that costs more than the threads save (measured on the 2-socket box: the scatter round 0.22 ms alone, 0.39 with four
    // threads wherever the scheduler put them).
    static bool l3_siblings(cpu_set_t *set)
    {
        const int cpu = sched_getcpu();
        if (cpu < 0)
            return false;
        char path[128], text[4096];
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
        FILE *f = fopen(path, "r");
        if (!f)
            return false;
        const bool got = fgets(text, sizeof(text), f) != nullptr;
        fclose(f);
        if (!got)
            return false;
        CPU_ZERO(set);
        int n = 0;
        for (char *q = text; *q && *q != '\n';) {  // "0-7,128-135"
            char *end = nullptr;
            const long a = strtol(q, &end, 10);
            if (end == q)
                return false;
            long b = a;
            q = end;
            if (*q == '-') {
                b = strtol(q + 1, &end, 10);
                q = end;
            }
            for (long c = a; c <= b && c < CPU_SETSIZE; c++, n++)
                CPU_SET((int)c, set);
            if (*q == ',')
                q++;
        }
        return n >= 2;
    }
    // How many CPUs the CALLER may run on (taskset / numactl / a launcher's binding): the pool never has more threads than that.
    static int allowed_cpus()
    {
        cpu_set_t mine;
        if (sched_getaffinity(0, sizeof(mine), &mine) != 0)
            return 0;
        return CPU_COUNT(&mine);
    }
    bool start(int nworkers)
    {
        cpu_set_t near, mine;
        bool pin = !(getenv("TS_HOST_PIN") && atoi(getenv("TS_HOST_PIN")) == 0) && l3_siblings(&near);
        // ... inside the caller's own affinity mask only: a CPU binding the user (or a launcher) chose is never widened, and
        // with fewer than two CPUs left in the intersection the workers are not pinned at all
        if (pin && sched_getaffinity(0, sizeof(mine), &mine) == 0) {
            CPU_AND(&near, &near, &mine);
            pin = CPU_COUNT(&near) >= 2;
        } else {
            pin = false;
        }
        const int allowed = allowed_cpus();
        if (allowed > 0)
            nworkers = std::min(nworkers, std::max(0, allowed - 1));
        try {
            for (int k = 0; k < nworkers; k++)
                workers.emplace_back([this, pin, near] {
                    if (pin)
                        (void)sched_setaffinity(0, sizeof(near), &near);
                    worker();
                });
        } catch (...) {
        }
        return !workers.empty();
    }
    void begin_burst()
    {
        if (workers.empty() || burst.load())
            return;
        {
            std::lock_guard<std::mutex> g(m);
            burst.store(true);
        }
        wake.notify_all();
    }
    void end_burst() { burst.store(false, std::memory_order_release); }
    void run(int n, const std::function<void(int)> &f)
    {
        if (workers.empty() || n <= 1 || (uint32_t)n > kMaxJobs || !burst.load()) {
            for (int k = 0; k < n; k++)
                f(k);
            return;
        }
        round++;
        jobs[round & 1u] = &f;
        done.store(0, std::memory_order_relaxed);
        ticket.store(((uint64_t)(round & 0xFFFFu) << 48) | ((uint64_t)(uint32_t)n << 32), std::memory_order_release);
        for (;;) {
            const uint64_t t = ticket.fetch_add(1, std::memory_order_acq_rel);
            if ((uint32_t)t >= (uint32_t)n)
                break;
            f((int)(uint32_t)t);
            done.fetch_add(1, std::memory_order_release);
        }
        while (done.load(std::memory_order_acquire) != (uint32_t)n)
            relax();
    }
    ~ts_line_pool()
    {
        {
            std::lock_guard<std::mutex> g(m);
            stop = true;
            burst.store(false);
        }
        wake.notify_all();
        for (std::thread &t : workers)
            t.join();
    }
};

// `rounds` rounds of 1 ... 64 jobs through `threads` threads (the caller's included): 0, or the first round in which a job did not
// run exactly once (ts_host_pool_selftest; tests/c/pool_harness.cpp runs it under ThreadSanitizer)
inline int ts_line_pool_selftest(int threads, int rounds)
{
    ts_line_pool pool;
    if (threads > 1)
        pool.start(threads - 1);
    pool.begin_burst();
    std::vector<std::atomic<uint32_t>> ran(64);
    int bad = 0;
    uint64_t x = 88172645463325252ull;
    for (int r = 1; r <= rounds && !bad; r++) {
        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
        const int n = 1 + (int)(x % 64u);  // (the counts jump about: a stale ticket of a short round lies inside a long one)
        for (auto &c : ran)
            c.store(0, std::memory_order_relaxed);
        const std::function<void(int)> f = [&](int k) {
            ran[(size_t)k].fetch_add(1, std::memory_order_relaxed);
            if ((x >> (k & 31)) & 1u)
                for (volatile int spin = 0; spin < 200; spin++) {
                }
        };
        pool.run(n, f);
        for (int k = 0; k < 64; k++)
            if (ran[(size_t)k].load() != (k < n ? 1u : 0u))
                bad = r;
    }
    pool.end_burst();
    return bad;
}

#endif
