// hub_harness — the in-process exchange (papr_exchange_open_local: n threads of one process meet at a hub) on its own, built with
// -fsanitize=thread by tests/test_sanitizers.py: `threads` threads run `rounds` rounds of the exchange's host-level collectives
// (papr_exchange_stats, papr_exchange_counts, the self-test's all-gather / all-reduce with ctx = NULL), check what comes back,
// then one of them cancels the exchange while the others wait in a collective (papr_exchange_abort: everyone is released with
// PAPR_E_STATE), and all close.  A third argument "async": the handles are papr_exchange_open_rccl_local_async's and are adopted
// first (see main).  Prints "ok" or what went wrong.  No GPU: the two runtime helpers the exchange links against are
// supplied here.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "papr_exchange.h"
#include "papr_hip.h"

namespace papr_rt {
double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}
}  // namespace papr_rt

int main(int argc, char **argv)
{
    if (argc != 3 && argc != 4)
        return 2;
    const int n = atoi(argv[1]), rounds = atoi(argv[2]);
    // "async": the handles come from papr_exchange_open_rccl_local_async — n set-up threads beside these — with the set-up made
    // to fail (PAPR_XCH_BIND_FAIL=all in the environment: no RCCL, no GPU is touched); every thread then adopts (all agree
    // through the hub that nobody has a communicator), and the handles must be the hub's for everything that follows
    const bool async = argc == 4 && !strcmp(argv[3], "async");
    std::vector<papr_exchange *> xs((size_t)n);
    std::vector<int> devices((size_t)n);
    for (int r = 0; r < n; r++)
        devices[(size_t)r] = r;
    if ((async ? papr_exchange_open_rccl_local_async(xs.data(), n, devices.data()) : papr_exchange_open_local(xs.data(), n)) != PAPR_OK) {
        printf("open failed: %s\n", papr_exchange_last_error(nullptr));
        return 1;
    }
    std::atomic<int> bad{0}, released{0};
    std::vector<std::thread> th;
    for (int r = 0; r < n; r++)
        th.emplace_back([&, r] {
            papr_exchange *x = xs[(size_t)r];
            if (async) {
                double setup = -1.0, waited = -1.0;
                // (the context is only looked into when a communicator is taken or ended: none exists here)
                if (papr_exchange_adopt_rccl(x, reinterpret_cast<papr_hip_ctx *>(&bad), 30.0, &setup, &waited) != PAPR_OK || papr_exchange_is_rccl(x) ||
                    waited < 0.0)
                    bad = 7;
                if (r == 0)  // (the library prints nothing: why there is no communicator is the handle's last error)
                    fprintf(stderr, "papr: %s\n", papr_exchange_last_error(x));
                if (papr_exchange_adopt_rccl(x, reinterpret_cast<papr_hip_ctx *>(&bad), 30.0, nullptr, nullptr) != PAPR_OK)  // (a second call: nothing pending)
                    bad = 8;
            }
            for (int k = 0; k < rounds && !bad.load(); k++) {
                papr_stats mine;
                memset(&mine, 0, sizeof(mine));
                mine.sum = (double)(r + 1) * (k + 1);
                mine.n = 1000u + (unsigned)r;
                mine.peak = (float)(r + k);
                mine.peak_idx = (unsigned long long)r;
                papr_stats total;
                double before = -1.0;
                std::vector<papr_stats> all((size_t)n);
                if (papr_exchange_stats(x, &mine, &total, &before, all.data()) != PAPR_OK) {
                    bad = 1;
                    break;
                }
                double want_before = 0.0, want_sum = 0.0;
                for (int q = 0; q < n; q++) {
                    if (q < r)
                        want_before += (double)(q + 1) * (k + 1);
                    want_sum += (double)(q + 1) * (k + 1);
                    if (all[(size_t)q].n != 1000u + (unsigned)q)
                        bad = 2;
                }
                if (total.sum != want_sum || before != want_before || total.n != (unsigned long long)n * 1000u + (unsigned long long)n * (n - 1) / 2)
                    bad = 3;
                uint64_t counts[31];
                for (int j = 0; j < 31; j++)
                    counts[j] = (uint64_t)(r + j + k);
                if (papr_exchange_counts(x, counts, 31) != PAPR_OK) {
                    bad = 4;
                    break;
                }
                for (int j = 0; j < 31; j++)
                    if (counts[j] != (uint64_t)n * (uint64_t)(j + k) + (uint64_t)n * (n - 1) / 2)
                        bad = 5;
                if ((k & 63) == 0 && papr_exchange_selftest(x, nullptr, 0) != PAPR_OK)
                    bad = 6;
            }
            // one rank gives up; the others are waiting for it in a collective
            if (r == n - 1) {
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
                papr_exchange_abort(x);
                released++;
            } else {
                papr_stats mine, total;
                memset(&mine, 0, sizeof(mine));
                double before;
                if (papr_exchange_stats(x, &mine, &total, &before, nullptr) == PAPR_E_STATE)
                    released++;
            }
        });
    for (auto &t : th)
        t.join();
    for (auto x : xs)
        papr_exchange_close(x);
    if (bad.load() || released.load() != n) {
        printf("failed: check %d, %d of %d released\n", bad.load(), released.load(), n);
        return 1;
    }
    printf("ok\n");
    return 0;
}
