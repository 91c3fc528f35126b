// papr_exchange.cpp — the exchange between the shards of a sharded papr run, behind the C ABI (include/papr_hip.h):
// all-gather of 96-byte pass-1 records + ordered merge, all-reduce of the per-level counters, all-gather of the
// exact-sum programs + chain.  Transport: RCCL (bound at run time with dlopen, so that a process that already
// carries an RCCL — PyTorch ships its own — uses that one instead of loading a second copy), or caller-supplied
// collectives (the CPU tests: gloo).  SURVEY.md 7.1 C1 / C2.

#include "papr_runtime_internal.h"

#include <atomic>
#include <chrono>
#include <memory>
#include <thread>
#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>  // types and enums only: the functions are looked up with dlsym

using namespace papr_rt;

namespace {

char g_xch_open_error[256] = "";

struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

RcclApi *rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // an RCCL that is already in the process first (RTLD_NOLOAD), then the ROCm installation's
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (int pass = 0; pass < 2 && !api.handle; pass++)
            for (const char *n : names) {
                api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (api.handle)
                    break;
            }
        if (!api.handle)
            api.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!api.handle)
            return;
#define SYM(name) api.name = reinterpret_cast<decltype(api.name)>(dlsym(api.handle, "nccl" #name))
        SYM(GetUniqueId);
        SYM(CommInitRank);
        SYM(CommDestroy);
        SYM(CommAbort);
        SYM(AllGather);
        SYM(AllReduce);
        SYM(Broadcast);
        SYM(GroupStart);
        SYM(GroupEnd);
        SYM(GetErrorString);
#undef SYM
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.AllReduce)
            api.handle = nullptr;
    });
    return api.handle ? &api : nullptr;
}

double now_us()
{
    return now_s() * 1e6;
}

}  // namespace

// in-process transport: the threads of one process meet at a barrier (bin/papr: one thread per GPU)
struct LocalHub {
    std::mutex m;
    std::condition_variable cv;
    int world = 0, arrived = 0, refs = 0;
    uint64_t generation = 0;
    std::vector<unsigned char> slots;  // world x bytes of the collective in flight
    size_t slot_bytes = 0;
    bool failed = false;  // a participant gave up (papr_exchange_abort): every collective fails from then on
    double timeout_s = 0; // PAPR_XCH_TIMEOUT_S: a thread that waits longer than this for its peers cancels the exchange
    // papr_exchange_open_rccl_local: the id every thread's ncclCommInitRank joins with (papr_exchange_bind)
    bool want_rccl = false;
    ncclUniqueId uid{};
    std::vector<papr_exchange *> members;  // (for papr_exchange_abort: every member's communicator is aborted)
    bool barrier(std::unique_lock<std::mutex> &lk)
    {
        if (failed)
            return false;
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
        } else if (timeout_s > 0) {
            if (!cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return generation != gen || failed; })) {
                failed = true;  // (the peers that do arrive later find the exchange cancelled instead of waiting in turn)
                cv.notify_all();
            }
        } else {
            cv.wait(lk, [&] { return generation != gen || failed; });
        }
        return !failed;
    }
};

// papr_exchange_open_rccl_local_async: what the n set-up threads and the n handles share.  The threads hold it through a
// shared_ptr and never touch a handle: one that is still inside ncclCommInitRank when its handle has given up on it (and is
// closed, or the process leaves) finds its memory alive.
struct BindShared {
    std::mutex m;
    std::condition_variable cv;
    int world = 0;
    bool uid_ready = false, failed = false, abandoned = false;
    ncclUniqueId uid{};
    char why[200] = "";
    struct Rank {
        int device = 0;
        bool done = false, ok = false, taken = false;
        ncclComm_t comm = nullptr;
        double t_begin = 0, t_end = 0;
    };
    std::vector<Rank> ranks;
};

struct papr_exchange {
    int rank = 0, world = 1;
    LocalHub *hub = nullptr;
    std::shared_ptr<BindShared> bind;  // communicators coming up beside the ingest (papr_exchange_adopt_rccl takes them)
    std::atomic<int> pins{0};          // papr_exchange_abort is working on this handle outside the hub's mutex: close waits
    char err[256] = "";
    // caller-supplied transport
    papr_exchange_ops ops{};
    bool use_ops = false;
    // RCCL transport
    papr_hip_ctx *ctx = nullptr;
    int device = -1;               // ctx's device, kept: papr_exchange_close must not look into a context its caller has closed already
    ncclComm_t comm = nullptr;
    std::atomic<bool> has_comm{false};  // `comm` is there (set behind it, by its owner): what papr_exchange_abort asks before it takes comm_m
    bool want_rccl = false;        // papr_exchange_open_rccl_local: papr_exchange_bind still has to create `comm`
    ncclUniqueId solo_uid{};       // ... for a world of one (no hub)
    std::atomic<bool> aborted{false};
    // `comm` is used (collectives queued) by its owner's thread and ended by whoever cancels the exchange: both under this
    // mutex, `aborted` checked inside it, so that nothing is ever queued on a communicator ncclCommAbort has freed.  Only
    // the owner's papr_exchange_close sets `comm` to null.
    std::timed_mutex comm_m;
    std::atomic<bool> comm_ended{false};  // ncclCommAbort has run on `comm` (it is freed: papr_exchange_close must not destroy it)
    char wd_err[256] = "";       // the watchdog's message (its own buffer: `err` belongs to the owner's thread)
    // PAPR_XCH_TIMEOUT_S: since when this rank has been waiting for its peers (0: it is not), and in what
    std::atomic<double> waiting_since{0.0};
    std::atomic<const char *> waiting_in{""};
    double timeout_s = 0;
    std::thread watchdog;
    std::atomic<bool> watchdog_stop{false};
    std::atomic<bool> timed_out{false};  // (set with release ordering AFTER wd_err is written: whoever sees it may read the text)
    bool selftest_done = false;
    unsigned char *d_send = nullptr, *d_recv = nullptr;  // device staging
    unsigned char *h_send = nullptr, *h_recv = nullptr;  // pinned mirrors
    size_t cap_send = 0, cap_recv = 0;
    std::vector<unsigned char> scratch;
    papr_exchange_timing timing{};
};

namespace {

int xfail(papr_exchange *x, int code, const char *fmt, ...)
{
    char *dst = x ? x->err : g_xch_open_error;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 256, fmt, ap);
    va_end(ap);
    return code;
}

int ensure_staging(papr_exchange *x, size_t send_bytes, size_t recv_bytes)
{
    if (send_bytes > x->cap_send) {
        if (x->d_send) (void)hipFree(x->d_send);
        if (x->h_send) (void)hipHostFree(x->h_send);
        x->d_send = x->h_send = nullptr;
        x->cap_send = 0;
        const size_t cap = std::max<size_t>(send_bytes, (size_t)PAPR_HIP_MAX_LEVELS * 8);
        if (hipMalloc((void **)&x->d_send, cap) != hipSuccess || hipHostMalloc((void **)&x->h_send, cap, hipHostMallocDefault) != hipSuccess)
            return xfail(x, PAPR_E_NOMEM, "cannot allocate %zu bytes of exchange staging", cap);
        x->cap_send = cap;
    }
    if (recv_bytes > x->cap_recv) {
        if (x->d_recv) (void)hipFree(x->d_recv);
        if (x->h_recv) (void)hipHostFree(x->h_recv);
        x->d_recv = x->h_recv = nullptr;
        x->cap_recv = 0;
        const size_t cap = std::max<size_t>(recv_bytes, (size_t)PAPR_HIP_MAX_LEVELS * 8);
        if (hipMalloc((void **)&x->d_recv, cap) != hipSuccess || hipHostMalloc((void **)&x->h_recv, cap, hipHostMallocDefault) != hipSuccess)
            return xfail(x, PAPR_E_NOMEM, "cannot allocate %zu bytes of exchange staging", cap);
        x->cap_recv = cap;
    }
    return PAPR_OK;
}

#define XHIP(x, call)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return xfail(x, PAPR_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_));         \
    } while (0)
// Every RCCL call on x->comm: under the communicator's mutex, and never once the exchange was cancelled (the communicator
// may be freed by then: papr_exchange_abort).
const char *const kCancelled = "the exchange was cancelled (papr_exchange_abort): nothing is queued on its communicator";
#define XNCCL(x, call)                                                                          \
    do {                                                                                        \
        std::lock_guard<std::timed_mutex> g_((x)->comm_m);                                      \
        if ((x)->aborted.load() || !(x)->comm)                                                  \
            return xfail(x, PAPR_E_STATE, "%s", kCancelled);                                    \
        ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return xfail(x, PAPR_E_HIP, "%s failed: %s", #call,                                 \
                         rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "RCCL error");   \
    } while (0)

// PAPR_XCH_TIMEOUT_S: "this rank is waiting for its peers, since now, in <what>" for the length of a scope
struct Waiting {
    papr_exchange *x;
    Waiting(papr_exchange *x_, const char *what) : x(x_)
    {
        if (x && x->timeout_s > 0) {
            x->waiting_in.store(what);
            x->waiting_since.store(now_s());
        }
    }
    ~Waiting()
    {
        if (x && x->timeout_s > 0)
            x->waiting_since.store(0.0);
    }
};

// The watchdog of one handle: a rank that has waited longer than PAPR_XCH_TIMEOUT_S for its peers — in a host exchange's
// wait, in the step's one wait behind in-stream collectives, in ncclCommInitRank — says so on stderr and cancels the
// exchange (papr_exchange_abort: the wait returns with an error, the peers are released).  Inside ncclCommInitRank there
// is no communicator to abort yet: the process is ended with status 254 — the only way out of a rendezvous that never
// completes (PAPR_XCH_TIMEOUT_EXIT=0: only the message).
void watchdog_main(papr_exchange *x)
{
    while (!x->watchdog_stop.load()) {
        usleep(50 * 1000);
        const double since = x->waiting_since.load();
        if (since <= 0 || now_s() - since <= x->timeout_s)
            continue;
        const char *what = x->waiting_in.load();
        snprintf(x->wd_err, sizeof(x->wd_err), "rank %d of %d waited more than %g s for its peers in %s (PAPR_XCH_TIMEOUT_S): "
                 "the exchange is cancelled", x->rank, x->world, x->timeout_s, what);
        x->timed_out.store(true, std::memory_order_release);
        fprintf(stderr, "papr exchange: %s\n", x->wd_err);
        fflush(stderr);
        x->waiting_since.store(0.0);
        const bool in_init = !strcmp(what, "ncclCommInitRank");
        papr_exchange_abort(x);
        if (in_init && env_int("PAPR_XCH_TIMEOUT_EXIT", 1))
            _exit(254);
    }
}

void start_watchdog(papr_exchange *x)
{
    const char *e = getenv("PAPR_XCH_TIMEOUT_S");
    const double t = e ? atof(e) : 0.0;
    if (!(t > 0) || x->watchdog.joinable())
        return;
    x->timeout_s = t;
    if (x->hub) {
        std::lock_guard<std::mutex> g(x->hub->m);
        x->hub->timeout_s = t;
    }
    try {
        x->watchdog = std::thread(watchdog_main, x);
    } catch (...) {
        x->timeout_s = 0;  // (no thread: no watchdog; the hub's own timed waits still hold)
    }
}

// recv = world x bytes_per_rank, in rank order
int allgather_bytes(papr_exchange *x, const void *send, void *recv, size_t bytes_per_rank)
{
    if (x->world == 1 && (x->use_ops || !x->comm)) {  // (an RCCL communicator of one rank still runs its collectives:
        memcpy(recv, send, bytes_per_rank);            // that is what `torchrun --nproc-per-node 1` measures)
        return PAPR_OK;
    }
    if (x->hub) {
        LocalHub &h = *x->hub;
        std::unique_lock<std::mutex> lk(h.m);
        const char *gone = "another shard's thread gave up (or PAPR_XCH_TIMEOUT_S ran out): the exchange is cancelled";
        if (!h.barrier(lk))  // everyone has left the previous collective
            return xfail(x, PAPR_E_STATE, "%s", gone);
        if (h.slots.size() != bytes_per_rank * (size_t)h.world) {
            try {
                h.slots.assign(bytes_per_rank * (size_t)h.world, 0);
            } catch (...) {
                h.failed = true;
                h.cv.notify_all();
                return xfail(x, PAPR_E_NOMEM, "out of host memory");
            }
        }
        h.slot_bytes = bytes_per_rank;
        if (!h.barrier(lk))  // the buffer has its size
            return xfail(x, PAPR_E_STATE, "%s", gone);
        memcpy(h.slots.data() + (size_t)x->rank * bytes_per_rank, send, bytes_per_rank);
        if (!h.barrier(lk))  // every slot is written
            return xfail(x, PAPR_E_STATE, "%s", gone);
        memcpy(recv, h.slots.data(), bytes_per_rank * (size_t)h.world);
        return PAPR_OK;
    }
    if (x->use_ops) {
        if (x->ops.allgather(x->ops.user, send, recv, bytes_per_rank) != 0)
            return xfail(x, PAPR_E_HIP, "the caller's all-gather failed");
        return PAPR_OK;
    }
    const size_t total = bytes_per_rank * (size_t)x->world;
    int rc = ensure_staging(x, bytes_per_rank, total);
    if (rc)
        return rc;
    papr_hip_ctx *ctx = x->ctx;
    XHIP(x, hipSetDevice(ctx->device));
    memcpy(x->h_send, send, bytes_per_rank);
    XHIP(x, hipMemcpyAsync(x->d_send, x->h_send, bytes_per_rank, hipMemcpyHostToDevice, ctx->stream));
    Waiting w(x, "a host-level all-gather over RCCL");
    XNCCL(x, rccl()->AllGather(x->d_send, x->d_recv, bytes_per_rank, ncclUint8, x->comm, ctx->stream));
    XHIP(x, hipMemcpyAsync(x->h_recv, x->d_recv, total, hipMemcpyDeviceToHost, ctx->stream));
    XHIP(x, hipStreamSynchronize(ctx->stream));
    if (x->aborted.load())
        return xfail(x, PAPR_E_STATE, "%s", x->timed_out.load(std::memory_order_acquire) ? x->wd_err : kCancelled);
    memcpy(recv, x->h_recv, total);
    return PAPR_OK;
}

int allreduce_u64(papr_exchange *x, uint64_t *buf, size_t count)
{
    if ((x->world == 1 && (x->use_ops || !x->comm)) || count == 0)
        return PAPR_OK;
    if (x->hub) {  // gather, then every thread adds the slots in rank order
        std::vector<uint64_t> all;
        try {
            all.resize(count * (size_t)x->world);
        } catch (...) {
            return xfail(x, PAPR_E_NOMEM, "out of host memory");
        }
        int rc = allgather_bytes(x, buf, all.data(), count * sizeof(uint64_t));
        if (rc)
            return rc;
        for (size_t k = 0; k < count; k++) {
            uint64_t sum = 0;
            for (int r = 0; r < x->world; r++)
                sum += all[(size_t)r * count + k];
            buf[k] = sum;
        }
        return PAPR_OK;
    }
    if (x->use_ops) {
        if (x->ops.allreduce_sum_u64(x->ops.user, buf, count) != 0)
            return xfail(x, PAPR_E_HIP, "the caller's all-reduce failed");
        return PAPR_OK;
    }
    const size_t bytes = count * sizeof(uint64_t);
    int rc = ensure_staging(x, bytes, bytes);
    if (rc)
        return rc;
    papr_hip_ctx *ctx = x->ctx;
    XHIP(x, hipSetDevice(ctx->device));
    memcpy(x->h_send, buf, bytes);
    XHIP(x, hipMemcpyAsync(x->d_send, x->h_send, bytes, hipMemcpyHostToDevice, ctx->stream));
    Waiting w(x, "a host-level all-reduce over RCCL");
    XNCCL(x, rccl()->AllReduce(x->d_send, x->d_recv, count, ncclUint64, ncclSum, x->comm, ctx->stream));
    XHIP(x, hipMemcpyAsync(x->h_recv, x->d_recv, bytes, hipMemcpyDeviceToHost, ctx->stream));
    XHIP(x, hipStreamSynchronize(ctx->stream));
    if (x->aborted.load())
        return xfail(x, PAPR_E_STATE, "%s", x->timed_out.load(std::memory_order_acquire) ? x->wd_err : kCancelled);
    memcpy(buf, x->h_recv, bytes);
    return PAPR_OK;
}

}  // namespace

extern "C" {

const char *papr_exchange_last_error(const papr_exchange *x)
{
    if (x && x->timed_out && x->wd_err[0])
        return x->wd_err;
    return x ? x->err : g_xch_open_error;
}

int papr_exchange_unique_id(void *id)
{
    static_assert(sizeof(ncclUniqueId) == PAPR_EXCHANGE_ID_BYTES, "ncclUniqueId size");
    if (!id)
        return PAPR_E_ARG;
    if (!rccl())
        return xfail(nullptr, PAPR_E_NO_DEVICE, "RCCL (librccl.so) could not be loaded");
    ncclUniqueId uid;
    if (rccl()->GetUniqueId(&uid) != ncclSuccess)
        return xfail(nullptr, PAPR_E_HIP, "ncclGetUniqueId failed");
    memcpy(id, &uid, sizeof(uid));
    return PAPR_OK;
}

// (internal) nobody to talk to and nothing to run: every exchange of this handle is the identity
bool papr_exchange_is_identity(const papr_exchange *x)
{
    return !x || (x->world == 1 && x->use_ops);
}

}  // extern "C"

// ---- in-stream collectives on device buffers (papr_sweep_rt.cpp: the single-wait step with peers) -----------------
// Only the RCCL transport has them: the collective is queued on the context's stream between the kernels that produce
// and consume its buffers — no host staging, no wait.  The caller-callback and in-process transports move host memory.
namespace papr_rt {

// PAPR_XCH_IN_STREAM=2 lets the host transports (caller callbacks, threads) STAND IN for them — the device buffer is
// brought to the host behind a wait, crosses the transport, and goes back — so that the sharded step's device side (the
// record kernels, the guess from all ranks' records, the ordered merge, the packed counters) can be run with real worlds
// of 2-8 ranks on a box with one GPU (tests/test_exchange_gloo.py, bench.py --backend gloo).  Not a fast path.
bool xch_in_stream(const papr_exchange *x, const papr_hip_ctx *ctx)
{
    if (!x)
        return false;
    const int mode = env_int("PAPR_XCH_IN_STREAM", 1);
    if (x->comm)
        return x->ctx == ctx && mode != 0;
    return mode == 2 && (x->hub || x->use_ops) && !(x->world == 1 && x->use_ops);
}
int xch_allgather_host(papr_exchange *x, const void *send, void *recv, size_t bytes_per_rank)
{
    return allgather_bytes(x, send, recv, bytes_per_rank);
}
int xch_rank(const papr_exchange *x) { return x ? x->rank : 0; }
int xch_world(const papr_exchange *x) { return x ? x->world : 1; }
// PAPR_XCH_TIMEOUT_S around the step's ONE wait behind in-stream collectives (papr_sweep_rt.cpp): a peer that never
// enters its collective leaves this rank's stream waiting; the watchdog then cancels the exchange and the wait returns.
void xch_wait_begin(papr_exchange *x, const char *what)
{
    if (x && x->timeout_s > 0) {
        x->waiting_in = what;
        x->waiting_since.store(now_s());
    }
}
void xch_wait_end(papr_exchange *x)
{
    if (x && x->timeout_s > 0)
        x->waiting_since.store(0.0);
}
bool xch_cancelled(const papr_exchange *x) { return x && (x->aborted.load() || x->timed_out); }

namespace {
// the stand-in: device -> host, the transport's own collective, host -> device (the stream is idle in between)
int through_the_host(papr_exchange *x, papr_hip_ctx *ctx, const void *send_dev, void *recv_dev, size_t send_bytes,
                     size_t recv_bytes, bool reduce_u64)
{
    try {
        x->scratch.resize(send_bytes + recv_bytes);
    } catch (...) {
        return xfail(x, PAPR_E_NOMEM, "out of host memory");
    }
    unsigned char *hs = x->scratch.data(), *hr = hs + send_bytes;
    XHIP(x, hipMemcpyAsync(hs, send_dev, send_bytes, hipMemcpyDeviceToHost, ctx->stream));
    XHIP(x, hipStreamSynchronize(ctx->stream));
    int rc;
    if (reduce_u64) {
        memcpy(hr, hs, send_bytes);
        rc = allreduce_u64(x, reinterpret_cast<uint64_t *>(hr), send_bytes / sizeof(uint64_t));
    } else {
        rc = allgather_bytes(x, hs, hr, send_bytes);
    }
    if (rc)
        return rc;
    XHIP(x, hipMemcpyAsync(recv_dev, hr, recv_bytes, hipMemcpyHostToDevice, ctx->stream));
    XHIP(x, hipStreamSynchronize(ctx->stream));  // (`scratch` is reused by the next call)
    x->timing.in_stream_calls++;
    return PAPR_OK;
}
}  // namespace

int xch_allgather_dev(papr_exchange *x, papr_hip_ctx *ctx, const void *send_dev, void *recv_dev, size_t bytes_per_rank)
{
    if (!x->comm)
        return through_the_host(x, ctx, send_dev, recv_dev, bytes_per_rank, bytes_per_rank * (size_t)x->world, false);
    XNCCL(x, rccl()->AllGather(send_dev, recv_dev, bytes_per_rank, ncclUint8, x->comm, x->ctx->stream));
    x->timing.in_stream_calls++;
    return PAPR_OK;
}

// every rank's block of its own size: recv_dev + offs[r] receives rank r's sizes[r] bytes (sizes / offs: the same on every
// rank).  RCCL: one broadcast per rank inside a group call — one launch; the stand-in pads to the largest.
int xch_allgatherv_dev(papr_exchange *x, papr_hip_ctx *ctx, const void *send_dev, void *recv_dev, const size_t *sizes,
                       const size_t *offs)
{
    const int world = x->world, me = x->rank;
    if (!x->comm) {
        size_t big = 0;
        for (int r = 0; r < world; r++)
            big = std::max(big, sizes[r]);
        try {
            x->scratch.resize(big * (size_t)(world + 1));
        } catch (...) {
            return xfail(x, PAPR_E_NOMEM, "out of host memory");
        }
        unsigned char *hs = x->scratch.data(), *hr = hs + big;
        memset(hs, 0, big);
        XHIP(x, hipMemcpyAsync(hs, send_dev, sizes[me], hipMemcpyDeviceToHost, ctx->stream));
        XHIP(x, hipStreamSynchronize(ctx->stream));
        std::vector<unsigned char> mine(hs, hs + big);  // (allgather_bytes may use x->scratch itself: not here, but keep the send apart)
        int rc = allgather_bytes(x, mine.data(), hr, big);
        if (rc)
            return rc;
        for (int r = 0; r < world; r++)
            XHIP(x, hipMemcpyAsync((unsigned char *)recv_dev + offs[r], hr + (size_t)r * big, sizes[r], hipMemcpyHostToDevice, ctx->stream));
        XHIP(x, hipStreamSynchronize(ctx->stream));
        x->timing.in_stream_calls++;
        return PAPR_OK;
    }
    if (!rccl()->Broadcast || !rccl()->GroupStart || !rccl()->GroupEnd)
        return xfail(x, PAPR_E_NO_DEVICE, "this RCCL has no ncclBroadcast / group calls");
    {
        std::lock_guard<std::timed_mutex> g(x->comm_m);  // (the whole group under the communicator's mutex)
        if (x->aborted.load() || !x->comm)
            return xfail(x, PAPR_E_STATE, "%s", kCancelled);
        ncclResult_t rr = rccl()->GroupStart();
        for (int r = 0; r < world && rr == ncclSuccess; r++) {
            unsigned char *dst = (unsigned char *)recv_dev + offs[r];
            rr = rccl()->Broadcast(r == me ? send_dev : (const void *)dst, dst, sizes[r], ncclUint8, r, x->comm, x->ctx->stream);
        }
        const ncclResult_t re = rccl()->GroupEnd();
        if (rr == ncclSuccess)
            rr = re;
        if (rr != ncclSuccess)
            return xfail(x, PAPR_E_HIP, "the grouped ncclBroadcasts of the all-gather-v failed: %s",
                         rccl()->GetErrorString ? rccl()->GetErrorString(rr) : "RCCL error");
    }
    x->timing.in_stream_calls++;
    return PAPR_OK;
}

int xch_allreduce_u64_dev(papr_exchange *x, papr_hip_ctx *ctx, const void *send_dev, void *recv_dev, size_t count)
{
    if (!x->comm)
        return through_the_host(x, ctx, send_dev, recv_dev, count * sizeof(uint64_t), count * sizeof(uint64_t), true);
    XNCCL(x, rccl()->AllReduce(send_dev, recv_dev, count, ncclUint64, ncclSum, x->comm, x->ctx->stream));
    x->timing.in_stream_calls++;
    return PAPR_OK;
}

}  // namespace papr_rt

extern "C" {

int papr_exchange_open_rccl(papr_exchange **out, papr_hip_ctx *ctx, const void *id, int rank, int world)
{
    if (!out || !ctx || !id || world < 1 || rank < 0 || rank >= world)
        return PAPR_E_ARG;
    *out = nullptr;
    if (!rccl())
        return xfail(nullptr, PAPR_E_NO_DEVICE, "RCCL (librccl.so) could not be loaded");
    papr_exchange *x = new (std::nothrow) papr_exchange();
    if (!x)
        return PAPR_E_NOMEM;
    x->rank = rank;
    x->world = world;
    x->ctx = ctx;
    x->device = ctx->device;
    if (hipSetDevice(ctx->device) != hipSuccess) {
        delete x;
        return xfail(nullptr, PAPR_E_HIP, "hipSetDevice(%d) failed", ctx->device);
    }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    start_watchdog(x);
    ncclResult_t r;
    {
        Waiting w(x, "ncclCommInitRank");
        r = rccl()->CommInitRank(&x->comm, world, uid, rank);
    }
    if (r != ncclSuccess) {
        xfail(nullptr, PAPR_E_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world,
              rccl()->GetErrorString ? rccl()->GetErrorString(r) : "RCCL error");
        x->comm = nullptr;
        papr_exchange_close(x);
        return PAPR_E_HIP;
    }
    x->has_comm.store(true, std::memory_order_release);
    *out = x;
    return PAPR_OK;
}

int papr_exchange_open_ops(papr_exchange **out, const papr_exchange_ops *ops, int rank, int world)
{
    if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && (!ops || !ops->allgather || !ops->allreduce_sum_u64)))
        return PAPR_E_ARG;
    papr_exchange *x = new (std::nothrow) papr_exchange();
    if (!x)
        return PAPR_E_NOMEM;
    x->rank = rank;
    x->world = world;
    x->use_ops = true;
    if (ops)
        x->ops = *ops;
    *out = x;
    return PAPR_OK;
}

int papr_exchange_open_local(papr_exchange **xs, int n)
{
    if (!xs || n < 1)
        return PAPR_E_ARG;
    LocalHub *hub = new (std::nothrow) LocalHub();
    if (!hub)
        return PAPR_E_NOMEM;
    hub->world = n;
    hub->refs = n;
    for (int r = 0; r < n; r++) {
        papr_exchange *x = new (std::nothrow) papr_exchange();
        if (!x) {
            for (int q = 0; q < r; q++)
                delete xs[q];
            delete hub;
            return PAPR_E_NOMEM;
        }
        x->rank = r;
        x->world = n;
        x->use_ops = true;  // (world == 1 short cuts apply)
        x->hub = n > 1 ? hub : nullptr;
        xs[r] = x;
    }
    if (n == 1) {
        delete hub;
    } else {
        const char *e = getenv("PAPR_XCH_TIMEOUT_S");  // (threads that meet at the hub: its waits are timed themselves)
        hub->timeout_s = e && atof(e) > 0 ? atof(e) : 0.0;
    }
    return PAPR_OK;
}

int papr_exchange_open_rccl_local(papr_exchange **xs, int n)
{
    if (!xs || n < 1)
        return PAPR_E_ARG;
    if (!rccl())
        return xfail(nullptr, PAPR_E_NO_DEVICE, "RCCL (librccl.so) could not be loaded");
    ncclUniqueId uid;
    if (rccl()->GetUniqueId(&uid) != ncclSuccess)
        return xfail(nullptr, PAPR_E_HIP, "ncclGetUniqueId failed");
    int rc = papr_exchange_open_local(xs, n);
    if (rc)
        return rc;
    for (int r = 0; r < n; r++) {
        xs[r]->want_rccl = true;
        xs[r]->solo_uid = uid;
        xs[r]->use_ops = n > 1;  // (a world of one still runs its collectives through RCCL: no identity short cut)
    }
    if (n > 1) {
        LocalHub *hub = xs[0]->hub;
        hub->want_rccl = true;
        hub->uid = uid;
        hub->members.assign(xs, xs + n);
    }
    return PAPR_OK;
}

int papr_exchange_bind(papr_exchange *x, papr_hip_ctx *ctx)
{
    if (!x || !ctx)
        return PAPR_E_ARG;
    if (!x->want_rccl)
        return PAPR_OK;  // (the other transports have nothing to bind)
    x->want_rccl = false;
    // One communicator cannot hold two ranks of one device (bin/papr with PAPR_OVERSUBSCRIBE): the threads settle
    // through the hub whether every shard has a GPU of its own; if not, the handles stay what papr_exchange_open_local
    // made them.
    // Every thread selects its device FIRST and the threads tell each other through the hub whether that worked: a thread
    // that cannot go on must say so before anybody is inside ncclCommInitRank, which only returns when all ranks joined.
    start_watchdog(x);
    const bool dev_ok = hipSetDevice(ctx->device) == hipSuccess;
    if (x->hub) {
        std::vector<uint64_t> devs((size_t)x->world);
        const uint64_t mine = (uint64_t)(uint32_t)ctx->device | (dev_ok ? 0ull : 1ull << 32);
        int rc = allgather_bytes(x, &mine, devs.data(), sizeof(uint64_t));
        if (rc)
            return rc;
        for (uint64_t d : devs)
            if (d >> 32)
                return dev_ok ? xfail(x, PAPR_E_STATE, "another shard's thread could not select its device (hipSetDevice(%d))", (int)(uint32_t)d)
                              : xfail(x, PAPR_E_HIP, "hipSetDevice(%d) failed", ctx->device);
        std::sort(devs.begin(), devs.end());
        if (std::adjacent_find(devs.begin(), devs.end()) != devs.end())
            return PAPR_OK;
    }
    if (!dev_ok)
        return xfail(x, PAPR_E_HIP, "hipSetDevice(%d) failed", ctx->device);
    const ncclUniqueId uid = x->hub ? x->hub->uid : x->solo_uid;
    ncclResult_t r;
    {
        Waiting w(x, "ncclCommInitRank");
        r = rccl()->CommInitRank(&x->comm, x->world, uid, x->rank);  // (every thread is in here at once)
    }
    if (r != ncclSuccess) {
        x->comm = nullptr;
        return xfail(x, PAPR_E_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", x->rank, x->world,
                     rccl()->GetErrorString ? rccl()->GetErrorString(r) : "RCCL error");
    }
    x->ctx = ctx;
    x->device = ctx->device;
    x->has_comm.store(true, std::memory_order_release);
    return PAPR_OK;
}

// ---- the communicators coming up BESIDE the ingest (bin/papr) ------------------------------------------------------------
// ncclCommInitRank costs seconds around a step of milliseconds and the first collective is only needed once the shards are
// loaded: n detached threads (rank 0 first loads librccl and makes the id) select their devices and join the communicator
// while the shards' own threads open their contexts and ingest.  The handles are papr_exchange_open_local's until
// papr_exchange_adopt_rccl — host-level exchanges meet at the hub all along — and stay that if RCCL is missing, fails, or is
// not up in time: a fall-back with one line on stderr, never an error.
static void bind_thread_main(std::shared_ptr<BindShared> sp, int r)
{
    BindShared &b = *sp;
    auto finish = [&](bool ok, ncclComm_t comm, const char *why) {
        std::lock_guard<std::mutex> g(b.m);
        BindShared::Rank &me = b.ranks[(size_t)r];
        me.done = true;
        me.ok = ok;
        me.comm = comm;
        me.t_end = now_s();
        if (!ok && !b.failed) {
            b.failed = true;
            snprintf(b.why, sizeof(b.why), "%s", why);
        }
        if (ok && b.abandoned && comm && rccl() && rccl()->CommAbort) {  // nobody will take it any more
            (void)rccl()->CommAbort(comm);
            me.comm = nullptr;
        }
        b.cv.notify_all();
    };
    {
        std::lock_guard<std::mutex> g(b.m);
        b.ranks[(size_t)r].t_begin = now_s();
    }
    const char *inject = getenv("PAPR_XCH_BIND_FAIL");  // tests: "all", or the rank that fails
    const bool injected = inject && inject[0] && (!strcmp(inject, "all") || atoi(inject) == r);
    const int delay_ms = env_int("PAPR_XCH_BIND_DELAY_MS", 0);  // tests: a set-up that is not ready in time
    if (delay_ms > 0)
        usleep((useconds_t)delay_ms * 1000);
    if (r == 0) {
        RcclApi *api = injected ? nullptr : rccl();
        ncclUniqueId uid{};
        if (!api || api->GetUniqueId(&uid) != ncclSuccess)
            return finish(false, nullptr, injected ? "an injected failure (PAPR_XCH_BIND_FAIL)" : !api ? "librccl.so could not be loaded" : "ncclGetUniqueId failed");
        std::lock_guard<std::mutex> g(b.m);
        b.uid = uid;
        b.uid_ready = true;
        b.cv.notify_all();
    } else {
        std::unique_lock<std::mutex> lk(b.m);
        b.cv.wait(lk, [&] { return b.uid_ready || b.failed || b.abandoned; });
        if (!b.uid_ready) {
            lk.unlock();
            return finish(false, nullptr, "another rank's set-up failed");
        }
    }
    if (injected)
        return finish(false, nullptr, "an injected failure (PAPR_XCH_BIND_FAIL)");
    if (hipSetDevice(b.ranks[(size_t)r].device) != hipSuccess)
        return finish(false, nullptr, "hipSetDevice failed in the set-up thread");
    ncclComm_t comm = nullptr;
    const ncclResult_t rr = rccl()->CommInitRank(&comm, b.world, b.uid, r);  // (returns when every rank has joined)
    if (rr != ncclSuccess) {
        char why[200];
        snprintf(why, sizeof(why), "ncclCommInitRank(rank %d of %d) failed: %s", r, b.world,
                 rccl()->GetErrorString ? rccl()->GetErrorString(rr) : "RCCL error");
        return finish(false, nullptr, why);
    }
    finish(true, comm, "");
}

int papr_exchange_open_rccl_local_async(papr_exchange **xs, int n, const int *devices)
{
    if (!xs || n < 1 || !devices)
        return PAPR_E_ARG;
    int rc = papr_exchange_open_local(xs, n);
    if (rc)
        return rc;
    // one communicator cannot hold two ranks of one device (PAPR_OVERSUBSCRIBE): such shards meet at the hub
    std::vector<int> devs(devices, devices + n);
    std::sort(devs.begin(), devs.end());
    if (std::adjacent_find(devs.begin(), devs.end()) != devs.end() && !env_int("PAPR_XCH_BIND_SHARED_OK", 0))  // (tests: the
        return PAPR_OK;                                                           // agreement path on a box with one GPU)
    try {
        auto sp = std::make_shared<BindShared>();
        sp->world = n;
        sp->ranks.resize((size_t)n);
        for (int r = 0; r < n; r++)
            sp->ranks[(size_t)r].device = devices[r];
        if (n > 1)
            xs[0]->hub->members.assign(xs, xs + n);  // (papr_exchange_abort ends every member's communicator)
        for (int r = 0; r < n; r++)
            xs[r]->bind = sp;
        for (int r = 0; r < n; r++)
            std::thread(bind_thread_main, sp, r).detach();
    } catch (...) {  // no thread / no memory: whatever was started fails alone; the handles stay the hub's
        for (int r = 0; r < n; r++)
            if (xs[r]->bind) {
                std::lock_guard<std::mutex> g(xs[r]->bind->m);
                xs[r]->bind->failed = true;
                snprintf(xs[r]->bind->why, sizeof(xs[r]->bind->why), "the set-up threads could not be started");
                xs[r]->bind->cv.notify_all();
            }
    }
    return PAPR_OK;
}

int papr_exchange_adopt_rccl(papr_exchange *x, papr_hip_ctx *ctx, double timeout_s, double *setup_s, double *waited_s)
{
    if (setup_s)
        *setup_s = 0.0;
    if (waited_s)
        *waited_s = 0.0;
    if (!x || !ctx)
        return PAPR_E_ARG;
    if (!x->bind)
        return PAPR_OK;  // (nothing is coming up: another transport, shards that share a device, or adopted already)
    std::shared_ptr<BindShared> sp = x->bind;
    BindShared &b = *sp;
    const double t0 = now_s();
    uint64_t mine = 0;  // 1: this rank's communicator is there
    char why[200] = "";
    {
        std::unique_lock<std::mutex> lk(b.m);
        BindShared::Rank &me = b.ranks[(size_t)x->rank];
        auto ready = [&] { return me.done || b.failed; };
        if (timeout_s > 0)
            b.cv.wait_for(lk, std::chrono::duration<double>(timeout_s), ready);
        mine = me.done && me.ok && !b.failed ? 1 : 0;
        if (b.failed)
            snprintf(why, sizeof(why), "%s", b.why);
        else if (!me.done)
            snprintf(why, sizeof(why), "ncclCommInitRank was not done %s", timeout_s > 0 ? "within PAPR_XCH_BIND_TIMEOUT_S" : "when the shards were loaded");
        if (setup_s && me.done)
            *setup_s = me.t_end - me.t_begin;
    }
    if (waited_s)
        *waited_s = now_s() - t0;
    // every shard's thread is here: RCCL for all of them or for none
    uint64_t all = mine;
    if (x->hub) {
        std::vector<uint64_t> flags((size_t)x->world);
        int rc = allgather_bytes(x, &mine, flags.data(), sizeof(uint64_t));
        if (rc)
            return rc;
        for (uint64_t f : flags)
            all &= f;
    }
    ncclComm_t comm = nullptr;
    {
        std::lock_guard<std::mutex> g(b.m);
        BindShared::Rank &me = b.ranks[(size_t)x->rank];
        if (me.done && me.comm && !me.taken) {
            comm = me.comm;
            me.taken = true;
        }
        if (!all)
            b.abandoned = true;  // (a communicator that still arrives is ended by the thread that made it)
        b.cv.notify_all();
    }
    x->bind.reset();
    if (all && comm) {
        x->comm = comm;
        x->has_comm.store(true, std::memory_order_release);
        x->ctx = ctx;
        x->device = ctx->device;
        x->use_ops = x->world > 1;  // (a world of one runs its collectives through RCCL: no identity short cut)
        start_watchdog(x);
        return PAPR_OK;
    }
    if (comm && rccl() && rccl()->CommAbort) {  // (not destroy: that may wait for peers that never joined)
        (void)hipSetDevice(ctx->device);
        (void)rccl()->CommAbort(comm);
    }
    // why not, for whoever wants to say so (papr_exchange_last_error: bin/papr prints it when RCCL was ASKED for — by default a
    // drop-in's stderr stays the reference's)
    xfail(x, PAPR_OK, "RCCL set-up did not complete (%s): the shards' results meet at the in-process hub instead",
          why[0] ? why : "another shard's communicator is missing");
    return PAPR_OK;
}

int papr_exchange_is_rccl(const papr_exchange *x)
{
    return x && x->comm ? 1 : 0;
}

void papr_exchange_abort(papr_exchange *x)
{
    if (!x)
        return;
    // End one member's communicator.  `aborted` first: its owner queues nothing more (XNCCL).  Then ncclCommAbort under the
    // communicator's mutex — unless the owner holds it for longer than a moment, which means it is INSIDE an RCCL call
    // waiting for a peer (the first collective of a kind connects the ranks and returns when all of them called it): that
    // is the wait ncclCommAbort exists to end, so it is called without the mutex then.
    auto end_comm = [](papr_exchange *m) {
        m->aborted.store(true);
        if (!m->has_comm.load(std::memory_order_acquire))
            return;  // (a handle of the hub alone: nothing to end, no mutex to take)
        const bool got = m->comm_m.try_lock_for(std::chrono::milliseconds(500));
        // (`comm` is written by its owner before any collective and by its close; read here without the mutex only when the
        // owner sits inside an RCCL call and will not touch it)
        if (m->comm && !m->comm_ended.exchange(true) && rccl() && rccl()->CommAbort)
            (void)rccl()->CommAbort(m->comm);
        if (got)
            m->comm_m.unlock();
    };
    if (x->hub) {
        // The hub's mutex only for as long as it takes to mark the exchange cancelled and to PIN the members
        // (papr_exchange_close strikes a handle off `members` under the same mutex and then waits for its pins to go before
        // it deletes the handle): the communicators are ended outside it — up to 500 ms each when an owner sits inside an
        // RCCL call — so that peers entering a hub barrier or closing are not held up for that long.
        std::vector<papr_exchange *> todo;
        {
            std::lock_guard<std::mutex> g(x->hub->m);
            x->hub->failed = true;
            x->hub->cv.notify_all();
            try {
                for (papr_exchange *m : x->hub->members)
                    if (m) {
                        todo.push_back(m);
                        m->pins.fetch_add(1);
                    }
            } catch (...) {  // (no memory for the list: end them under the mutex after all)
                for (papr_exchange *m : todo)
                    m->pins.fetch_sub(1);
                todo.clear();
                for (papr_exchange *m : x->hub->members)
                    if (m)
                        end_comm(m);
                return;
            }
        }
        // peers that wait inside (or for) an RCCL collective are released by aborting the communicators
        for (papr_exchange *m : todo) {
            end_comm(m);
            m->pins.fetch_sub(1);
        }
        return;
    }
    end_comm(x);  // (one process per GPU: the peers' collectives fail instead of waiting)
}

void papr_exchange_close(papr_exchange *x)
{
    if (!x)
        return;
    if (x->watchdog.joinable()) {
        x->watchdog_stop.store(true);
        x->watchdog.join();
    }
    if (x->hub) {
        bool last;
        {
            std::lock_guard<std::mutex> g(x->hub->m);
            for (papr_exchange *&m : x->hub->members)
                if (m == x)
                    m = nullptr;
            last = --x->hub->refs == 0;
        }
        if (last)
            delete x->hub;
        x->hub = nullptr;
    }
    while (x->pins.load() > 0)  // (a papr_exchange_abort that pinned this handle is still ending its communicator)
        usleep(200);
    if (x->bind) {  // set-up threads this handle never adopted: a communicator that still arrives is theirs to end
        std::lock_guard<std::mutex> g(x->bind->m);
        x->bind->abandoned = true;
    }
    if (x->device >= 0)  // (the context itself may be gone: bin/papr with PAPR_TEARDOWN=1 closed it first — found by ThreadSanitizer, round 6)
        (void)hipSetDevice(x->device);
    {
        std::lock_guard<std::timed_mutex> g(x->comm_m);
        if (x->comm && rccl() && !x->comm_ended)
            (void)rccl()->CommDestroy(x->comm);
        x->comm = nullptr;  // (only here: the owner's close)
    }
    if (x->d_send) (void)hipFree(x->d_send);
    if (x->d_recv) (void)hipFree(x->d_recv);
    if (x->h_send) (void)hipHostFree(x->h_send);
    if (x->h_recv) (void)hipHostFree(x->h_recv);
    delete x;
}

// ---- self-test ---------------------------------------------------------------------------------------------------
// Every collective the sharded step uses, once, on tiny buffers with contents every rank can predict for every other
// rank — so that a first run on N GPUs that goes wrong says WHICH collective did, instead of hanging in the step.
namespace {
inline unsigned char st_byte(int r, size_t i) { return (unsigned char)(0xA5u ^ (unsigned)(r * 37 + (int)i * 11 + 1)); }
inline uint64_t st_word(int r, size_t k) { return (uint64_t)(r + 1) * 1000003ull + (uint64_t)k * k + 7; }
inline size_t st_vsize(int r) { return 64 + 32 * (size_t)(r % 5); }  // (unequal on purpose)
}  // namespace

int papr_exchange_selftest(papr_exchange *x, papr_hip_ctx *ctx, int verbose)
{
    if (!x)
        return PAPR_E_ARG;
    const int world = x->world, me = x->rank;
    const bool say = verbose && me == 0;
    const char *transport = x->comm ? "RCCL" : x->hub ? "threads" : "caller's collectives";
    auto bad = [&](const char *what, const char *detail) {
        xfail(x, PAPR_E_STATE, "exchange self-test: %s FAILED on rank %d of %d (%s): %s", what, me, world, transport, detail);
        fprintf(stderr, "papr %s\n", x->err);
        char keep[256];
        memcpy(keep, x->err, sizeof(keep));
        papr_exchange_abort(x);  // the peers are not left waiting in the next collective
        memcpy(x->err, keep, sizeof(keep));
        return PAPR_E_STATE;
    };
    auto ok = [&](const char *what, double us) {
        if (say)
            fprintf(stderr, "papr exchange self-test: %-44s ok  %8.1f us  (%d rank%s, %s)\n", what, us, world, world == 1 ? "" : "s", transport);
    };
    constexpr size_t kRec = 96, kWords = 301;
    std::vector<unsigned char> send, recv;
    std::vector<uint64_t> words(kWords), want(kWords);
    try {
        send.resize(256);
        recv.resize(256 * (size_t)world + 256);
    } catch (...) {
        return xfail(x, PAPR_E_NOMEM, "out of host memory");
    }
    for (size_t k = 0; k < kWords; k++) {
        want[k] = 0;
        for (int r = 0; r < world; r++)
            want[k] += st_word(r, k);
    }
    // 1. host-level all-gather (the agreement, programs that outgrew their slot)
    for (size_t i = 0; i < kRec; i++)
        send[i] = st_byte(me, i);
    double t0 = now_us();
    int rc = allgather_bytes(x, send.data(), recv.data(), kRec);
    if (rc)
        return bad("host-level all-gather (96 B per rank)", x->err);
    for (int r = 0; r < world; r++)
        for (size_t i = 0; i < kRec; i++)
            if (recv[(size_t)r * kRec + i] != st_byte(r, i))
                return bad("host-level all-gather (96 B per rank)", "a rank's record arrived with other bytes than it sent");
    ok("host-level all-gather (96 B per rank)", now_us() - t0);
    // 2. host-level all-reduce
    for (size_t k = 0; k < kWords; k++)
        words[k] = st_word(me, k);
    t0 = now_us();
    rc = allreduce_u64(x, words.data(), kWords);
    if (rc)
        return bad("host-level all-reduce (301 x u64)", x->err);
    if (!(world == 1 && (x->use_ops || !x->comm)) && words != want)
        return bad("host-level all-reduce (301 x u64)", "the sums are not the sums of what the ranks sent");
    ok("host-level all-reduce (301 x u64)", now_us() - t0);
    if (!ctx || !xch_in_stream(x, ctx)) {
        x->selftest_done = true;
        return PAPR_OK;
    }
    // 3.-5. the in-stream collectives on device buffers (the single-wait step's)
    size_t vtotal = 0;
    std::vector<size_t> sizes((size_t)world), offs((size_t)world + 1);
    for (int r = 0; r < world; r++) {
        sizes[(size_t)r] = st_vsize(r);
        offs[(size_t)r] = vtotal;
        vtotal += (st_vsize(r) + 15) & ~(size_t)15;
    }
    offs[(size_t)world] = vtotal;
    const size_t need = std::max<size_t>({kRec * (size_t)world, vtotal, kWords * 8}) + 256;
    unsigned char *d_s = nullptr, *d_r = nullptr;
    XHIP(x, hipSetDevice(ctx->device));
    XHIP(x, hipMalloc((void **)&d_s, 4096));
    if (hipMalloc((void **)&d_r, need) != hipSuccess) {
        (void)hipFree(d_s);
        return xfail(x, PAPR_E_NOMEM, "cannot allocate the self-test's device buffers");
    }
    std::vector<unsigned char> back(need);
    auto run = [&](const char *what, const void *src, size_t src_bytes, size_t back_bytes, auto &&queue, auto &&check) -> int {
        if (hipMemcpy(d_s, src, src_bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemset(d_r, 0, need) != hipSuccess)
            return bad(what, "hipMemcpy to the device failed");
        const double t = now_us();
        int qrc = queue();
        if (qrc)
            return bad(what, x->err);
        hipError_t e;
        {
            Waiting w(x, what);
            e = hipStreamSynchronize(ctx->stream);
        }
        const double us = now_us() - t;
        if (e != hipSuccess || x->aborted.load())
            return bad(what, e != hipSuccess ? hipGetErrorString(e) : (x->timed_out.load(std::memory_order_acquire) ? x->wd_err : kCancelled));
        if (hipMemcpy(back.data(), d_r, back_bytes, hipMemcpyDeviceToHost) != hipSuccess)
            return bad(what, "hipMemcpy from the device failed");
        if (!check())
            return bad(what, "the device buffer does not hold what the ranks sent");
        ok(what, us);
        return PAPR_OK;
    };
    rc = run("in-stream all-gather (96 B per rank)", send.data(), kRec, kRec * (size_t)world,
             [&] { return xch_allgather_dev(x, ctx, d_s, d_r, kRec); },
             [&] {
                 for (int r = 0; r < world; r++)
                     for (size_t i = 0; i < kRec; i++)
                         if (back[(size_t)r * kRec + i] != st_byte(r, i))
                             return false;
                 return true;
             });
    if (rc == PAPR_OK) {
        std::vector<unsigned char> mine(st_vsize(me));
        for (size_t i = 0; i < mine.size(); i++)
            mine[i] = st_byte(me + 100, i);
        rc = run("in-stream all-gather-v (64..192 B, unequal)", mine.data(), mine.size(), vtotal,
                 [&] { return xch_allgatherv_dev(x, ctx, d_s, d_r, sizes.data(), offs.data()); },
                 [&] {
                     for (int r = 0; r < world; r++)
                         for (size_t i = 0; i < sizes[(size_t)r]; i++)
                             if (back[offs[(size_t)r] + i] != st_byte(r + 100, i))
                                 return false;
                     return true;
                 });
    }
    if (rc == PAPR_OK) {
        for (size_t k = 0; k < kWords; k++)
            words[k] = st_word(me, k);
        rc = run("in-stream all-reduce (301 x u64)", words.data(), kWords * 8, kWords * 8,
                 [&] { return xch_allreduce_u64_dev(x, ctx, d_s, d_r, kWords); },
                 [&] { return memcmp(back.data(), want.data(), kWords * 8) == 0; });
    }
    (void)hipFree(d_s);
    (void)hipFree(d_r);
    if (rc == PAPR_OK)
        x->selftest_done = true;
    return rc;
}

// PAPR_XCH_SELFTEST=1: once per handle, in front of the first step that has peers (papr_hip_analyze)
int papr_exchange_selftest_once(papr_exchange *x, papr_hip_ctx *ctx)
{
    if (!x || x->selftest_done || !env_int("PAPR_XCH_SELFTEST", 0))
        return PAPR_OK;
    if (x->world == 1 && !x->comm)
        return PAPR_OK;
    return papr_exchange_selftest(x, ctx, 1);
}

int papr_exchange_stats(papr_exchange *x, const papr_stats *local, papr_stats *total, double *sum_before, papr_stats *all)
{
    if (!x || !local || !total)
        return PAPR_E_ARG;
    const double t0 = now_us();
    try {
        x->scratch.resize((size_t)x->world * sizeof(papr_stats));
    } catch (...) {
        return xfail(x, PAPR_E_NOMEM, "out of host memory");
    }
    int rc = allgather_bytes(x, local, x->scratch.data(), sizeof(papr_stats));
    if (rc)
        return rc;
    const papr_stats *recs = reinterpret_cast<const papr_stats *>(x->scratch.data());
    papr_stats acc;
    papr_stats_init(&acc);
    double before = 0.0;
    for (int r = 0; r < x->world; r++) {
        if (r == x->rank)
            before = acc.sum;
        papr_stats_merge(&acc, &recs[r]);
    }
    *total = acc;
    if (sum_before)
        *sum_before = before;
    if (all)
        memcpy(all, recs, (size_t)x->world * sizeof(papr_stats));
    x->timing.stats_calls++;
    x->timing.stats_us += now_us() - t0;
    return PAPR_OK;
}

int papr_exchange_counts(papr_exchange *x, uint64_t *counts, int n)
{
    if (!x || n < 0 || (n && !counts))
        return PAPR_E_ARG;
    const double t0 = now_us();
    int rc = allreduce_u64(x, counts, (size_t)n);
    if (rc)
        return rc;
    x->timing.counts_calls++;
    x->timing.counts_us += now_us() - t0;
    return PAPR_OK;
}

int papr_exchange_exact_sum(papr_exchange *x, const void *program, size_t bytes, double *sum)
{
    if (!x || !program || !sum)
        return PAPR_E_ARG;
    const double t0 = now_us();
    int rc = PAPR_OK;
    if (x->world == 1 && x->use_ops) {
        const void *progs[1] = {program};
        const size_t sizes[1] = {bytes};
        rc = papr_exact_chain(progs, sizes, 1, sum);
    } else {
        // sizes first, then the programs padded to the largest
        std::vector<uint64_t> sizes((size_t)x->world);
        const uint64_t mine = bytes;
        rc = allgather_bytes(x, &mine, sizes.data(), sizeof(uint64_t));
        if (rc)
            return rc;
        uint64_t cap = 8;
        for (uint64_t s : sizes)
            cap = std::max(cap, s);
        cap = (cap + 7) & ~7ull;
        std::vector<unsigned char> send, recv;
        try {
            send.assign((size_t)cap, 0);
            recv.resize((size_t)cap * (size_t)x->world);
        } catch (...) {
            return xfail(x, PAPR_E_NOMEM, "out of host memory");
        }
        memcpy(send.data(), program, bytes);
        rc = allgather_bytes(x, send.data(), recv.data(), (size_t)cap);
        if (rc)
            return rc;
        std::vector<const void *> progs((size_t)x->world);
        std::vector<size_t> lens((size_t)x->world);
        for (int r = 0; r < x->world; r++) {
            progs[(size_t)r] = recv.data() + (size_t)r * (size_t)cap;
            lens[(size_t)r] = (size_t)sizes[(size_t)r];
        }
        rc = papr_exact_chain(progs.data(), lens.data(), x->world, sum);
    }
    if (rc)
        return xfail(x, rc, "papr_exact_chain failed over the gathered programs (code %d)", rc);
    x->timing.exact_calls++;
    x->timing.exact_us += now_us() - t0;
    return PAPR_OK;
}

int papr_exchange_get_timing(papr_exchange *x, papr_exchange_timing *out, int reset)
{
    if (!x || !out)
        return PAPR_E_ARG;
    *out = x->timing;
    if (reset)
        memset(&x->timing, 0, sizeof(x->timing));
    return PAPR_OK;
}

}  // extern "C"
