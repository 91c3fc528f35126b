// papr_exchange.cpp — the exchange between the shards of a sharded papr run, behind the C ABI (include/papr_hip.h):
// all-gather of 96-byte pass-1 records + ordered merge, all-reduce of the per-level counters, all-gather of the
// exact-sum programs + chain.  Transport: RCCL (bound at run time with dlopen, so that a process that already
// carries an RCCL — PyTorch ships its own — uses that one instead of loading a second copy), or caller-supplied
// collectives (the CPU tests: gloo).  SURVEY.md 7.1 C1 / C2.

#include "papr_runtime_internal.h"

#include <atomic>
#include <dlfcn.h>
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
        } else {
            cv.wait(lk, [&] { return generation != gen || failed; });
        }
        return !failed;
    }
};

struct papr_exchange {
    int rank = 0, world = 1;
    LocalHub *hub = nullptr;
    char err[256] = "";
    // caller-supplied transport
    papr_exchange_ops ops{};
    bool use_ops = false;
    // RCCL transport
    papr_hip_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    bool want_rccl = false;        // papr_exchange_open_rccl_local: papr_exchange_bind still has to create `comm`
    ncclUniqueId solo_uid{};       // ... for a world of one (no hub)
    std::atomic<bool> aborted{false};
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
#define XNCCL(x, call)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return xfail(x, PAPR_E_HIP, "%s failed: %s", #call,                                 \
                         rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "RCCL error");   \
    } while (0)

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
        const char *gone = "another shard's thread gave up: the exchange is cancelled";
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
    XNCCL(x, rccl()->AllGather(x->d_send, x->d_recv, bytes_per_rank, ncclUint8, x->comm, ctx->stream));
    XHIP(x, hipMemcpyAsync(x->h_recv, x->d_recv, total, hipMemcpyDeviceToHost, ctx->stream));
    XHIP(x, hipStreamSynchronize(ctx->stream));
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
    XNCCL(x, rccl()->AllReduce(x->d_send, x->d_recv, count, ncclUint64, ncclSum, x->comm, ctx->stream));
    XHIP(x, hipMemcpyAsync(x->h_recv, x->d_recv, bytes, hipMemcpyDeviceToHost, ctx->stream));
    XHIP(x, hipStreamSynchronize(ctx->stream));
    memcpy(buf, x->h_recv, bytes);
    return PAPR_OK;
}

}  // namespace

extern "C" {

const char *papr_exchange_last_error(const papr_exchange *x)
{
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
    XNCCL(x, rccl()->GroupStart());
    for (int r = 0; r < world; r++) {
        unsigned char *dst = (unsigned char *)recv_dev + offs[r];
        const ncclResult_t rr = rccl()->Broadcast(r == me ? send_dev : (const void *)dst, dst, sizes[r], ncclUint8, r, x->comm, x->ctx->stream);
        if (rr != ncclSuccess) {
            (void)rccl()->GroupEnd();
            return xfail(x, PAPR_E_HIP, "ncclBroadcast failed: %s", rccl()->GetErrorString ? rccl()->GetErrorString(rr) : "RCCL error");
        }
    }
    XNCCL(x, rccl()->GroupEnd());
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
    if (hipSetDevice(ctx->device) != hipSuccess) {
        delete x;
        return xfail(nullptr, PAPR_E_HIP, "hipSetDevice(%d) failed", ctx->device);
    }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    const ncclResult_t r = rccl()->CommInitRank(&x->comm, world, uid, rank);
    if (r != ncclSuccess) {
        xfail(nullptr, PAPR_E_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world,
              rccl()->GetErrorString ? rccl()->GetErrorString(r) : "RCCL error");
        delete x;
        return PAPR_E_HIP;
    }
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
    if (x->hub) {
        std::vector<uint64_t> devs((size_t)x->world);
        const uint64_t mine = (uint64_t)ctx->device;
        int rc = allgather_bytes(x, &mine, devs.data(), sizeof(uint64_t));
        if (rc)
            return rc;
        std::sort(devs.begin(), devs.end());
        if (std::adjacent_find(devs.begin(), devs.end()) != devs.end())
            return PAPR_OK;
    }
    if (hipSetDevice(ctx->device) != hipSuccess)
        return xfail(x, PAPR_E_HIP, "hipSetDevice(%d) failed", ctx->device);
    const ncclUniqueId uid = x->hub ? x->hub->uid : x->solo_uid;
    const ncclResult_t r = rccl()->CommInitRank(&x->comm, x->world, uid, x->rank);  // (every thread is in here at once)
    if (r != ncclSuccess) {
        x->comm = nullptr;
        return xfail(x, PAPR_E_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", x->rank, x->world,
                     rccl()->GetErrorString ? rccl()->GetErrorString(r) : "RCCL error");
    }
    x->ctx = ctx;
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
    if (x->hub) {
        std::vector<papr_exchange *> members;
        {
            std::lock_guard<std::mutex> g(x->hub->m);
            x->hub->failed = true;
            x->hub->cv.notify_all();
            members = x->hub->members;
        }
        // peers that wait inside (or for) an RCCL collective are released by aborting the communicators
        for (papr_exchange *m : members)
            if (m && m->comm && rccl() && rccl()->CommAbort && !m->aborted.exchange(true))
                (void)rccl()->CommAbort(m->comm);
        return;
    }
    if (x->comm && rccl() && rccl()->CommAbort && !x->aborted.exchange(true))
        (void)rccl()->CommAbort(x->comm);  // (one process per GPU: the peers' collectives fail instead of waiting)
}

void papr_exchange_close(papr_exchange *x)
{
    if (!x)
        return;
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
    if (x->ctx)
        (void)hipSetDevice(x->ctx->device);
    if (x->comm && rccl() && !x->aborted.load())
        (void)rccl()->CommDestroy(x->comm);
    if (x->d_send) (void)hipFree(x->d_send);
    if (x->d_recv) (void)hipFree(x->d_recv);
    if (x->h_send) (void)hipHostFree(x->h_send);
    if (x->h_recv) (void)hipHostFree(x->h_recv);
    delete x;
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
