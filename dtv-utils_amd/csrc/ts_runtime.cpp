// ts_runtime.cpp — host runtime behind include/ts_hip.h: the stream's residency in HBM and the scan: ONE launch over
// all spans (regular packets in one-lane-per-packet blocks, everything else by the packet walker on the device,
// ts_kernels.hip), the chain check + merge, and — only where a span's speculated entry was not where the chain
// arrived — one more launch of that span from the true state.  The host never looks at stream bytes.

#include "ts_hip.h"
#include "papr_hip.h"  // the PAPR_E_* codes (one error vocabulary for the library)
#include "ts_kernels.h"
#include "ts_synth.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "ts_line_pool.h"

struct ts_hip_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    char err[256] = "";
    int num_cus = 256;
    unsigned char *d_data = nullptr;
    bool owns = false;
    uint64_t cap = 0, n = 0;
    bool loaded = false;
    // scan work buffers
    ts_wg_entry *d_lists = nullptr;                         // per span: its PIDs
    ts_span_rec *d_recs = nullptr;                          // per span: where it started, where it ended, what it counted
    uint32_t *d_count = nullptr;                            // stream-wide tables: count | first | last
    unsigned long long *d_first = nullptr, *d_last = nullptr;
    unsigned long long *d_span_base = nullptr;              // per span: stream-wide number of its first packet
    unsigned long long *d_span_bridge_base = nullptr;       // ... and of the first packet of the bridge in front of it
    uint32_t *d_span_attempt = nullptr;                     // per span: the attempt whose record the chain took (0: none)
    ts_cc_entry *d_cc_lists = nullptr;                      // per span: its PIDs' first / last continuity counter
    ts_bridge_rec *d_bridges = nullptr;                     // per span: does the chain get there from the span in front (ts_bridge_kernel)
    ts_span_out *d_span_out = nullptr, *h_span_out = nullptr;  // per span: what the host needs of it (one D2H copy; pinned)
    ts_event *d_events = nullptr;                           // events (report lines) of the scan's launches
    uint32_t event_cap = 0;
    unsigned int *d_event_count = nullptr;
    ts_merge_out *h_out = nullptr, *h_out_dev = nullptr;    // mapped: what the merge kernel reports
    void *h_tables = nullptr;                               // pinned: count / first / last read back at the end
    std::vector<ts_sync_error> errors;                      // every sync error of the last scan, in stream order
    std::vector<ts_discontinuity> discs;                    // every discontinuity of the last scan, in stream order
    ts_event *h_events = nullptr;                           // pinned mirror of the event list (grows with it)
    size_t h_events_cap = 0;
    bool events_on_host = false;                            // the event list IS pinned host memory the kernels write through the link (TS_SCAN_EVENTS_HOST=1)
    std::vector<unsigned char> line_scratch;                // the host's working copy of the lines, reused from scan to scan
    ts_line_pool *pool = nullptr;                           // host threads for the lines of a damaged stream (lazily)
    int pool_threads = -1;                                  // TS_HOST_THREADS (default: 8, at most the cores; 1: none)
    int spans = 0;                                          // spans of a scan with the full tables (one workgroup per CU)
    int spans_slots = 0;                                    // ... of the slot form (two per CU)
    int form = 0;                                           // 0: full tables, given up for the slot form when the stream is damaged;
                                                            // 1: full tables only; 2: slot form first (TS_SCAN_FORM=auto|full|slots)
    bool start_slots = false;                               // form 0: the last scan was given up for the slot form — start there
    uint32_t slot_limit = 0;                                // (tests: TS_SCAN_SLOT_LIMIT)
    uint32_t overlap = 1;                                   // TS_SCAN_OVERLAP=0: a span the one in front reached into is scanned again (tests)
    uint32_t lookahead = 1;                                 // TS_SCAN_LOOKAHEAD=0: a damaged spot's trips to memory one at a time (tests, measurements)
    int bridges_mode = -1;                                  // TS_SCAN_BRIDGES: 1 ts_bridge_kernel in front of every merge, 0 of none (tests)
    hipEvent_t ev_a = nullptr, ev_m = nullptr, ev_b = nullptr;
};

namespace {

char g_ts_open_error[256] = "";
constexpr uint32_t kEventCapInitial = 1u << 16;  // events the first scan has room for (a scan that needs more is repeated)

// spans of the slot form per CU.  (Three fit once the kernel is held to 80 VGPRs and are no faster: 1.246 against 1.242 ms at
// --damage 3e-4; more spans are more span boundaries for damage to sit on, and a span the chain reaches behind its assumed
// entry is scanned again by ONE workgroup: 768 spans took three launches and 3.08 ms at 1e-3 where 512 take one and 1.65.)
int env_spans_per_cu()
{
    const char *e = getenv("TS_SCAN_SLOT_SPANS_PER_CU");
    return e ? std::max(1, std::min(atoi(e), 4)) : 2;
}

int env_int_ts(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

int ts_fail(ts_hip_ctx *ctx, int code, const char *fmt, ...)
{
    char *dst = ctx ? ctx->err : g_ts_open_error;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 256, fmt, ap);
    va_end(ap);
    return code;
}

#define TSCHK(ctx, call)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return ts_fail(ctx, PAPR_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_));     \
    } while (0)

void release(ts_hip_ctx *ctx)
{
    if (ctx->owns && ctx->d_data)
        (void)hipFree(ctx->d_data);
    ctx->d_data = nullptr;
    ctx->owns = false;
    ctx->cap = ctx->n = 0;
    ctx->loaded = false;
}

int ensure_capacity(ts_hip_ctx *ctx, uint64_t nbytes)
{
    if (ctx->owns && ctx->d_data && ctx->cap >= nbytes)
        return PAPR_OK;
    release(ctx);
    const size_t bytes = (size_t)nbytes + 256;
    const hipError_t e = hipMalloc((void **)&ctx->d_data, bytes);
    if (e != hipSuccess) {
        ctx->d_data = nullptr;
        return ts_fail(ctx, PAPR_E_NOMEM, "hipMalloc(%zu bytes) for the stream failed: %s", bytes, hipGetErrorString(e));
    }
    ctx->owns = true;
    ctx->cap = nbytes;
    return PAPR_OK;
}

}  // namespace

extern "C" {

const char *ts_hip_last_error(const ts_hip_ctx *ctx)
{
    return ctx ? ctx->err : g_ts_open_error;
}

// the event list: device memory + a pinned mirror that grows with what a scan needs, or (events_on_host) ONE pinned, mapped
// buffer that is both — d_events is then its device-side address and no copy is ever made
static void free_events(ts_hip_ctx *ctx)
{
    if (ctx->events_on_host) {  // (d_events is the device-side address of the same buffer)
        if (ctx->h_events) (void)hipHostFree(ctx->h_events);
    } else {
        if (ctx->d_events) (void)hipFree(ctx->d_events);
        if (ctx->h_events) (void)hipHostFree(ctx->h_events);
    }
    ctx->d_events = ctx->h_events = nullptr;
    ctx->event_cap = 0;
    ctx->h_events_cap = 0;
}
static hipError_t alloc_events(ts_hip_ctx *ctx, size_t cap)
{
    free_events(ctx);
    hipError_t e;
    if (ctx->events_on_host) {
        e = hipHostMalloc((void **)&ctx->h_events, cap * sizeof(ts_event), hipHostMallocMapped);
        if (e == hipSuccess)
            e = hipHostGetDevicePointer((void **)&ctx->d_events, ctx->h_events, 0);
        if (e == hipSuccess)
            ctx->h_events_cap = cap;
    } else {
        e = hipMalloc((void **)&ctx->d_events, cap * sizeof(ts_event));
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        free_events(ctx);
        return e;
    }
    ctx->event_cap = (uint32_t)std::min<size_t>(cap, 0xFFFFFFFFu);
    return hipSuccess;
}

int ts_hip_open(ts_hip_ctx **out, int device)
{
    if (!out)
        return PAPR_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return ts_fail(nullptr, PAPR_E_NO_DEVICE, "no HIP device available (the TS scan has no CPU fallback)");
    if (device < 0 || device >= n)
        return ts_fail(nullptr, PAPR_E_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    ts_hip_ctx *ctx = new (std::nothrow) ts_hip_ctx();
    if (!ctx)
        return PAPR_E_NOMEM;
    ctx->device = device;
    auto bail = [&](int code) {
        snprintf(g_ts_open_error, sizeof(g_ts_open_error), "%s", ctx->err);
        ts_hip_close(ctx);
        return code;
    };
#define OPENCHK(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            ts_fail(ctx, PAPR_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_));               \
            return bail(PAPR_E_HIP);                                                               \
        }                                                                                          \
    } while (0)
    OPENCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    OPENCHK(hipGetDeviceProperties(&prop, device));
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->spans = std::min(ctx->num_cus, TS_MAX_SPANS);            // one 1024-thread workgroup (96 KiB of LDS) per CU
    ctx->spans_slots = std::min(env_spans_per_cu() * ctx->num_cus, TS_MAX_SPANS);  // 512-thread workgroups with per-slot tables (46 KiB)
    if (const char *e = getenv("TS_SCAN_SPANS"))  // (tests: many small spans exercise the chain check on small streams)
        ctx->spans = ctx->spans_slots = std::max(1, std::min(atoi(e), TS_MAX_SPANS));
    if (const char *e = getenv("TS_SCAN_FORM"))
        ctx->form = !strcmp(e, "full") ? 1 : !strcmp(e, "slots") ? 2 : 0;
    ctx->bridges_mode = env_int_ts("TS_SCAN_BRIDGES", -1);
    ctx->lookahead = env_int_ts("TS_SCAN_LOOKAHEAD", 1) != 0 ? 1u : 0u;
    ctx->overlap = env_int_ts("TS_SCAN_OVERLAP", 1) != 0 ? 1u : 0u;
    if (const char *e = getenv("TS_SCAN_SLOT_LIMIT"))
        ctx->slot_limit = (uint32_t)std::max(0, atoi(e));
    OPENCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    OPENCHK(hipMalloc((void **)&ctx->d_lists, (size_t)TS_MAX_SPANS * TS_PIDS * sizeof(ts_wg_entry)));
    OPENCHK(hipMalloc((void **)&ctx->d_recs, TS_MAX_SPANS * sizeof(ts_span_rec)));
    OPENCHK(hipMalloc((void **)&ctx->d_cc_lists, (size_t)TS_MAX_SPANS * TS_PIDS * sizeof(ts_cc_entry)));
    OPENCHK(hipMalloc((void **)&ctx->d_span_out, TS_MAX_SPANS * sizeof(ts_span_out)));
    OPENCHK(hipMalloc((void **)&ctx->d_bridges, TS_MAX_SPANS * sizeof(ts_bridge_rec)));
    OPENCHK(hipHostMalloc((void **)&ctx->h_span_out, TS_MAX_SPANS * sizeof(ts_span_out), hipHostMallocDefault));
    OPENCHK(hipMalloc((void **)&ctx->d_span_base, TS_MAX_SPANS * sizeof(unsigned long long)));
    OPENCHK(hipMalloc((void **)&ctx->d_span_bridge_base, TS_MAX_SPANS * sizeof(unsigned long long)));
    OPENCHK(hipMalloc((void **)&ctx->d_span_attempt, TS_MAX_SPANS * sizeof(uint32_t)));
    OPENCHK(hipMalloc((void **)&ctx->d_count, TS_PIDS * (sizeof(uint32_t) + 2 * sizeof(unsigned long long))));
    ctx->d_first = reinterpret_cast<unsigned long long *>(ctx->d_count + TS_PIDS);
    ctx->d_last = ctx->d_first + TS_PIDS;
    OPENCHK(hipMalloc((void **)&ctx->d_event_count, 4 * sizeof(unsigned int)));  // [0] events wanted, [1] a span overflowed its PID slots, [2] damaged: the slot form's
    // The report's lines are only ever WRITTEN by the kernels (32 bytes a line, a few MB for a badly damaged stream): with
    // TS_SCAN_EVENTS_HOST=1 the list is pinned host memory and the lines cross the link as posted writes while the scan
    // runs, instead of as a copy behind it (0.09 ms for 129 000 lines) — see alloc_events
    ctx->events_on_host = env_int_ts("TS_SCAN_EVENTS_HOST", 1) != 0;
    OPENCHK(alloc_events(ctx, kEventCapInitial));
    OPENCHK(hipHostMalloc((void **)&ctx->h_out, sizeof(ts_merge_out), hipHostMallocMapped));
    OPENCHK(hipHostGetDevicePointer((void **)&ctx->h_out_dev, ctx->h_out, 0));
    OPENCHK(hipHostMalloc(&ctx->h_tables, TS_PIDS * (sizeof(uint32_t) + 2 * sizeof(unsigned long long)), hipHostMallocDefault));
    OPENCHK(hipEventCreate(&ctx->ev_a));
    OPENCHK(hipEventCreate(&ctx->ev_b));
    OPENCHK(hipEventCreate(&ctx->ev_m));
#undef OPENCHK
    ts_kernels_prepare_device();
    *out = ctx;
    return PAPR_OK;
}

void ts_hip_close(ts_hip_ctx *ctx)
{
    if (!ctx)
        return;
    if (ctx->device >= 0)
        (void)hipSetDevice(ctx->device);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    delete ctx->pool;
    ctx->pool = nullptr;
    release(ctx);
    if (ctx->d_lists) (void)hipFree(ctx->d_lists);
    if (ctx->d_recs) (void)hipFree(ctx->d_recs);
    if (ctx->d_cc_lists) (void)hipFree(ctx->d_cc_lists);
    if (ctx->d_span_out) (void)hipFree(ctx->d_span_out);
    if (ctx->d_bridges) (void)hipFree(ctx->d_bridges);
    if (ctx->h_span_out) (void)hipHostFree(ctx->h_span_out);
    free_events(ctx);
    if (ctx->d_span_base) (void)hipFree(ctx->d_span_base);
    if (ctx->d_span_bridge_base) (void)hipFree(ctx->d_span_bridge_base);
    if (ctx->d_span_attempt) (void)hipFree(ctx->d_span_attempt);
    if (ctx->d_count) (void)hipFree(ctx->d_count);
    if (ctx->d_event_count) (void)hipFree(ctx->d_event_count);
    if (ctx->h_out) (void)hipHostFree(ctx->h_out);
    if (ctx->h_tables) (void)hipHostFree(ctx->h_tables);
    if (ctx->ev_a) (void)hipEventDestroy(ctx->ev_a);
    if (ctx->ev_b) (void)hipEventDestroy(ctx->ev_b);
    if (ctx->ev_m) (void)hipEventDestroy(ctx->ev_m);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int ts_hip_upload(ts_hip_ctx *ctx, const void *bytes, uint64_t nbytes)
{
    if (!ctx || (!bytes && nbytes))
        return PAPR_E_ARG;
    TSCHK(ctx, hipSetDevice(ctx->device));
    int rc = ensure_capacity(ctx, nbytes);
    if (rc)
        return rc;
    if (nbytes)
        TSCHK(ctx, hipMemcpy(ctx->d_data, bytes, nbytes, hipMemcpyHostToDevice));
    ctx->n = nbytes;
    ctx->loaded = true;
    return PAPR_OK;
}

int ts_hip_adopt(ts_hip_ctx *ctx, void *device_bytes, uint64_t nbytes)
{
    if (!ctx || (!device_bytes && nbytes))
        return PAPR_E_ARG;
    if (((uintptr_t)device_bytes & 3u) != 0)
        return ts_fail(ctx, PAPR_E_ARG, "adopted device memory must be 4-byte aligned");
    TSCHK(ctx, hipSetDevice(ctx->device));
    release(ctx);
    ctx->d_data = (unsigned char *)device_bytes;
    ctx->cap = ctx->n = nbytes;
    ctx->loaded = true;
    return PAPR_OK;
}

// the file -> pinned host -> HBM copy, double-buffered (replaces the fread loop of xport.c:241-244)
int ts_hip_load_file(ts_hip_ctx *ctx, const char *path)
{
    if (!ctx || !path)
        return PAPR_E_ARG;
    TSCHK(ctx, hipSetDevice(ctx->device));
    const int fd = open(path, O_RDONLY);
    if (fd < 0)
        return ts_fail(ctx, PAPR_E_IO, "cannot open %s", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {
        close(fd);
        return ts_fail(ctx, PAPR_E_IO, "cannot stat %s (or not a regular file)", path);
    }
    const uint64_t size = (uint64_t)sb.st_size;
    int rc = ensure_capacity(ctx, size);
    if (rc) {
        close(fd);
        return rc;
    }
    constexpr size_t kChunk = 16u << 20;
    void *stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    for (int b = 0; b < 2 && rc == PAPR_OK; b++) {
        if (hipHostMalloc(&stage[b], kChunk, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&done[b], hipEventDisableTiming) != hipSuccess)
            rc = ts_fail(ctx, PAPR_E_NOMEM, "cannot allocate the pinned staging buffers");
    }
    uint64_t off = 0;
    for (uint64_t c = 0; rc == PAPR_OK && off < size; c++) {
        const int b = (int)(c & 1);
        if (c >= 2 && hipEventSynchronize(done[b]) != hipSuccess)
            rc = ts_fail(ctx, PAPR_E_HIP, "hipEventSynchronize failed while recycling a staging buffer");
        const uint64_t want = std::min<uint64_t>(kChunk, size - off);
        uint64_t got = 0;
        while (rc == PAPR_OK && got < want) {
            const ssize_t r = pread(fd, (char *)stage[b] + got, want - got, (off_t)(off + got));
            if (r <= 0)
                rc = ts_fail(ctx, PAPR_E_IO, "read error in %s", path);
            else
                got += (uint64_t)r;
        }
        if (rc == PAPR_OK && (hipMemcpyAsync(ctx->d_data + off, stage[b], want, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                              hipEventRecord(done[b], ctx->stream) != hipSuccess))
            rc = ts_fail(ctx, PAPR_E_HIP, "hipMemcpyAsync of a stream chunk failed");
        off += want;
    }
    (void)hipStreamSynchronize(ctx->stream);
    for (int b = 0; b < 2; b++) {
        if (stage[b]) (void)hipHostFree(stage[b]);
        if (done[b]) (void)hipEventDestroy(done[b]);
    }
    close(fd);
    if (rc)
        return rc;
    ctx->n = size;
    ctx->loaded = true;
    return PAPR_OK;
}

int ts_hip_generate(ts_hip_ctx *ctx, uint64_t seed, uint64_t npackets, int hdmv)
{
    if (!ctx)
        return PAPR_E_ARG;
    TSCHK(ctx, hipSetDevice(ctx->device));
    const uint32_t unit = hdmv ? 192u : 188u;
    const uint64_t nbytes = npackets * unit;
    if (!(ctx->d_data && ctx->cap >= nbytes)) {
        int rc = ensure_capacity(ctx, nbytes);
        if (rc)
            return rc;
    }
    ts_launch_generate(ctx->stream, ctx->d_data, npackets, unit, seed, hdmv != 0);
    TSCHK(ctx, hipGetLastError());
    TSCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n = nbytes;
    ctx->loaded = true;
    return PAPR_OK;
}

int ts_hip_generate_damaged(ts_hip_ctx *ctx, uint64_t seed, uint64_t npackets, uint64_t period)
{
    if (!ctx)
        return PAPR_E_ARG;
    if (period == 0 || npackets % (4 * period) != 0)
        return ts_fail(ctx, PAPR_E_ARG, "npackets must be a multiple of 4 * period");
    TSCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t nbytes = ts_synth_damaged_size(npackets, period);
    // (the generator writes whole dwords — up to three bytes behind an nbytes that is no multiple of 4: only the context's
    // own buffers have that slack; an adopted one of exactly nbytes is replaced)
    if (!(ctx->d_data && ctx->owns && ctx->cap >= nbytes)) {
        int rc = ensure_capacity(ctx, nbytes);
        if (rc)
            return rc;
    }
    ts_launch_generate_damaged(ctx->stream, ctx->d_data, nbytes, period, seed);
    TSCHK(ctx, hipGetLastError());
    TSCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n = nbytes;
    ctx->loaded = true;
    return PAPR_OK;
}

int ts_hip_download(ts_hip_ctx *ctx, void *bytes, uint64_t first, uint64_t nbytes)
{
    if (!ctx || (!bytes && nbytes))
        return PAPR_E_ARG;
    if (!ctx->loaded || first + nbytes > ctx->n)
        return ts_fail(ctx, PAPR_E_ARG, "download range outside the stream");
    TSCHK(ctx, hipSetDevice(ctx->device));
    if (nbytes)
        TSCHK(ctx, hipMemcpy(bytes, ctx->d_data + first, nbytes, hipMemcpyDeviceToHost));
    return PAPR_OK;
}

// scan_with: nothing of the scan stands —
constexpr int kSlotsOverflowed = 1;  // a span met more PIDs than the slot form has slots
constexpr int kGaveUp = 2;           // the full-table form met a damaged stream (abort_walks)
constexpr uint32_t kAbortWalks = 8;  // (four until the rate went from one walk in 3072 packets to one in 6144: a spot that takes two walks made four of them in a span's first 24576 packets even at one spot in 10000)

// one scan in one form; what the forms tried before it cost goes into the result's `launches` and `kernel_ms`
static double host_now_ms()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + 1e-6 * (double)ts.tv_nsec;
}

static int scan_with(ts_hip_ctx *ctx, int hdmv, ts_scan_result *out, bool slots, uint32_t abort_walks, uint32_t launches_before,
                     double ms_before, double merge_ms_before)
{
    // TS_HOST_TRACE=1: where a scan's wall time goes, one stderr line per attempt (host clock, ms)
    static const bool trace = getenv("TS_HOST_TRACE") && atoi(getenv("TS_HOST_TRACE")) > 0;
    const double t_enter = trace ? host_now_ms() : 0.0;
    double t_synced = 0.0, t_events = 0.0, t_sorted = 0.0, t_linked = 0.0, t_counted = 0.0, t_scattered = 0.0;
    const double t_memset0 = t_enter;
    (void)t_memset0;
    memset(out, 0, sizeof(*out));
    out->bytes = ctx->n;
    // (the two lists keep their size from scan to scan — resize() below then value-initialises only what is new — and are
    // empty whenever a scan leaves early)
    struct ListsGuard {
        ts_hip_ctx *c;
        bool keep;
        ~ListsGuard()
        {
            if (!keep) {
                c->errors.clear();
                c->discs.clear();
            }
        }
    } lists{ctx, false};
    if (ctx->n == 0)
        return PAPR_OK;
    const uint32_t stride = hdmv ? 192u : 188u, sync_offset = hdmv ? 4u : 0u;
    // one span per CU, at least 64 KiB each (a span must hold a few packets for its entry to be found)
    const int spans_wanted = slots ? ctx->spans_slots : ctx->spans;
    uint64_t span_bytes = (ctx->n + (uint64_t)spans_wanted - 1) / (uint64_t)spans_wanted;
    const uint64_t min_span = (uint64_t)std::max(4096, atoi(getenv("TS_SCAN_MIN_SPAN") ? getenv("TS_SCAN_MIN_SPAN") : "65536"));
    span_bytes = std::max<uint64_t>((span_bytes + 4095) & ~4095ull, min_span);
    const uint32_t nspans = (uint32_t)((ctx->n + span_bytes - 1) / span_bytes);
    std::vector<ts_span_rec> recs(nspans);
    std::vector<unsigned long long> base(nspans), bridge_base(nspans);
    std::vector<uint32_t> taken(nspans);
    float ms_total = 0.f, ms_merge = 0.f;
    for (int round = 0;; round++) {  // (a second round only when the event list turned out too small)
        ts_launch_reset(ctx->stream, ctx->d_count, ctx->d_first, ctx->d_last, ctx->d_event_count, ctx->d_span_attempt, nspans);
        ts_scan_params p{};
        p.data = ctx->d_data;
        p.nbytes = ctx->n;
        p.span_bytes = span_bytes;
        p.first_span = 0;
        p.nspans_total = nspans;
        p.stride = stride;
        p.sync_offset = sync_offset;
        p.hdmv = hdmv ? 1u : 0u;
        p.attempt = 1;
        p.explicit_entry = 0;
        p.quirk_events = getenv("TS_SCAN_QUIRK_EVENTS") ? (uint32_t)atoi(getenv("TS_SCAN_QUIRK_EVENTS")) : 1u;
        p.slots = slots ? 1u : 0u;
        p.slot_limit = ctx->slot_limit;
        p.abort_walks = abort_walks;
        p.lookahead = ctx->lookahead;
        p.overlap = ctx->overlap;
        ts_walk_init(&p.entry, hdmv);
        p.lists = ctx->d_lists;
        p.recs = ctx->d_recs;
        p.cc_lists = ctx->d_cc_lists;
        p.events = ctx->d_events;
        p.event_cap = ctx->event_cap;
        p.event_count = ctx->d_event_count;
        // the bridges between the spans, all at once in front of the merge: for the slot form — the damaged stream's, where
        // most spans begin behind damage their speculated entry skipped — and for every launch that scans a span again; the
        // full-table form's first launch (a stream in order: nothing to bridge) does without the extra launch
        p.bridges = nullptr;
        ts_walk_state cur;
        ts_walk_init(&cur, hdmv);
        uint64_t packets = 0;
        uint32_t from = 0;
        uint32_t ahead = 0;  // events already on the host when the last launch's wait returned (events_on_host: all of them)
        out->launches = launches_before;
        for (;;) {
            // ---- scan (every span from its speculated entry; or ONE span again, from the state the chain arrived with) ...
            TSCHK(ctx, hipEventRecord(ctx->ev_a, ctx->stream));
            ts_launch_scan(ctx->stream, p.explicit_entry ? 1 : (int)nspans, p);
            TSCHK(ctx, hipEventRecord(ctx->ev_m, ctx->stream));
            p.bridges = (slots || p.explicit_entry || ctx->bridges_mode > 0) && ctx->bridges_mode != 0 ? ctx->d_bridges : nullptr;
            ts_launch_bridges(ctx->stream, p, from, cur);
            // ---- ... and the chain check + merge from there on
            ts_launch_merge(ctx->stream, p, from, packets, cur, ctx->d_count, ctx->d_first, ctx->d_last, ctx->d_span_base,
                            ctx->d_span_bridge_base, ctx->d_span_attempt, ctx->h_out_dev, ctx->d_span_out);
            TSCHK(ctx, hipEventRecord(ctx->ev_b, ctx->stream));
            TSCHK(ctx, hipGetLastError());
            // (the event counter behind the merge — its bridges write events too, in every workgroup: only now is it final)
            TSCHK(ctx, hipMemcpyAsync(&ctx->h_out->events, ctx->d_event_count, sizeof(unsigned int), hipMemcpyDeviceToHost,
                                      ctx->stream));
            // (the stream-wide tables as they stand behind this merge: final if the chain turns out complete — the usual
            // case — so that a scan is ONE wait)
            TSCHK(ctx, hipMemcpyAsync(ctx->h_tables, ctx->d_count, TS_PIDS * (sizeof(uint32_t) + 2 * sizeof(unsigned long long)),
                                      hipMemcpyDeviceToHost, ctx->stream));
            // (... and every span's numbering base and the head of its continuity list: the host links the spans)
            TSCHK(ctx, hipMemcpyAsync(ctx->h_span_out, ctx->d_span_out, (size_t)nspans * sizeof(ts_span_out), hipMemcpyDeviceToHost,
                                      ctx->stream));
            TSCHK(ctx, hipStreamSynchronize(ctx->stream));
            if (trace)
                t_synced = host_now_ms();
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_m) == hipSuccess)
                ms_total += ms;
            if (hipEventElapsedTime(&ms, ctx->ev_m, ctx->ev_b) == hipSuccess)
                ms_merge += ms;
            out->launches++;
            const ts_merge_out mo = *ctx->h_out;
            if ((slots && mo.pad2) || (abort_walks && mo.pad)) {
                out->kernel_ms = ms_before + ms_total;  // (what the dropped scan cost is part of the answer's cost)
                out->merge_ms = merge_ms_before + ms_merge;
                if (trace)
                    fprintf(stderr, "ts scan (%s form, given up): enter->synced %.3f ms (scan kernel %.3f, merge kernel %.3f)\n",
                            slots ? "slot" : "full-table", t_synced - t_enter, ms_total, ms_merge);
                return slots ? kSlotsOverflowed : kGaveUp;  // (garbage read as packets carries any PID: the full tables take it)
            }
            out->gpu_packets += mo.block_packets;
            out->walks += (uint32_t)std::min<uint64_t>(mo.walks, 0xFFFFFFFFull);
            packets = mo.packets;
            cur = mo.cur;
            if (mo.valid_upto >= nspans)
                break;
            if (out->launches - launches_before > 2 * nspans + 4)
                return ts_fail(ctx, PAPR_E_INTERNAL, "the span chain does not converge (span %u)", mo.valid_upto);
            // the chain arrived in front of span `valid_upto` somewhere else (or in another state) than the span assumed:
            // that span once more, from the true state
            if (trace) {  // (why: where the chain stands, and where the span thought it would)
                ts_span_rec r{};
                (void)hipMemcpy(&r, ctx->d_recs + mo.valid_upto, sizeof(r), hipMemcpyDeviceToHost);
                fprintf(stderr, "ts scan: span %u of %u again: the chain stands at %llu (skipped %llu, stale_af %u), the span [%llu, ...) entered at %lld = chain %+lld\n",
                        mo.valid_upto, nspans, (unsigned long long)cur.pos, (unsigned long long)cur.skipped, cur.stale_af,
                        (unsigned long long)((uint64_t)mo.valid_upto * span_bytes), (long long)r.entry, (long long)(r.entry - cur.pos));
            }
            from = mo.valid_upto;
            p.first_span = from;
            p.explicit_entry = 1;
            p.entry = cur;
            p.attempt++;
        }
        out->packets = packets;
        const unsigned int nev = ctx->h_out->events;  // (what the last merge saw: every scan launch of this round is behind it)
        if (nev > ctx->event_cap) {  // more sync errors than the list held: make room for all of them and scan again
            if (round > 0)
                return ts_fail(ctx, PAPR_E_INTERNAL, "the sync-error list overflowed twice (%u events)", nev);
            const size_t want = (size_t)nev + (size_t)nev / 4 + 1024;
            if (alloc_events(ctx, want) != hipSuccess)
                return ts_fail(ctx, PAPR_E_NOMEM, "cannot allocate the sync-error list (%zu events)", want);
            out->gpu_packets = 0;
            out->walks = 0;
            continue;
        }
        // ---- the report's lines: the events of the attempts the chain took, with stream-wide packet numbers, in the order
        // the reference prints them — and the continuity of every PID ACROSS the spans, which no span could check: a span
        // lists, per PID, the counter of its first and of its last payload-carrying packet; the spans are linked here, in
        // stream order, through the table the reference keeps (xport.c:2659), together with the packets the merge kernel's
        // bridges walked between them ----
        struct Line {
            uint64_t key;      // 2 * packet number (+ 1 for a sync error, which is printed behind that packet's own lines)
            uint64_t skipped;
            uint32_t kind, info;
        };
        if (ctx->events_on_host)
            ahead = nev;  // (they are here: the kernels wrote them through the link, the scan's wait saw them arrive)
        if (!ctx->events_on_host && nev > ctx->h_events_cap) {  // (pinned: a pageable destination made this copy the longest part of a damaged scan)
            if (ctx->h_events) (void)hipHostFree(ctx->h_events);
            ctx->h_events = nullptr;
            ctx->h_events_cap = 0;
            ahead = 0;  // (what crossed ahead went with the old buffer)
            const size_t want = std::max<size_t>((size_t)nev + nev / 4, 1u << 16);
            TSCHK(ctx, hipHostMalloc((void **)&ctx->h_events, want * sizeof(ts_event), hipHostMallocDefault));
            ctx->h_events_cap = want;
        }
        const bool fetch_more = nev > ahead;  // (the events the wait did not bring: all of them for a context's first scan)
        if (fetch_more)
            TSCHK(ctx, hipMemcpyAsync(ctx->h_events + ahead, ctx->d_events + ahead, (size_t)(nev - ahead) * sizeof(ts_event), hipMemcpyDeviceToHost,
                                      ctx->stream));
        // Few lines: this thread alone.  Many (a damaged stream): T threads, each a range of the event list, then each a
        // range of the spans; only the linking of the spans' continuity counters in between is serial (a few entries per
        // span).  The workers are woken HERE, while the lines cross the link.
        if (ctx->pool_threads < 0) {
            const char *e = getenv("TS_HOST_THREADS");
            const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
            ctx->pool_threads = std::max(1, std::min(e ? atoi(e) : 8, std::min(hw, 64)));
        }
        int burst_threads = 1;
        if (nev >= 16384 && ctx->pool_threads > 1) {
            if (!ctx->pool) {
                ctx->pool = new ts_line_pool();
                ctx->pool->start(ctx->pool_threads - 1);
            }
            if (!ctx->pool->workers.empty()) {
                ctx->pool->begin_burst();
                burst_threads = (int)ctx->pool->workers.size() + 1;
            }
        }
        if (fetch_more) {
            const hipError_t se = hipStreamSynchronize(ctx->stream);
            if (se != hipSuccess) {
                if (ctx->pool)
                    ctx->pool->end_burst();
                return ts_fail(ctx, PAPR_E_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(se));
            }
        }
        if (trace)
            t_events = host_now_ms();
        const ts_event *ev_begin = ctx->h_events;
        const ts_span_out *so = ctx->h_span_out;
        auto takes = [&](const ts_event &e) {  // (else: an attempt the chain did not take)
            if (!(e.span < nspans && so[e.span].attempt != 0 && so[e.span].attempt == (e.attempt & ~TS_EVENT_BRIDGE)))
                return false;
            // (the span's first `dup` packets were the span's in front already, which printed their lines: ts_overlap_packets)
            // (a discontinuity carries its packet's number within the span; a sync error the count in front of it — and no plain packet has one)
            return (e.attempt & TS_EVENT_BRIDGE) != 0 || e.kind != TS_EV_DISC || e.at_rel > so[e.span].dup;
        };
        // Every span's lines come in three runs, each in the reference's order already: what the merge kernel's bridge in
        // front of it reported, what the span itself reported (the event list's slots are handed out in stream order
        // within a span: ts_kernels.hip), and the few the linking below adds.  A counting sort over (span, run) — stable —
        // lays them out; the runs are then merged.  Nothing is sorted unless a run turns out not to be in order.
        if (trace)
            for (uint32_t k = 0; k < nspans; k++)
                if (so[k].dup)
                    fprintf(stderr, "ts scan: span %u starts %u packet(s) inside the span in front's last walk (its numbering from %llu)\n", k, so[k].dup,
                            (unsigned long long)so[k].base);
        auto run_of = [](const ts_event &e) { return (e.attempt & TS_EVENT_BRIDGE) ? (e.kind == TS_EV_BRIDGE_CC ? 0u : 1u) : 2u; };
        const int T = burst_threads;  // (the workers were woken when the scan learnt how many lines it has)
        auto parallel = [&](int n, const std::function<void(int)> &f) {
            if (T > 1)
                ctx->pool->run(n, f);
            else
                for (int k = 0; k < n; k++)
                    f(k);
        };
        struct BurstEnd {  // (whatever way this scan leaves: the workers go back to sleep)
            ts_line_pool *p;
            ~BurstEnd()
            {
                if (p)
                    p->end_burst();
            }
        } burst_end{T > 1 ? ctx->pool : nullptr};
        const size_t nb = 3 * (size_t)nspans;                       // buckets: (span, run)
        const size_t per_chunk = ((size_t)nev + T - 1) / (size_t)T;  // events per thread
        auto chunk = [&](int c, const ts_event *&b0, const ts_event *&e0) {
            b0 = ev_begin + std::min<size_t>((size_t)c * per_chunk, nev);
            e0 = ev_begin + std::min<size_t>((size_t)(c + 1) * per_chunk, nev);
        };
        // ---- a counting sort over (span, run), stable: per-thread counts, one prefix over (bucket, thread), the scatter ----
        std::vector<uint32_t> counts((size_t)T * nb, 0);
        parallel(T, [&](int c) {
            const ts_event *b0, *e0;
            chunk(c, b0, e0);
            uint32_t *h = counts.data() + (size_t)c * nb;
            for (const ts_event *pe = b0; pe != e0; pe++)
                if (takes(*pe))
                    h[3 * (size_t)pe->span + run_of(*pe)]++;
        });
        if (trace)
            t_counted = host_now_ms();
        std::vector<uint32_t> first(nb + 1, 0);
        {
            uint32_t running = 0;
            for (size_t k = 0; k < nb; k++) {
                first[k] = running;
                for (int c = 0; c < T; c++) {
                    const uint32_t h = counts[(size_t)c * nb + k];
                    counts[(size_t)c * nb + k] = running;  // (becomes: where thread c's first line of this bucket goes)
                    running += h;
                }
            }
            first[nb] = running;
        }
        const size_t nlines = first[nb];
        if (ctx->line_scratch.size() < nlines * sizeof(Line))
            ctx->line_scratch.resize(nlines * sizeof(Line) + (nlines * sizeof(Line)) / 4);
        Line *lines = reinterpret_cast<Line *>(ctx->line_scratch.data());
        parallel(T, [&](int c) {
            const ts_event *b0, *e0;
            chunk(c, b0, e0);
            uint32_t *at = counts.data() + (size_t)c * nb;
            for (const ts_event *pe = b0; pe != e0; pe++)
                if (takes(*pe)) {
                    const ts_event &e = *pe;
                    // (an event of a bridge carries TS_EVENT_BRIDGE and counts from the bridge's first packet)
                    const uint64_t num = ((e.attempt & TS_EVENT_BRIDGE) ? so[e.span].bridge_base : so[e.span].base) + e.at_rel;
                    lines[at[3 * (size_t)e.span + run_of(e)]++] = Line{2 * num + (e.kind == TS_EV_SYNC ? 1u : 0u), e.skipped, e.kind, e.info};
                }
        });
        if (trace)
            t_scattered = host_now_ms();
        auto by_key = [](const Line &a, const Line &b) { return a.key < b.key; };
        auto in_order = [&](Line *b, Line *e) {
            if (!std::is_sorted(b, e, by_key))
                std::sort(b, e, by_key);  // (keys are unique: a packet has one line of each kind at most)
        };
        // ---- every span's three runs in order, and how many lines of either kind it will print (by ranges of spans) ----
        const int span_jobs = T > 1 ? std::min<int>((int)nspans, 4 * T) : 1;
        auto span_range = [&](int j, uint32_t &k0, uint32_t &k1) {
            k0 = (uint32_t)((uint64_t)nspans * (uint64_t)j / (uint64_t)span_jobs);
            k1 = (uint32_t)((uint64_t)nspans * (uint64_t)(j + 1) / (uint64_t)span_jobs);
        };
        std::vector<uint32_t> nsync(nspans + 1, 0), ndisc(nspans + 1, 0);
        parallel(span_jobs, [&](int j) {
            uint32_t k0, k1;
            span_range(j, k0, k1);
            for (uint32_t k = k0; k < k1; k++) {
                if (so[k].attempt == 0)
                    continue;
                Line *r0 = lines + first[3 * (size_t)k], *r1 = lines + first[3 * (size_t)k + 1], *r2 = lines + first[3 * (size_t)k + 2],
                     *r3 = lines + first[3 * (size_t)k + 3];
                in_order(r0, r1);
                in_order(r1, r2);
                in_order(r2, r3);
                uint32_t ns = 0;
                for (const Line *l = r1; l != r3; l++)
                    ns += l->kind == TS_EV_SYNC ? 1u : 0u;
                nsync[k] = ns;
                ndisc[k] = (uint32_t)(r3 - r1) - ns;
            }
        });
        if (trace)
            t_sorted = host_now_ms();
        // ---- link the spans (serial): continuity_counter[] as the reference would hold it at each span's start ----
        std::vector<uint8_t> cc_state(TS_PIDS, 0);  // last counter + 1, 0 = none yet
        std::vector<Line> linked;                   // the lines the linking adds, span after span
        std::vector<uint32_t> linked_first(nspans + 1, 0);
        // (spans with more PIDs than travel with the scan's one wait — garbage read as packets carries any PID —: the heads
        // of ALL lists once more, as long as the longest, in one copy)
        std::vector<ts_cc_entry> big;
        uint32_t max_ncc = 0;
        for (uint32_t k = 0; k < nspans; k++)
            if (so[k].attempt)
                max_ncc = std::max(max_ncc, so[k].ncc);
        if (max_ncc > TS_CC_OUT) {
            big.resize((size_t)nspans * max_ncc);
            TSCHK(ctx, hipMemcpy2D(big.data(), (size_t)max_ncc * sizeof(ts_cc_entry), ctx->d_cc_lists, (size_t)TS_PIDS * sizeof(ts_cc_entry),
                                   (size_t)max_ncc * sizeof(ts_cc_entry), nspans, hipMemcpyDeviceToHost));
        }
        for (uint32_t k = 0; k < nspans; k++) {
            linked_first[k] = (uint32_t)linked.size();
            if (so[k].attempt == 0)
                continue;
            const Line *r0 = lines + first[3 * (size_t)k], *r1 = lines + first[3 * (size_t)k + 1];
            auto check = [&](uint32_t pid, uint32_t cc, uint64_t num) {
                const uint32_t last = cc_state[pid];
                if (last != 0 && pid != 0x1fffu && (last & 0xfu) != cc)
                    linked.push_back(Line{2 * num, 0, TS_EV_DISC, (pid << 8) | (cc << 4) | (last & 0xfu)});
            };
            for (const Line *l = r0; l != r1; l++) {  // the bridge's payload-carrying packets, in front of the span, one by one
                const uint32_t pid = l->info >> 8, cc = (l->info >> 4) & 0xfu;
                check(pid, cc, l->key / 2);
                cc_state[pid] = (uint8_t)(cc + 1u);
            }
            const ts_cc_entry *list = so[k].ncc > TS_CC_OUT ? big.data() + (size_t)k * max_ncc : so[k].cc;
            for (uint32_t j = 0; j < so[k].ncc; j++)
                if (list[j].first_rel >= so[k].dup)  // (else: the very packet the span in front left the counter at)
                    check(list[j].pid, list[j].first_cc, so[k].base + list[j].first_rel + 1);
            for (uint32_t j = 0; j < so[k].ncc; j++)
                cc_state[list[j].pid] = (uint8_t)(list[j].last_cc + 1u);
            in_order(linked.data() + linked_first[k], linked.data() + linked.size());
            ndisc[k] += (uint32_t)(linked.size() - linked_first[k]);
        }
        linked_first[nspans] = (uint32_t)linked.size();
        // where every span's lines go in the two lists (exclusive prefix sums)
        {
            uint32_t es = 0, ds = 0;
            for (uint32_t k = 0; k <= nspans; k++) {
                const uint32_t a = k < nspans ? nsync[k] : 0, d = k < nspans ? ndisc[k] : 0;
                nsync[k] = es;
                ndisc[k] = ds;
                es += a;
                ds += d;
            }
        }
        ctx->errors.resize(nsync[nspans]);
        ctx->discs.resize(ndisc[nspans]);
        if (trace)
            t_linked = host_now_ms();
        // ---- per span: the three-way merge of the bridge's lines, the span's lines and the linked ones, into place ----
        parallel(span_jobs, [&](int j) {
            uint32_t k0, k1;
            span_range(j, k0, k1);
            for (uint32_t k = k0; k < k1; k++) {
                if (so[k].attempt == 0)
                    continue;
                const Line *a = lines + first[3 * (size_t)k + 1], *r2 = lines + first[3 * (size_t)k + 2], *b = r2,
                           *r3 = lines + first[3 * (size_t)k + 3];
                const Line *c = linked.data() + linked_first[k], *ce = linked.data() + linked_first[k + 1];
                ts_sync_error *eo = ctx->errors.data() + nsync[k];
                ts_discontinuity *dout = ctx->discs.data() + ndisc[k];
                uint64_t errors_so_far = nsync[k];
                for (;;) {
                    const Line *pick = nullptr;
                    int which = -1;
                    if (a != r2) { pick = a; which = 0; }
                    if (b != r3 && (!pick || b->key < pick->key)) { pick = b; which = 1; }
                    if (c != ce && (!pick || c->key < pick->key)) { pick = c; which = 2; }
                    if (!pick)
                        break;
                    if (pick->kind == TS_EV_SYNC) {
                        *eo++ = ts_sync_error{pick->skipped, pick->key / 2};
                        errors_so_far++;
                    } else {
                        ts_discontinuity d{};
                        d.at_packet = pick->key / 2;
                        d.after_sync_errors = errors_so_far;
                        d.pid = pick->info >> 8;
                        d.received = (uint8_t)((pick->info >> 4) & 0xfu);
                        d.expected = (uint8_t)(pick->info & 0xfu);
                        *dout++ = d;
                    }
                    if (which == 0) a++; else if (which == 1) b++; else c++;
                }
            }
        });
        memcpy(out->cc_state, cc_state.data(), TS_PIDS);
        break;
    }
    lists.keep = true;
    out->nsync_errors = ctx->errors.size();
    for (size_t k = 0; k < ctx->errors.size() && k < TS_MAX_SYNC_ERRORS; k++)
        out->sync_errors[k] = ctx->errors[k];
    out->ndiscontinuities = ctx->discs.size();
    for (size_t k = 0; k < ctx->discs.size() && k < TS_MAX_DISCONTINUITIES; k++)
        out->discontinuities[k] = ctx->discs[k];
    // the stream-wide tables (absolute packet numbers: min / max are order-independent; copied behind the last merge)
    const uint32_t *gc = (const uint32_t *)ctx->h_tables;
    const unsigned long long *gf = (const unsigned long long *)(gc + TS_PIDS), *gl = gf + TS_PIDS;
    for (int pid = 0; pid < TS_PIDS; pid++) {
        if (gf[pid] == ~0ull)
            continue;  // never seen
        out->count[pid] = gc[pid];
        out->first[pid] = gf[pid];
        out->last[pid] = gl[pid];
    }
    out->kernel_ms = ms_before + ms_total;
    out->merge_ms = merge_ms_before + ms_merge;
    if (trace)
        fprintf(stderr, "ts scan (%s form): enter->synced %.3f ms (scan kernels %.3f, merge kernels %.3f, %u launches), events here +%.3f, "
                "lines counted +%.3f, scattered +%.3f, in order +%.3f, spans linked +%.3f, merged and copied out +%.3f; %zu lines\n", slots ? "slot" : "full-table",
                t_synced - t_enter, ms_total, ms_merge, out->launches - launches_before, t_events - t_synced, t_counted - t_events,
                t_scattered - t_counted, t_sorted - t_scattered,
                t_linked - t_sorted, host_now_ms() - t_linked, ctx->errors.size() + ctx->discs.size());
    return PAPR_OK;
}

static int ts_hip_scan_impl(ts_hip_ctx *ctx, int hdmv, ts_scan_result *out)
{
    // The full-table form (one 1024-thread workgroup per CU) is the faster one on a stream that is in order (by 4 %), the
    // slot form (two spans per CU) on a damaged one (one damaged packet in 10000: even; in 3000: by a quarter; in 1000: by
    // half): the scan starts in the first and, if a span meets damage more often than once in 6144 packets (with the look-ahead across damaged spots the two forms are even at about that rate) (four
    // walks in: the first twentieth of a span or less), is done again in the second — which itself hands a stream with more PIDs
    // in a span than it has slots back to the first.
    // A context remembers: after a stream that was given up for the slot form the next scan STARTS in the slot form (a capture's
    // next file is damaged like the last one; the attempt that is given up costs 0.1-0.4 ms), until a scan in the slot form
    // walks less than once in 12288 packets.
    int rc = PAPR_OK;
    out->launches = 0;
    out->kernel_ms = out->merge_ms = 0.0;
    if (ctx->form == 0 && !ctx->start_slots) {
        rc = scan_with(ctx, hdmv, out, false, kAbortWalks, 0, 0.0, 0.0);
        if (rc != kGaveUp)
            return rc;
        ctx->start_slots = true;
    }
    if (ctx->form != 1) {
        rc = scan_with(ctx, hdmv, out, true, 0, out->launches, out->kernel_ms, out->merge_ms);
        if (rc != kSlotsOverflowed) {
            if (rc == PAPR_OK && ctx->form == 0 && (uint64_t)out->walks * 12288u < out->packets)
                ctx->start_slots = false;  // (in order again: the full tables are the faster form)
            return rc;
        }
        ctx->start_slots = false;
    }
    return scan_with(ctx, hdmv, out, false, 0, out->launches, out->kernel_ms, out->merge_ms);
}

int ts_hip_scan(ts_hip_ctx *ctx, int hdmv, ts_scan_result *out)
{
    if (!ctx || !out)
        return PAPR_E_ARG;
    if (!ctx->loaded)
        return ts_fail(ctx, PAPR_E_STATE, "ts_hip_scan called before a stream was loaded");
    TSCHK(ctx, hipSetDevice(ctx->device));
    try {
        return ts_hip_scan_impl(ctx, hdmv, out);
    } catch (const std::bad_alloc &) {
        return ts_fail(ctx, PAPR_E_NOMEM, "out of host memory");
    }
}

uint64_t ts_hip_sync_error_count(const ts_hip_ctx *ctx)
{
    return ctx ? (uint64_t)ctx->errors.size() : 0;
}

int ts_host_pool_selftest(int threads, int rounds)
{
    return ts_line_pool_selftest(threads, rounds);
}

size_t ts_hip_result_size(void)
{
    return sizeof(ts_scan_result);
}

uint64_t ts_hip_discontinuity_count(const ts_hip_ctx *ctx)
{
    return ctx ? (uint64_t)ctx->discs.size() : 0;
}

int ts_hip_get_discontinuities(const ts_hip_ctx *ctx, uint64_t first, uint64_t n, ts_discontinuity *out)
{
    if (!ctx || (n && !out) || first > ctx->discs.size() || n > ctx->discs.size() - first)
        return PAPR_E_ARG;
    if (n)
        memcpy(out, ctx->discs.data() + first, (size_t)n * sizeof(ts_discontinuity));
    return PAPR_OK;
}

int ts_hip_get_sync_errors(const ts_hip_ctx *ctx, uint64_t first, uint64_t n, ts_sync_error *out)
{
    if (!ctx || (n && !out) || first > ctx->errors.size() || n > ctx->errors.size() - first)
        return PAPR_E_ARG;
    if (n)
        memcpy(out, ctx->errors.data() + first, (size_t)n * sizeof(ts_sync_error));
    return PAPR_OK;
}

}  // extern "C"
