// ts_runtime.cpp — host runtime behind include/ts_hip.h: the stream's residency in HBM and the scan loop that
// alternates GPU launches over stretches of regular packets (ts_kernels.hip) with the closed-form host walker
// (ts_host.c) across the irregular ones.  No CPU compute path for the bulk: the walker sees a few hundred bytes per
// irregular packet, copied back from the device.

#include "ts_hip.h"
#include "papr_hip.h"  // the PAPR_E_* codes (one error vocabulary for the library)
#include "ts_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

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
    ts_wg_entry *d_lists = nullptr;
    uint32_t *d_list_counts = nullptr, *d_span_stopped = nullptr, *d_count = nullptr;
    unsigned long long *d_span_done = nullptr, *d_first = nullptr, *d_last = nullptr, *d_taken = nullptr;
    uint32_t *d_events = nullptr, *d_event_counts = nullptr, *d_merged_events = nullptr;
    uint32_t *h_events = nullptr;                           // pinned
    unsigned long long *h_taken = nullptr;                  // pinned: [0] units taken, [1] quirk events
    unsigned char *h_window = nullptr;                      // pinned: what the walker looks at
    void *h_tables = nullptr;                               // pinned: count / first / last read back at the end
    int spans = 0;
    int unroll = 1;  // TS_SCAN_UNROLL (measurement knob): packets per lane between two barriers of the scan kernel
    int block = 1024, agg = 0;  // TS_SCAN_BLOCK / TS_SCAN_AGG: threads per workgroup; per-(wave, PID) table updates
    hipEvent_t ev_a = nullptr, ev_m = nullptr, ev_b = nullptr;
};

namespace {

char g_ts_open_error[256] = "";
constexpr size_t kWindow = 1 << 16;       // bytes the walker gets per hand-over
constexpr uint64_t kMaxUnitsPerLaunch = 1ull << 31;
constexpr uint32_t kEventCap = 1024;             // per workgroup: harmless read-boundary events (one packet in 4096 at most)
constexpr uint32_t kMergedEventCap = 1u << 20;

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
    ctx->spans = ctx->num_cus;  // one 1024-thread workgroup (96 KiB of LDS) per CU
    {   // measurement knobs: packets per lane between barriers, workgroup size, aggregated table update
        int u = ctx->unroll, b = ctx->block, a = ctx->agg;
        if (const char *e = getenv("TS_SCAN_UNROLL")) u = atoi(e);
        if (const char *e = getenv("TS_SCAN_BLOCK")) b = atoi(e);
        if (const char *e = getenv("TS_SCAN_AGG")) a = atoi(e) != 0;
        if (ts_scan_form_exists(u, b, a)) {
            ctx->unroll = u;
            ctx->block = b;
            ctx->agg = a;
        }
    }
    OPENCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    OPENCHK(hipMalloc((void **)&ctx->d_lists, (size_t)ctx->spans * TS_PIDS * sizeof(ts_wg_entry)));
    OPENCHK(hipMalloc((void **)&ctx->d_list_counts, ctx->spans * sizeof(uint32_t)));
    OPENCHK(hipMalloc((void **)&ctx->d_span_stopped, ctx->spans * sizeof(uint32_t)));
    OPENCHK(hipMalloc((void **)&ctx->d_span_done, ctx->spans * sizeof(unsigned long long)));
    OPENCHK(hipMalloc((void **)&ctx->d_count, TS_PIDS * (sizeof(uint32_t) + 2 * sizeof(unsigned long long))));
    ctx->d_first = reinterpret_cast<unsigned long long *>(ctx->d_count + TS_PIDS);
    ctx->d_last = ctx->d_first + TS_PIDS;
    OPENCHK(hipMalloc((void **)&ctx->d_taken, 2 * sizeof(unsigned long long)));
    OPENCHK(hipHostMalloc((void **)&ctx->h_taken, 2 * sizeof(unsigned long long), hipHostMallocDefault));
    OPENCHK(hipMalloc((void **)&ctx->d_events, (size_t)ctx->spans * kEventCap * sizeof(uint32_t)));
    OPENCHK(hipMalloc((void **)&ctx->d_event_counts, ctx->spans * sizeof(uint32_t)));
    OPENCHK(hipMalloc((void **)&ctx->d_merged_events, (size_t)kMergedEventCap * sizeof(uint32_t)));
    OPENCHK(hipHostMalloc((void **)&ctx->h_events, (size_t)kMergedEventCap * sizeof(uint32_t), hipHostMallocDefault));
    OPENCHK(hipHostMalloc((void **)&ctx->h_window, kWindow, hipHostMallocDefault));
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
    release(ctx);
    if (ctx->d_lists) (void)hipFree(ctx->d_lists);
    if (ctx->d_list_counts) (void)hipFree(ctx->d_list_counts);
    if (ctx->d_span_stopped) (void)hipFree(ctx->d_span_stopped);
    if (ctx->d_span_done) (void)hipFree(ctx->d_span_done);
    if (ctx->d_count) (void)hipFree(ctx->d_count);
    if (ctx->d_taken) (void)hipFree(ctx->d_taken);
    if (ctx->d_events) (void)hipFree(ctx->d_events);
    if (ctx->d_event_counts) (void)hipFree(ctx->d_event_counts);
    if (ctx->d_merged_events) (void)hipFree(ctx->d_merged_events);
    if (ctx->h_events) (void)hipHostFree(ctx->h_events);
    if (ctx->h_taken) (void)hipHostFree(ctx->h_taken);
    if (ctx->h_window) (void)hipHostFree(ctx->h_window);
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

int ts_hip_scan(ts_hip_ctx *ctx, int hdmv, ts_scan_result *out)
{
    if (!ctx || !out)
        return PAPR_E_ARG;
    if (!ctx->loaded)
        return ts_fail(ctx, PAPR_E_STATE, "ts_hip_scan called before a stream was loaded");
    TSCHK(ctx, hipSetDevice(ctx->device));
    memset(out, 0, sizeof(*out));
    out->bytes = ctx->n;
    const uint32_t stride = hdmv ? 192u : 188u, sync_offset = hdmv ? 4u : 0u;
    TSCHK(ctx, hipMemsetAsync(ctx->d_count, 0, TS_PIDS * (sizeof(uint32_t) + 2 * sizeof(unsigned long long)), ctx->stream));
    TSCHK(ctx, hipMemsetAsync(ctx->d_first, 0xFF, TS_PIDS * sizeof(unsigned long long), ctx->stream));  // min table
    ts_walk_state st;
    ts_walk_init(&st, hdmv);
    float ms_total = 0.f, ms_merge = 0.f;
    for (;;) {
        // ---- GPU: every regular unit from a clean position on ----
        if (ts_walk_is_clean(&st) && st.pos + sync_offset + 188 <= ctx->n) {
            const uint64_t units = std::min<uint64_t>((ctx->n - st.pos + stride - 1) / stride, kMaxUnitsPerLaunch);
            ts_scan_params p{};
            p.data = ctx->d_data;
            p.nbytes = ctx->n;
            p.first_unit = st.pos;
            p.nunits = units;
            p.stride = stride;
            p.sync_offset = sync_offset;
            p.lists = ctx->d_lists;
            p.list_counts = ctx->d_list_counts;
            p.span_done = ctx->d_span_done;
            p.span_stopped = ctx->d_span_stopped;
            p.events = ctx->d_events;
            p.event_counts = ctx->d_event_counts;
            p.event_cap = kEventCap;
            p.merged_events = ctx->d_merged_events;
            p.merged_event_cap = kMergedEventCap;
            const int blocks = (int)std::min<uint64_t>((uint64_t)ctx->spans, (units + 1023) / 1024);
            TSCHK(ctx, hipEventRecord(ctx->ev_a, ctx->stream));
            ts_launch_scan(ctx->stream, blocks, ctx->unroll, ctx->block, ctx->agg, p);
            TSCHK(ctx, hipEventRecord(ctx->ev_m, ctx->stream));
            ts_launch_merge(ctx->stream, p, (uint32_t)blocks, out->packets, ctx->d_count, ctx->d_first, ctx->d_last, ctx->d_taken);
            TSCHK(ctx, hipEventRecord(ctx->ev_b, ctx->stream));
            TSCHK(ctx, hipGetLastError());
            TSCHK(ctx, hipMemcpyAsync(ctx->h_taken, ctx->d_taken, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
            TSCHK(ctx, hipStreamSynchronize(ctx->stream));
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_m) == hipSuccess)
                ms_total += ms;
            if (hipEventElapsedTime(&ms, ctx->ev_m, ctx->ev_b) == hipSuccess)
                ms_merge += ms;
            const uint64_t taken = ctx->h_taken[0];
            uint64_t nev = ctx->h_taken[1];
            if (nev > kMergedEventCap)
                return ts_fail(ctx, PAPR_E_LIMIT, "more than %u read-boundary events in one launch", kMergedEventCap);
            if (nev) {
                // packets that ended one byte past a 16384-byte read of the reference: each is one `skipped 1 bytes`
                // line, reported when the stream locks again, i.e. with that packet's own number
                TSCHK(ctx, hipMemcpyAsync(ctx->h_events, ctx->d_merged_events, nev * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
                TSCHK(ctx, hipStreamSynchronize(ctx->stream));
                std::sort(ctx->h_events, ctx->h_events + nev);
                for (uint64_t k = 0; k < nev; k++) {
                    if (ctx->h_events[k] >= taken)
                        continue;  // (cannot happen: events come from the spans in front of the stop)
                    if (out->nsync_errors < TS_MAX_SYNC_ERRORS) {
                        out->sync_errors[out->nsync_errors].skipped = 1;
                        out->sync_errors[out->nsync_errors].at_packet = out->packets + ctx->h_events[k] + 1;
                    }
                    out->nsync_errors++;
                }
            }
            out->launches++;
            out->packets += taken;
            out->gpu_packets += taken;
            st.pos += taken * stride;
            if (taken == units && st.pos >= ctx->n)
                break;  // the stream ended on a packet boundary
            if (taken == units)
                continue;  // (a launch-size limit: go on from here)
        }
        if (st.pos >= ctx->n)
            break;
        // ---- host walker: across the irregular packet(s), on a window copied back from the device ----
        const uint64_t want = std::min<uint64_t>(kWindow, ctx->n - st.pos);
        const int eof = st.pos + want >= ctx->n;
        TSCHK(ctx, hipMemcpyAsync(ctx->h_window, ctx->d_data + st.pos, want, hipMemcpyDeviceToHost, ctx->stream));
        TSCHK(ctx, hipStreamSynchronize(ctx->stream));
        const uint64_t before = st.pos;
        const uint64_t walked = ts_walk(&st, ctx->h_window, before, want, eof, 2, out);
        out->walks++;
        if (eof && (st.pos >= ctx->n || (walked == 0 && st.pos == before)))
            break;  // the walker consumed the tail
        if (!eof && walked == 0 && st.pos == before && want < 189)
            return ts_fail(ctx, PAPR_E_INTERNAL, "the packet walker made no progress at offset %llu", (unsigned long long)before);
    }
    // fold the device-side tables into the result (absolute packet numbers: min / max are order-independent)
    TSCHK(ctx, hipMemcpyAsync(ctx->h_tables, ctx->d_count, TS_PIDS * (sizeof(uint32_t) + 2 * sizeof(unsigned long long)),
                              hipMemcpyDeviceToHost, ctx->stream));
    TSCHK(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t *gc = (const uint32_t *)ctx->h_tables;
    const unsigned long long *gf = (const unsigned long long *)(gc + TS_PIDS), *gl = gf + TS_PIDS;
    for (int pid = 0; pid < TS_PIDS; pid++) {
        if (gf[pid] == ~0ull)
            continue;  // never seen by a launch
        out->count[pid] += gc[pid];
        if (out->first[pid] == 0 || gf[pid] < out->first[pid])
            out->first[pid] = gf[pid];
        if (gl[pid] > out->last[pid])
            out->last[pid] = gl[pid];
    }
    out->kernel_ms = ms_total;
    out->merge_ms = ms_merge;
    return PAPR_OK;
}

}  // extern "C"
