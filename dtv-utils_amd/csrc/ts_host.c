/*
 * ts_host.c — the GPU-free half of the transport-stream packet scan (include/ts_hip.h): the packet walker as a host
 * function and the report formatter.
 *
 * The walker's step — sync search + the packet it ends on, in closed form — lives in ts_walk_core.h and is compiled
 * twice: here, into ts_walk (exported; the tests run it against the oracle and the reference's recordings without a
 * GPU), and into the scan kernel, where one wave runs it across every packet that is not at its place in the regular
 * stride (ts_kernels.hip).  What the GPU executes across damage is therefore the logic these tests pin.
 */
#include <stdio.h>
#include <string.h>

#include "ts_hip.h"

void ts_walk_init(ts_walk_state *st, int hdmv)
{
    memset(st, 0, sizeof(*st));
    st->hdmv = hdmv != 0;
    st->extra_pending = st->hdmv ? 4u : 0u; /* xport.c:2662 */
}

int ts_walk_is_clean(const ts_walk_state *st)
{
    return st->skipped == 0 && st->stale_af == 0 && (!st->hdmv || st->extra_pending == 4u);
}

typedef struct host_walk {
    const unsigned char *data; /* file bytes [base, ...) */
    uint64_t base;
    ts_scan_result *res;
} host_walk;

static uint64_t host_find_sync(const host_walk *w, uint64_t from, uint64_t end)
{
    const unsigned char *hit = (const unsigned char *)memchr(w->data + (from - w->base), 0x47, (size_t)(end - from));
    return hit ? w->base + (uint64_t)(hit - w->data) : end;
}

static void host_count(const host_walk *w, unsigned h1, unsigned h2)
{
    ts_scan_result *res = w->res;
    res->packets++; /* xport.c:2860 */
    if ((h1 & 0x80u) == 0) { /* transport_error_indicator clear, xport.c:2861-2867 */
        const unsigned pid = ((h1 & 0x1fu) << 8) | h2;
        res->count[pid]++;
        if (res->first[pid] == 0)
            res->first[pid] = res->packets;
        res->last[pid] = res->packets;
    }
}

/* xport.c:2872-2889: the packet's counter against the PID's last one (ts_scan_result.cc_state: last counter + 1, 0 = none) */
static void host_cc(const host_walk *w, unsigned pid, unsigned h3)
{
    ts_scan_result *res = w->res;
    const unsigned cc = h3 & 0xfu, last = res->cc_state[pid];
    if ((h3 & 0x10u) == 0)
        return; /* no payload: neither checked nor remembered */
    if (last != 0 && pid != 0x1fffu && (last & 0xfu) != cc) { /* (last = counter + 1, so last & 15 is the successor) */
        if (res->ndiscontinuities < TS_MAX_DISCONTINUITIES) {
            ts_discontinuity *d = &res->discontinuities[res->ndiscontinuities];
            d->at_packet = res->packets;
            d->after_sync_errors = res->nsync_errors;
            d->pid = pid;
            d->received = (uint8_t)cc;
            d->expected = (uint8_t)(last & 0xfu);
            d->pad[0] = d->pad[1] = 0;
        }
        res->ndiscontinuities++;
    }
    if (pid != 0)
        res->cc_state[pid] = (uint8_t)(cc + 1u);
}

static void host_sync_error(const host_walk *w, uint64_t skipped)
{
    ts_scan_result *res = w->res;
    if (res->nsync_errors < TS_MAX_SYNC_ERRORS) { /* (the inline list holds the first ones; the count is complete) */
        res->sync_errors[res->nsync_errors].skipped = skipped;
        res->sync_errors[res->nsync_errors].at_packet = res->packets;
    }
    res->nsync_errors++;
}

#define TS_CORE_QUAL static
#define TS_CORE_NAME host_walk_step
#define TS_CORE_CTX const host_walk *
#define TS_CORE_BYTE(ctx, off) ((unsigned)(ctx)->data[(off) - (ctx)->base])
#define TS_CORE_FIND_SYNC(ctx, from, end) host_find_sync(ctx, from, end)
#define TS_CORE_COUNT(ctx, h1, h2) host_count(ctx, h1, h2)
#define TS_CORE_CC(ctx, pid, h3) host_cc(ctx, pid, h3)
#define TS_CORE_SYNC_ERROR(ctx, skipped) host_sync_error(ctx, skipped)
#include "ts_walk_core.h"

uint64_t ts_walk(ts_walk_state *st, const unsigned char *data, uint64_t base, uint64_t n, int eof, uint64_t min_packets,
                 ts_scan_result *res)
{
    const host_walk w = {data, base, res};
    const uint64_t end = base + n;
    uint64_t taken = 0;
    if (st->pos < base)
        return 0;
    for (;;) {
        if (taken >= min_packets && ts_walk_is_clean(st))
            return taken;
        if (!host_walk_step(st, &w, end, eof))
            return taken; /* the window (or the stream) ran out in front of the next packet */
        taken++;
        if (st->pos >= end)
            return taken;
    }
}

static size_t format_lines(const ts_scan_result *res, const ts_sync_error *errs, uint64_t nerrs, const ts_discontinuity *discs,
                           uint64_t ndiscs, char *buf, size_t cap)
{
    size_t used = 0;
    if (!buf || cap == 0)
        return 0;
    buf[0] = 0;
    uint64_t d = 0;
    for (uint64_t k = 0; used + 1 < cap; k++) { /* the two kinds of line in the order the reference printed them */
        for (; d < ndiscs && discs[d].after_sync_errors <= k && used + 1 < cap; d++) {
            const int w = snprintf(buf + used, cap - used, "Discontinuity!, pid = %d <0x%04x>, received = %2d, expected = %2d, at %lld\n",
                                   (int)discs[d].pid, (unsigned)discs[d].pid, (int)discs[d].received, (int)discs[d].expected,
                                   (long long)discs[d].at_packet);
            if (w < 0 || (size_t)w >= cap - used)
                return used;
            used += (size_t)w;
        }
        if (k >= nerrs)
            break;
        const int w = snprintf(buf + used, cap - used, "Transport Sync Error, skipped %d bytes, at %lld\n",
                               (int)errs[k].skipped, (long long)errs[k].at_packet);
        if (w < 0 || (size_t)w >= cap - used)
            return used;
        used += (size_t)w;
    }
    for (int i = 0; i < TS_PIDS && used + 1 < cap; i++) {
        if (res->count[i] == 0)
            continue;
        const int w = snprintf(buf + used, cap - used, "packets for pid %4d <0x%04x> = %d, first = %lld, last = %lld\n", i, i,
                               (int)res->count[i], (long long)res->first[i], (long long)res->last[i]);
        if (w < 0 || (size_t)w >= cap - used)
            return used;
        used += (size_t)w;
    }
    return used;
}

size_t ts_format_report(const ts_scan_result *res, char *buf, size_t cap)
{
    if (res->nsync_errors > TS_MAX_SYNC_ERRORS || res->ndiscontinuities > TS_MAX_DISCONTINUITIES) { /* not all inline: no truncated report */
        if (buf && cap)
            buf[0] = 0;
        return 0;
    }
    return format_lines(res, res->sync_errors, res->nsync_errors, res->discontinuities, res->ndiscontinuities, buf, cap);
}

size_t ts_format_report_all(const ts_scan_result *res, const ts_sync_error *errors, uint64_t nerrors,
                            const ts_discontinuity *discs, uint64_t ndiscs, char *buf, size_t cap)
{
    return format_lines(res, errors, nerrors, discs, ndiscs, buf, cap);
}
