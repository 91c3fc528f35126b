/*
 * ts_host.c — the GPU-free half of the transport-stream packet scan (include/ts_hip.h): the closed-form packet
 * walker that carries the scan across everything the GPU launches do not take (sync loss, false sync bytes,
 * malformed adaptation fields, the read-boundary quirk, the truncated tail), and the report formatter.
 *
 * The reference walks the stream byte by byte through one state machine (xport.c:2842-4375).  For the report this
 * scan reproduces, what a packet does to that machine has a closed form — where the next sync search starts, and
 * what is left over for the next packet — which is what ts_walk evaluates, one packet per step:
 *
 *   sync search   (xport.c:4317-4373)  bytes that are not 0x47 are skipped and counted; HDMV mode swallows four
 *                                      bytes of tp_extra_header unconditionally in front of every search
 *   header        (xport.c:2844-2906)  bytes 1, 2: error indicator + PID, packet_counter++, the PID's statistics;
 *                                      byte 3: adaptation_field_control
 *   adaptation    (xport.c:2908-2984)  a fresh length byte replaces whatever an earlier malformed field still owed;
 *                                      the field's bytes are taken singly and stop at the packet's 188th byte —
 *                                      the remainder is owed by the NEXT packet's payload (`stale_af`)
 *   payload       PID 0 and 0x1ffb     (xport.c:2985-3112, 3875-4295) consumed within the read: the packet ends at 188
 *                 every other PID      (xport.c:4296-4315) skipped in one step whose bound check `(length - i) >=
 *                                      xport_packet_length` lets a packet that ends exactly one byte past a
 *                                      16384-byte read finish one byte early: the search resumes ON its last byte
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

static void count_packet(ts_scan_result *res, unsigned h1, unsigned h2)
{
    res->packets++; /* xport.c:2860 */
    if ((h1 & 0x80u) == 0) { /* transport_error_indicator clear, xport.c:2861-2867 */
        const unsigned pid = ((h1 & 0x1fu) << 8) | h2;
        res->count[pid]++;
        if (res->first[pid] == 0)
            res->first[pid] = res->packets;
        res->last[pid] = res->packets;
    }
}

uint64_t ts_walk(ts_walk_state *st, const unsigned char *data, uint64_t base, uint64_t n, int eof, uint64_t min_packets,
                 ts_scan_result *res)
{
    const uint64_t end = base + n;
    uint64_t taken = 0;
    if (st->pos < base)
        return 0;
    for (;;) {
        if (taken >= min_packets && ts_walk_is_clean(st))
            return taken;
        /* ---- sync search ---- */
        uint64_t p = st->pos;
        while (p < end) {
            if (st->hdmv && st->extra_pending) {
                st->extra_pending--;
            } else if (data[p - base] == 0x47) {
                break;
            } else {
                st->skipped++;
            }
            p++;
        }
        st->pos = p;
        if (p >= end)
            return taken; /* the window (or the stream) ends inside the search */
        if (!eof && end - p < 189)
            return taken; /* the packet — and the byte behind it — must be in the window: ask for a later one */
        const uint64_t s = p, avail = end - s;
        if (st->skipped) { /* xport.c:4324-4327 */
            if (res->nsync_errors < TS_MAX_SYNC_ERRORS) {
                res->sync_errors[res->nsync_errors].skipped = st->skipped;
                res->sync_errors[res->nsync_errors].at_packet = res->packets;
            }
            res->nsync_errors++;
            st->skipped = 0;
        }
        if (st->hdmv)
            st->extra_pending = 4;
        const unsigned char *b = data + (s - base);
        /* ---- header ---- */
        if (avail < 3) { /* the stream ends before the PID is complete: nothing is counted */
            st->pos = end;
            return taken;
        }
        count_packet(res, b[1], b[2]);
        taken++;
        const unsigned pid = ((b[1] & 0x1fu) << 8) | b[2];
        if (avail < 4) {
            st->pos = end;
            return taken;
        }
        uint64_t q = s + 4;     /* next unconsumed byte */
        uint32_t left = 184;    /* bytes of this packet still to consume */
        uint32_t af = st->stale_af;
        if (b[3] & 0x20u) { /* adaptation_field_control & 2: a length byte follows (and replaces what was owed) */
            if (q >= end) {
                st->pos = end;
                return taken;
            }
            af = b[4];
            q++;
            left--;
        }
        const uint32_t take = af < left ? af : left;
        q += take;
        left -= take;
        st->stale_af = af - take;
        const uint64_t p_end = s + 188;
        uint64_t next = p_end;
        if (left != 0 && pid != 0 && pid != 0x1ffbu) {
            /* the one-step skip, entered at byte q: the read that holds q ends at the next multiple of 16384 */
            const uint64_t read_end = (q / TS_READ_CHUNK + 1) * TS_READ_CHUNK;
            if (p_end == read_end + 1)
                next = read_end; /* declared finished one byte early */
        }
        if (next > end) { /* truncated tail (only with eof) */
            st->pos = end;
            return taken;
        }
        st->pos = next;
    }
}

size_t ts_format_report(const ts_scan_result *res, char *buf, size_t cap)
{
    size_t used = 0;
    if (!buf || cap == 0)
        return 0;
    buf[0] = 0;
    for (uint64_t k = 0; k < res->nsync_errors && k < TS_MAX_SYNC_ERRORS && used + 1 < cap; k++) {
        const int w = snprintf(buf + used, cap - used, "Transport Sync Error, skipped %d bytes, at %lld\n",
                               (int)res->sync_errors[k].skipped, (long long)res->sync_errors[k].at_packet);
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
