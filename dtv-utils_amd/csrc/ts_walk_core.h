/*
 * ts_walk_core.h — ONE step of the transport-stream packet walker: the sync search and the packet it ends on.
 *
 * The reference walks a stream byte by byte through one state machine (xport.c:2842-4375).  For the report this scan
 * reproduces, what a packet does to that machine has a closed form — where the next sync search starts and what is
 * left over for the next packet — which is what this step evaluates.  It is written once and compiled twice: by gcc
 * into the host walker (ts_host.c: ts_walk, exported, tested without a GPU against the oracle and the reference's
 * recordings) and by hipcc into the scan kernel (ts_kernels.hip), where one wave runs it across every packet that is
 * not at its place in the regular stride.  The includer supplies how bytes are fetched, how a run of non-sync bytes is
 * passed over, and where counts and messages go:
 *
 *   TS_CORE_QUAL                      qualifiers of the generated function
 *   TS_CORE_NAME                      its name
 *   TS_CORE_CTX                       type of the opaque context argument handed to the hooks
 *   TS_CORE_BYTE(ctx, off)            the stream's byte at file offset `off`
 *   TS_CORE_FIND_SYNC(ctx, from, end) first offset in [from, end) whose byte is 0x47, or `end`
 *   TS_CORE_COUNT(ctx, h1, h2)        a packet: header bytes 1 and 2 (xport.c:2860-2867)
 *   TS_CORE_CC(ctx, pid, b3)          header byte 3 of the packet just counted: its continuity counter (xport.c:2872-2889)
 *   TS_CORE_SYNC_ERROR(ctx, skipped)  the stream locked again after `skipped` bytes (xport.c:4324-4327)
 *   TS_CORE_STOP_AT(ctx, s)           optional: the search has ended on the sync byte at `s` — stop in front of that packet
 *                                     (nothing reported, nothing consumed; st->pos == s, st->skipped as it stands)?
 *                                     Then the step returns 2.  (The scan's merge walks up to a place it knows.)
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
 *
 * `end` = file offset behind the last byte that may be looked at; `eof` = that is the end of the stream.  Returns 1 if a
 * packet was counted, 0 if the step stopped in front of one (the data ran out, or — eof == 0 — the packet and the byte
 * behind it are not all there yet: st->pos then says from where a later window must go on).
 */
#ifndef TS_CORE_STOP_AT
#define TS_CORE_STOP_AT(ctx, s) 0
#endif
TS_CORE_QUAL int TS_CORE_NAME(ts_walk_state *st, TS_CORE_CTX ctx, uint64_t end, int eof)
{
    uint64_t p = st->pos;
    /* ---- sync search ---- */
    if (st->hdmv && st->extra_pending && p < end) {
        const uint64_t k = st->extra_pending < end - p ? st->extra_pending : end - p;
        p += k;
        st->extra_pending -= (uint32_t)k;
    }
    if (!(st->hdmv && st->extra_pending) && p < end) {
        const uint64_t q = TS_CORE_FIND_SYNC(ctx, p, end);
        st->skipped += q - p;
        p = q;
    }
    st->pos = p;
    if (p >= end)
        return 0; /* the window (or the stream) ends inside the search */
    if (!eof && end - p < 189)
        return 0; /* the packet — and the byte behind it — must be in the window: ask for a later one */
    const uint64_t s = p, avail = end - s;
    if (TS_CORE_STOP_AT(ctx, s))
        return 2;
    if (st->skipped) { /* xport.c:4324-4327 */
        TS_CORE_SYNC_ERROR(ctx, st->skipped);
        st->skipped = 0;
    }
    if (st->hdmv)
        st->extra_pending = 4;
    /* ---- header ---- */
    if (avail < 3) { /* the stream ends before the PID is complete: nothing is counted */
        st->pos = end;
        return 0;
    }
    const unsigned h1 = TS_CORE_BYTE(ctx, s + 1), h2 = TS_CORE_BYTE(ctx, s + 2);
    TS_CORE_COUNT(ctx, h1, h2);
    const unsigned pid = ((h1 & 0x1fu) << 8) | h2;
    if (avail < 4) {
        st->pos = end;
        return 1;
    }
    uint64_t q = s + 4;  /* next unconsumed byte */
    uint32_t left = 184; /* bytes of this packet still to consume */
    uint32_t af = st->stale_af;
    const unsigned h3 = TS_CORE_BYTE(ctx, s + 3);
    TS_CORE_CC(ctx, pid, h3);
    if (h3 & 0x20u) { /* adaptation_field_control & 2: a length byte follows (and replaces what was owed) */
        if (q >= end) {
            st->pos = end;
            return 1;
        }
        af = TS_CORE_BYTE(ctx, s + 4);
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
    st->pos = next > end ? end : next; /* (> end: the truncated tail, only with eof) */
    return 1;
}
