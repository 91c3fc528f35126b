// ts_kernels.hip — gfx950 kernels of the transport-stream packet scan (include/ts_hip.h; reference xport.c).
//
// The stream is cut into one byte range ("span") per CU.  Inside a span the packets are taken the fast way as long as
// they sit on their grid: from a clean position the next 1024 units are classified one lane per packet header (8
// aligned bytes out of the unit's 188 / 192: the scan touches a third of the stream's 64-byte sectors, not its payload)
// and everything in front of the first irregular one is counted into per-workgroup tables in LDS (count / first /
// last per PID, 96 KiB).  Irregular = anything whose effect on the reference's state machine is not local to the
// packet: sync byte missing, packet cut off by the end of the stream, adaptation field longer than the packet, the
// packet ends exactly one byte past a 16384-byte read of the reference while its payload is skipped in one step
// (xport.c:4302) and what follows is not simply the next sync byte.  Across those, wave 0 of the workgroup runs the
// packet walker itself — ts_walk_core.h, the same closed-form step the host library exports as ts_walk and the tests
// pin against the reference without a GPU — until the stream is back on a grid (any grid: an inserted or deleted
// byte moves it), and the workgroup goes on in blocks from there.  Damage therefore costs its own bytes, once, on
// the device: there is no hand-over to the host and no relaunch per irregularity.
//
// A span cannot know where the chain of packets enters it, so it SPECULATES: the first position in its first
// stride-ful of bytes from which eight sync bytes in a row sit at the packet stride.  ts_merge_kernel then checks
// the chain — every span must have started exactly where, and in the state in which, the one in front of it ended —
// and folds the spans' tables (with stream-wide packet numbers) as far as it holds; where it does not (damage across
// a span boundary, a payload that imitates a grid), the host launches THAT span again from the true state and the
// merge goes on.  Speculation decides how many launches a scan takes, never a number in its result.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ts_hip.h"
#include "ts_kernels.h"
#include "ts_synth.h"

namespace {

constexpr int kScanBlock = 1024;   // the full-table form: one workgroup per CU
constexpr int kSlotBlock = 512;    // the slot-table form: two workgroups per CU
constexpr int kMergeBlock = 256;
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kEntrySyncs = 8;  // sync bytes in a row, at the stride, that make a position a span's speculated entry
constexpr uint32_t kEntryBytes = 16384;  // ... looked for in the span's first 16 KiB
constexpr uint32_t kEntryLater = 6;      // packets a span that begins in damage enters behind the first such position

// ---- the walker on the device: ts_walk_core.h with these hooks, run by wave 0 with all 64 lanes in step ----
// The walker reads the stream through a WINDOW in LDS: 4 KiB brought in by one cooperative load (four 16-byte loads per
// lane in flight: one trip to memory), from which its header bytes and its sync searches are served.  Read straight from
// global memory every byte the walker step looks at was a dependent trip of its own, 1-2 us each, and a damaged spot —
// a burst of garbage with a false sync byte every few hundred bytes — cost ~30 us of its span's time.
// ---- PID slots (the scan's second form) ----
// 96 KiB of per-PID tables allow ONE workgroup per CU, and everything a span does at a damaged spot — a partial block, the
// walker's window, the next block's headers: three dependent trips to memory with nothing else to run — is then the CU's
// time.  A stream uses a few dozen PIDs, not 8192: the slot form keeps count / first / last / continuity state per SLOT
// (kSlots of them), a PID gets its slot at its first packet in the span (s_slot: PID -> slot + 1), the workgroup needs
// 46 KiB and 512 threads, and TWO spans share a CU: one's stalls are the other's time.  A span that meets more PIDs than
// it has slots (garbage read as packets carries any PID) says so; the host then scans again with the full tables.
constexpr uint32_t kSlots = 1024;
constexpr uint32_t kSlotClaim = 0xFFFFu;  // a PID's half word of s_slot while the lane that saw the PID first takes a slot for it

// s_slot packs two PIDs into a word (16 KiB for the 8192 of them): 0 = no slot yet, kSlotClaim, or slot + 1
__device__ __forceinline__ uint32_t slot_peek(const uint32_t *s_slot, uint32_t pid)
{
    return (__hip_atomic_load(&s_slot[pid >> 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >> ((pid & 1u) * 16u)) & 0xFFFFu;
}

// slot of `pid` (slot_limit: what a dummy slot is handed out beyond — its numbers are never used)
__device__ __forceinline__ uint32_t slot_of(uint32_t *s_slot, uint32_t *s_nslots, uint16_t *s_slot_pid, uint32_t *s_over, uint32_t slot_limit,
                                            uint32_t pid)
{
    const uint32_t sh = (pid & 1u) * 16u;
    uint32_t *wp = &s_slot[pid >> 1];
    uint32_t w = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    uint32_t v = (w >> sh) & 0xFFFFu;
    if (v == 0u || v == kSlotClaim) {
        bool mine = false;
        while (((w >> sh) & 0xFFFFu) == 0u) {  // claim the half word if it is still empty (the other half may change meanwhile)
            const uint32_t seen = atomicCAS(wp, w, w | (kSlotClaim << sh));
            if (seen == w) {
                mine = true;
                break;
            }
            w = seen;
        }
        if (mine) {  // this lane assigns it
            const uint32_t n = atomicAdd(s_nslots, 1u);
            if (n < slot_limit) {
                s_slot_pid[n] = (uint16_t)pid;
                v = n + 1u;
            } else {
                *s_over = 1u;
                v = kSlots + 1u;  // the dummy slot
            }
            __threadfence_block();
            atomicXor(wp, (kSlotClaim ^ v) << sh);  // claim -> slot + 1, the other half untouched
        }
        // (another lane does — of this wave: it has, the branch above lies in front of this loop — or of another wave)
        do {
            v = slot_peek(s_slot, pid);
        } while (v == kSlotClaim);
    }
    return v - 1u;
}

constexpr uint32_t kWalkWindow = 4096;
constexpr uint32_t kSpecBefore = 5;  // the look-ahead across a damaged spot covers a grid that moved by -kSpecBefore .. +3 bytes (16 bytes a lane)
constexpr uint32_t kWinAhead = 1024;  // ... and this much of the walker's window (one 16-byte piece a lane)
constexpr uint64_t kBridgeMax = 1u << 20;   // a bridge longer than this is the host's (a launch of the span from the true state)
constexpr uint32_t kBridgeSteps = 8192;     // ... or one of more packets than this
constexpr uint32_t kBridgeEvent = 0x80000000u;  // ts_event::attempt of a bridge's events (numbered from the bridge's first packet)

struct DevWalk {
    const unsigned char *data;
    uint32_t *s_count, *s_first, *s_last;  // the workgroup's tables (LDS); the other waves wait at a barrier meanwhile
    uint64_t packets;                      // the span's packet counter (the same in every lane)
    ts_event *events;
    unsigned int *event_count;
    uint32_t event_cap, span, attempt, lane;
    unsigned char *win;                    // LDS, 16-byte aligned
    uint64_t nbytes, win_base;             // the window holds the stream's bytes [win_base, win_base + win_len)
    uint32_t win_len;
    // the merge kernel's bridges (below) count into the STREAM-WIDE tables, or not at all (a dry run):
    uint32_t *g_count;                     // null: the workgroup's LDS tables above
    unsigned long long *g_first, *g_last;
    uint64_t abs0;                         // stream-wide number of the walk's first packet
    uint64_t stop_at;                      // the sync byte in front of which the walk is to stop (TS_NO_ENTRY: nowhere)
    uint32_t quiet;                        // 1: count packets only — no table, no event
    // continuity counters (xport.c:2872-2889): a span's walker keeps them in the workgroup's table; a bridge's packets
    // are reported one by one for the host to check (s_cc == nullptr)
    unsigned char *s_cc;                   // LDS, per PID: last counter + 1, 0 = no payload packet of the PID in this span yet
    uint32_t *s_ncc;                       // LDS: entries in the span's continuity list
    ts_cc_entry *cc_list;
    uint32_t *s_ev;                        // LDS, two words: next reserved slot of the event list, slots left (null: none kept)
    // the slot form of the tables (null: they are indexed by the PID itself)
    uint32_t *s_slot, *s_nslots, *s_over;
    uint16_t *s_slot_pid;
    uint32_t slot_limit;
};

// where the walker's tables keep `pid` (called by ONE lane)
__device__ __forceinline__ uint32_t walk_idx(const DevWalk *w, uint32_t pid)
{
    return w->s_slot ? slot_of(w->s_slot, w->s_nslots, w->s_slot_pid, w->s_over, w->slot_limit, pid) : pid;
}

__device__ __forceinline__ bool walk_is_clean(const ts_walk_state &st)
{
    return st.skipped == 0 && st.stale_af == 0 && (!st.hdmv || st.extra_pending == 4u);
}

// bring the window to `off` (wave-uniform; off < nbytes).  Its base is `off` rounded down to a 16-byte boundary OF THE
// BUFFER, so that whole 16-byte loads can be used (an adopted buffer need not be aligned: then it is bytes)
__device__ __forceinline__ void win_fill(DevWalk *w, uint64_t off)
{
    const uint32_t lane = w->lane;
    const uint32_t mis = (uint32_t)(((uintptr_t)w->data + off) & 15u);
    const bool aligned = off >= mis;
    const uint64_t base = aligned ? off - mis : off;
    const uint64_t room = w->nbytes - base;
    const uint32_t len = room < kWalkWindow ? (uint32_t)room : kWalkWindow;
    __builtin_amdgcn_wave_barrier();  // (earlier reads of the window are done: same wave, in order — this is for the compiler)
    const uint32_t quads = aligned ? len / 16u : 0u;
    for (uint32_t q = lane; q < quads; q += 64u)  // (kWalkWindow / 16 / 64 = 4 loads per lane, issued back to back)
        *reinterpret_cast<uint4 *>(w->win + 16u * q) = *reinterpret_cast<const uint4 *>(w->data + base + 16u * q);
    for (uint32_t b = 16u * quads + lane; b < len; b += 64u)
        w->win[b] = w->data[base + b];
    __builtin_amdgcn_wave_barrier();
    w->win_base = base;
    w->win_len = len;
}

__device__ __forceinline__ unsigned dev_byte(DevWalk *w, uint64_t off)
{
    if (off - w->win_base >= (uint64_t)w->win_len)  // (also when off lies in front of the window)
        win_fill(w, off);
    return w->win[off - w->win_base];
}

// first offset in [from, end) that holds 0x47, or end: 1 KiB of the window per step (16 bytes per lane)
__device__ __forceinline__ uint64_t dev_find_sync(DevWalk *w, uint64_t from, uint64_t end)
{
    const uint32_t lane = w->lane;
    uint64_t pos = from;
    while (pos < end) {  // (wave-uniform)
        if (pos - w->win_base >= (uint64_t)w->win_len)
            win_fill(w, pos);
        const uint32_t rel = (uint32_t)(pos - w->win_base);  // < win_len
        const uint64_t left = end - w->win_base;
        const uint32_t lim = left < (uint64_t)w->win_len ? (uint32_t)left : w->win_len;  // bytes of the window that may be looked at
        const uint32_t c0 = (rel & ~15u) + 16u * lane;  // this lane's 16 bytes
        uint32_t idx = 16;
        if (c0 < lim) {
            const uint4 v = *reinterpret_cast<const uint4 *>(w->win + c0);  // (bytes behind win_len: stale, masked off below)
            const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
            uint32_t bits = 0;  // bit i: byte i of the 16 is 0x47 (exact: no borrow tricks, every flagged byte is a match)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t x = wd[k] ^ 0x47474747u;
                const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);  // 0x80 in every zero byte of x
                bits |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * k);
            }
            const uint32_t lo = rel > c0 ? rel - c0 : 0u;                    // (only the first lane's chunk starts in front of pos)
            const uint32_t hi = lim - c0 < 16u ? lim - c0 : 16u;
            bits &= ((1u << hi) - 1u) & ~((1u << lo) - 1u);
            if (bits)
                idx = (uint32_t)__ffs((int)bits) - 1u;
        }
        const unsigned long long m = __ballot(idx < 16u);
        if (m) {
            const int l = __ffsll((long long)m) - 1;
            return w->win_base + (rel & ~15u) + 16u * (uint32_t)l + (uint32_t)__builtin_amdgcn_readlane((int)idx, l);
        }
        const uint32_t next = (rel & ~15u) + 1024u;
        pos = w->win_base + (next < lim ? next : lim);
    }
    return end;
}

__device__ __forceinline__ void dev_count(DevWalk *w, unsigned h1, unsigned h2)
{
    const uint32_t rel = (uint32_t)w->packets;  // (a span counts < 2^32 packets)
    w->packets++;
    if (w->quiet)
        return;
    if (w->g_count) {
        if (w->lane == 0 && (h1 & 0x80u) == 0) {
            const uint32_t pid = ((h1 & 0x1fu) << 8) | h2;
            atomicAdd(&w->g_count[pid], 1u);
            atomicMin(&w->g_first[pid], w->abs0 + rel + 1);
            atomicMax(&w->g_last[pid], w->abs0 + rel + 1);
        }
        return;
    }
    if (w->lane == 0 && (h1 & 0x80u) == 0) {  // transport_error_indicator clear, xport.c:2861-2867
        const uint32_t ix = walk_idx(w, ((h1 & 0x1fu) << 8) | h2);
        w->s_count[ix]++;
        if (rel < w->s_first[ix])
            w->s_first[ix] = rel;
        if (rel > w->s_last[ix])
            w->s_last[ix] = rel;
    }
}

__device__ __forceinline__ void put_event(ts_event *events, unsigned int *event_count, uint32_t event_cap, uint32_t span,
                                          uint32_t attempt, uint32_t kind, uint64_t skipped, uint64_t at_rel, uint32_t info)
{
    const unsigned int slot = atomicAdd(event_count, 1u);  // (counts what no longer fits: the host sees the overflow)
    if (slot < event_cap) {
        ts_event e;
        e.skipped = skipped;
        e.at_rel = at_rel;
        e.span = span;
        e.attempt = attempt;
        e.kind = kind;
        e.info = info;
        events[slot] = e;
    }
}

// A span's lines get their slots of the event list out of a POOL the workgroup keeps in LDS (s_ev: next slot, slots left),
// refilled kEvBatch slots at a time with one returning atomic on the list's counter: the walker's lines come one at a time,
// and an atomic each (~2 us of the one wave that walks) was a third of a damaged stream's scan.  A refill marks its slots
// empty (a span number no span has: the host skips them) before any is used.  All of a span's slots come out of the pool
// one after the other, so they grow with the stream: the host finds a span's lines in the reference's order.
// Called by ALL lanes of one wave with the same arguments; returns the first of the n slots.
constexpr uint32_t kEvBatch = 64;
__device__ __forceinline__ unsigned int ev_take(uint32_t *s_ev, uint32_t n, uint32_t lane, ts_event *events, unsigned int *event_count,
                                                uint32_t event_cap)
{
    uint32_t next = s_ev[0], left = s_ev[1];
    __builtin_amdgcn_wave_barrier();
    if (left < n) {  // (wave-uniform) what is left of the old batch is dropped: it lies in front of these slots
        const uint32_t want = n > kEvBatch ? n : kEvBatch;
        unsigned int base = 0;
        if (lane == 0)
            base = atomicAdd(event_count, want);  // (counts what no longer fits: the host sees the overflow)
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
        for (uint32_t k = lane; k < want; k += 64u)
            if (base + k < event_cap) {
                ts_event e;
                e.skipped = e.at_rel = 0;
                e.span = 0xFFFFFFFFu;
                e.attempt = e.kind = e.info = 0;
                events[base + k] = e;
            }
        next = base;
        left = want;
    }
    if (lane == 0) {
        s_ev[0] = next + n;
        s_ev[1] = left - n;
    }
    __builtin_amdgcn_wave_barrier();
    return next;
}

// a line of the walker (all 64 lanes of the walking wave are here, in step)
__device__ __forceinline__ void walk_event(const DevWalk *w, uint32_t kind, uint64_t skipped, uint64_t at_rel, uint32_t info)
{
    if (w->quiet)
        return;
    if (!w->s_ev) {  // a bridge of the merge kernel: no pool (its lines are few)
        if (w->lane == 0)
            put_event(w->events, w->event_count, w->event_cap, w->span, w->attempt, kind, skipped, at_rel, info);
        return;
    }
    const unsigned int slot = ev_take(w->s_ev, 1u, w->lane, w->events, w->event_count, w->event_cap);
    if (w->lane == 0 && slot < w->event_cap) {
        ts_event e;
        e.skipped = skipped;
        e.at_rel = at_rel;
        e.span = w->span;
        e.attempt = w->attempt;
        e.kind = kind;
        e.info = info;
        w->events[slot] = e;
    }
}

__device__ __forceinline__ void dev_event(const DevWalk *w, uint64_t skipped, uint64_t at_rel)
{
    walk_event(w, TS_EV_SYNC, skipped, at_rel, 0u);
}

// header byte 3 of the packet the walker has just counted (w->packets is its number within the walk, 1-based); all lanes
// of the walking wave are here with the same arguments
__device__ __forceinline__ void dev_cc(DevWalk *w, unsigned pid, unsigned h3)
{
    if (w->quiet || (h3 & 0x10u) == 0 || pid == 0)
        return;  // no payload: neither checked nor remembered; PID 0 is never remembered, hence never reported
    const uint32_t cc = h3 & 0xfu;
    if (!w->s_cc) {  // a bridge: the host checks it between the spans it links
        walk_event(w, TS_EV_BRIDGE_CC, 0, w->packets, (pid << 8) | (cc << 4));
        return;
    }
    uint32_t ix = 0;
    if (w->lane == 0)
        ix = walk_idx(w, pid);
    ix = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix);
    const uint32_t last = w->s_cc[ix];  // (every lane reads the same byte)
    __builtin_amdgcn_wave_barrier();
    if (last == 0) {
        if (w->lane == 0) {
            ts_cc_entry e;
            e.pid = (uint16_t)pid;
            e.first_cc = (uint8_t)cc;
            e.last_cc = 0;
            e.first_rel = (uint32_t)(w->packets - 1);
            w->cc_list[(*w->s_ncc)++] = e;
        }
    } else if (pid != 0x1fffu && (last & 0xfu) != cc) {
        walk_event(w, TS_EV_DISC, 0, w->packets, (pid << 8) | (cc << 4) | (last & 0xfu));
    }
    if (w->lane == 0)
        w->s_cc[ix] = (unsigned char)(cc + 1u);
    __builtin_amdgcn_wave_barrier();
}

#define TS_CORE_QUAL __device__ __forceinline__
#define TS_CORE_NAME dev_walk_step
#define TS_CORE_CTX DevWalk *
#define TS_CORE_BYTE(ctx, off) dev_byte(ctx, off)
#define TS_CORE_FIND_SYNC(ctx, from, end) dev_find_sync(ctx, from, end)
#define TS_CORE_COUNT(ctx, h1, h2) dev_count(ctx, h1, h2)
#define TS_CORE_CC(ctx, pid, h3) dev_cc(ctx, pid, h3)
#define TS_CORE_SYNC_ERROR(ctx, skipped) dev_event(ctx, skipped, (ctx)->packets)
#define TS_CORE_STOP_AT(ctx, s) ((ctx)->stop_at == (s))
#include "ts_walk_core.h"

}  // namespace

template <int BLOCK, bool SLOTS>
__global__ __launch_bounds__(BLOCK) void ts_scan_kernel(const ts_scan_params prm)
{
    constexpr int kScanBlock = BLOCK;               // (the workgroup's size: a block of the scan is one packet per thread)
    constexpr uint32_t NT = SLOTS ? kSlots + 1u : (uint32_t)TS_PIDS;  // entries of the per-PID (per-slot) tables
    // (the fields the loop needs, as values: taken out of the by-value argument block once — left inside the struct the
    // compiler re-reads them from its stack copy in every iteration)
    struct {
        const unsigned char *data;
        uint64_t nbytes, span_bytes;
        uint32_t first_span, nspans_total, stride, sync_offset, hdmv, attempt, explicit_entry, quirk_events, event_cap, slot_limit, abort_walks, lookahead;
        ts_wg_entry *lists;
        ts_span_rec *recs;
        ts_cc_entry *cc_lists;
        ts_event *events;
        unsigned int *event_count;
    } p;
    p.data = prm.data;
    p.nbytes = prm.nbytes;
    p.span_bytes = prm.span_bytes;
    p.first_span = prm.first_span;
    p.nspans_total = prm.nspans_total;
    p.stride = prm.stride;
    p.sync_offset = prm.sync_offset;
    p.hdmv = prm.hdmv;
    p.attempt = prm.attempt;
    p.explicit_entry = prm.explicit_entry;
    p.quirk_events = prm.quirk_events;
    p.event_cap = prm.event_cap;
    p.slot_limit = prm.slot_limit && prm.slot_limit < kSlots ? prm.slot_limit : kSlots;
    p.abort_walks = SLOTS ? 0u : prm.abort_walks;
    p.lookahead = prm.lookahead;
    p.lists = prm.lists;
    p.recs = prm.recs;
    p.cc_lists = prm.cc_lists;
    p.events = prm.events;
    p.event_count = prm.event_count;
    extern __shared__ __attribute__((aligned(16))) uint32_t ts_smem[];  // 3 x NT words (full tables: 96 KiB, one workgroup per CU)
    uint32_t *s_count = ts_smem, *s_first = ts_smem + NT, *s_last = ts_smem + 2 * NT;
    __shared__ ts_walk_state s_st;
    __shared__ unsigned long long s_packets, s_block_packets, s_run_start;
    __shared__ uint32_t s_stop, s_walks, s_entries, s_cand, s_ncc, s_ev[2], s_evbase, s_nid, s_evn[kScanBlock / 64];
    __shared__ unsigned char s_cc[NT];  // per PID: last continuity counter + 1 (0: no payload packet in this span yet)
    // the slot form: PID -> slot + 1 (0: none yet), the slots' PIDs, slots handed out, "more PIDs than slots"
    __shared__ uint32_t s_slot[SLOTS ? TS_PIDS / 2 : 1];  // (two PIDs a word)
    __shared__ uint16_t s_slot_pid[SLOTS ? kSlots : 1];
    __shared__ uint32_t s_nslots, s_over, s_abort;
    // the continuity check across the waves of ONE block (below): per PID the number of its pair of words for this block,
    // the waves that hold the PID, and each such wave's last counter of it (a nibble per wave)
    __shared__ uint32_t s_bid[NT];
    __shared__ uint32_t s_waves[kScanBlock + 1];
    __shared__ unsigned long long s_lastcc[kScanBlock + 1];
    __shared__ __attribute__((aligned(16))) unsigned char s_window[kWalkWindow];  // the walker's view of the stream (wave 0)
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t span = p.first_span + blockIdx.x;
    const uint64_t B0 = (uint64_t)span * p.span_bytes;
    const uint64_t B1 = (span + 1 == p.nspans_total || B0 + p.span_bytes > p.nbytes) ? p.nbytes : B0 + p.span_bytes;
    for (uint32_t k = t; k < NT; k += kScanBlock) {
        s_count[k] = 0;
        s_first[k] = kNone;
        s_last[k] = 0;
        s_cc[k] = 0;
        s_bid[k] = 0;
    }
    if constexpr (SLOTS)
        for (uint32_t k = t; k < TS_PIDS / 2; k += kScanBlock)
            s_slot[k] = 0;
    // where the tables keep a PID
    auto idx = [&](uint32_t pid) -> uint32_t {
        if constexpr (SLOTS)
            return slot_of(s_slot, &s_nslots, s_slot_pid, &s_over, p.slot_limit, pid);
        else
            return pid;
    };
    for (uint32_t k = t; k < kScanBlock + 1; k += kScanBlock) {
        s_waves[k] = 0;
        s_lastcc[k] = 0;
    }
    ts_cc_entry *cc_list = p.cc_lists + (size_t)span * TS_PIDS;
    if (t == 0) {
        s_nslots = 0;
        s_over = 0;
        s_abort = 0;
        s_nid = 0;
        s_ncc = 0;
        s_ev[0] = s_ev[1] = 0;
        s_stop = kNone;
        s_walks = 0;
        s_entries = 0;
        s_cand = kNone;
        s_packets = 0;
        s_block_packets = 0;
        ts_walk_state st;
        st.pos = 0;
        st.skipped = 0;
        st.stale_af = 0;
        st.extra_pending = p.hdmv ? 4u : 0u;
        st.hdmv = (int)p.hdmv;
        if (p.explicit_entry) {
            st.pos = prm.entry.pos;
            st.skipped = prm.entry.skipped;
            st.stale_af = prm.entry.stale_af;
            st.extra_pending = prm.entry.extra_pending;
        }
        s_st = st;
    }
    __syncthreads();
    // ---- where does the chain of packets enter this span?  (span 0: at the stream's first byte; a launch of one span
    // from a state the host hands in: there; otherwise speculated — the first unit start in the span's first stride-ful
    // of bytes behind which the sync bytes sit on the grid) ----
    if (!p.explicit_entry && span != 0) {
        // (the first offset of the span behind which the sync bytes sit on the grid — normally within its first stride-ful of
        // bytes; with damage right there, further in: the merge's bridge walks what lies in front, so the span is not lost)
        for (uint32_t round = 0; round < kEntryBytes / (uint32_t)kScanBlock; round++) {
            const uint64_t c = (uint64_t)round * kScanBlock + t;
            const uint64_t sy = B0 + c + p.sync_offset;
            bool ok = B0 + c < B1 && sy < p.nbytes;
            for (int i = 0; ok && i < kEntrySyncs; i++) {
                const uint64_t at = sy + (uint64_t)i * p.stride;
                if (at >= p.nbytes)
                    break;
                ok = p.data[at] == 0x47u;
            }
            if (ok)
                atomicMin(&s_cand, (uint32_t)c);
            __syncthreads();
            if (s_cand != kNone || (uint64_t)(round + 1) * kScanBlock >= B1 - B0)  // (workgroup-uniform)
                break;
            __syncthreads();  // (everyone has read s_cand before the next round's candidates go in)
        }
        if (s_cand == kNone) {  // (workgroup-uniform) nothing regular here: the span in front carries the chain across
            if (t == 0) {
                ts_span_rec r;
                r.entry = TS_NO_ENTRY;
                r.exit_pos = r.exit_skipped = 0;
                r.exit_stale_af = r.exit_extra = 0;
                r.packets = r.block_packets = 0;
                r.walks = r.nlist = 0;
                r.attempt = p.attempt;
                r.explicit_entry = 0;
                r.ncc = r.pad = 0;
                r.first_take = r.pad3 = 0;
                r.exit_run_start = TS_NO_ENTRY;
                p.recs[span] = r;
            }
            return;
        }
        // Nothing regular in the span's first stride-ful of bytes: there is damage right here, and the chain — the reference's
        // state machine coming out of it: a false sync byte in a payload, a bogus packet, another search — need not be back on
        // the grid at the FIRST position behind which eight sync bytes line up.  kEntryLater packets further on it almost
        // always is: the span enters there (if the sync bytes still line up), and what lies in front is the bridge's.
        if (s_cand >= p.stride) {  // (workgroup-uniform)
            const uint64_t later = (uint64_t)s_cand + (uint64_t)kEntryLater * p.stride;
            const uint64_t sy = B0 + later + p.sync_offset;
            bool ok = B0 + later < B1;
            if (t < (uint32_t)kEntrySyncs) {
                const uint64_t at = sy + (uint64_t)t * p.stride;
                if (at < p.nbytes && p.data[at] != 0x47u)
                    atomicOr(&s_entries, 1u);  // (s_entries is not in use yet: cleared again below)
            }
            __syncthreads();
            if (s_entries != 0u)
                ok = false;
            __syncthreads();
            if (t == 0) {
                s_entries = 0;
                if (ok)
                    s_cand = (uint32_t)later;
            }
            __syncthreads();
        }
        if (t == 0)
            s_st.pos = B0 + s_cand;
    }
    __syncthreads();
    // The state every thread carries (the same in all of them): while the stream is on its grid only `pos` and the
    // counters move, and they move by the same amount in every thread — no LDS, no extra barrier.  Only the walker's
    // result goes through LDS (s_st, s_packets).
    ts_walk_state st = s_st;
    const uint64_t entry_pos = st.pos;
    uint64_t packets = 0, block_packets = 0;
    uint32_t units_seen = 0;  // units the blocks of this span have looked at so far: block-independent indices for s_stop
    uint32_t walks = 0;
    uint32_t first_take = kNone;     // packets the span's very first block committed (kNone: no block yet)
    uint64_t run_start = st.pos;     // where the packets consumed back to back up to here began (blocks go on with a run, a walker's
                                     // packet that does not start where the one in front ended — or ends early — begins a new one)
    uint64_t pre_pos = TS_NO_ENTRY;  // the position whose block's header words are in pre_w0 / pre_w1 already
    uint32_t pre_w0 = 0, pre_w1 = 0;
    // Looking ahead across a damaged spot (below): the block behind it is expected one unit behind the first irregular one,
    // give or take a few bytes — 16 bytes around this lane's sync byte there, asked for when the partial block is known ...
    uint64_t spec_pos = TS_NO_ENTRY;  // the position those bytes were asked for (TS_NO_ENTRY: none)
    uint32_t spec_w[4] = {0, 0, 0, 0};
    bool spec_ok = false;
    // ... and so is the head of the walker's window (wave 0: one 16-byte piece per lane)
    uint64_t win_pre_base = 0;
    uint32_t win_pre_len = 0;
    uint4 win_pre = {0, 0, 0, 0};
    bool walk_next = false;
    for (;;) {
        const bool clean = walk_is_clean(st);
        if (st.pos >= p.nbytes || (clean && !walk_next && st.pos >= B1))
            break;  // (workgroup-uniform)
        if (clean && !walk_next) {
            // ---- a block: the next units on the grid, one lane each (those that START in this span) ----
            // (all 1024 units start inside the span — the usual case, decided without the 64-bit division)
            const uint64_t room = B1 - st.pos;
            const uint32_t nblk = room > (uint64_t)(kScanBlock - 1) * p.stride ? (uint32_t)kScanBlock
                                                                                : (uint32_t)((room + p.stride - 1) / p.stride);
            const uint64_t sy = st.pos + (uint64_t)t * p.stride + p.sync_offset;  // file offset of this lane's sync byte
            const bool mine = t < nblk, whole = mine && sy + 188 <= p.nbytes;
            // the five bytes that matter — sync, two PID bytes, adaptation_field_control, adaptation_field_length — out of
            // two aligned dwords (already on their way if the block in front of this one was a whole one: below)
            uint32_t w0, w1;
            if (pre_pos == st.pos) {
                w0 = pre_w0;
                w1 = pre_w1;
            } else {
                // (behind a walk: were this lane's bytes asked for in advance — kSpecBefore bytes early to 3 late?)
                const int64_t D = (int64_t)(st.pos - spec_pos);
                const bool hit = spec_pos != TS_NO_ENTRY && D >= -(int64_t)kSpecBefore && D <= 3 && spec_ok && whole;
                if (hit) {
                    const uint64_t q = spec_pos + (uint64_t)t * p.stride + p.sync_offset;
                    const uint32_t o = (uint32_t)(sy - ((q - kSpecBefore) & ~3ull));  // 0 .. 11: bytes o .. o + 4 of the 16
                    const uint32_t i = o >> 2;
                    w0 = i == 0 ? spec_w[0] : i == 1 ? spec_w[1] : spec_w[2];
                    w1 = i == 0 ? spec_w[1] : i == 1 ? spec_w[2] : spec_w[3];
                } else {
                    const uint64_t a = whole ? (sy & ~3ull) : 0ull;
                    w0 = *reinterpret_cast<const uint32_t *>(p.data + a);
                    w1 = *reinterpret_cast<const uint32_t *>(p.data + a + 4);
                }
            }
            spec_pos = TS_NO_ENTRY;
            const uint32_t sh = (uint32_t)(sy & 3u);
            const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, sh);  // bytes sy .. sy+3
            const uint32_t b4 = (w1 >> (8 * sh)) & 0xffu;                // byte sy+4 (sh <= 3: inside w1)
            const uint32_t b0 = lo & 0xffu, b1 = (lo >> 8) & 0xffu, b2 = (lo >> 16) & 0xffu, b3 = lo >> 24;
            const uint32_t tei = b1 >> 7, pid = ((b1 & 0x1fu) << 8) | b2;
            const bool has_af = (b3 & 0x20u) != 0;
            const uint32_t af_len = has_af ? b4 : 0u;
            bool regular = whole && b0 == 0x47u && af_len <= 183u;  // (!whole: cut off by the end of the stream)
            bool quirk = false;
            // the reference's one-step payload skip, entered before the last byte of a packet that ends one byte past a
            // 16384-byte read, finishes the packet a byte early (xport.c:4302): its last byte goes to the sync search.
            // `skipped 1 bytes` and nothing else happens only if that search ends on the NEXT unit's sync byte: that unit
            // must be there, whole, with its 0x47 in place, and the byte the search tests first — the packet's last byte,
            // or in HDMV mode (which swallows four bytes in front of every search, xport.c:4317) the last byte of the next
            // tp_extra_header — must not be a 0x47 itself.  Anything else (last packet of the stream: no line at all;
            // damage behind it: ONE line with the sum of the bytes skipped; a false sync: a re-lock one byte early) is
            // the walker's.
            if (regular && ((sy + 187) & (TS_READ_CHUNK - 1)) == 0 && pid != 0u && pid != 0x1ffbu && (!has_af || af_len <= 181u)) {
                const uint64_t probe = sy + 187 + p.sync_offset, next_sync = sy + p.stride;
                if (!p.quirk_events || next_sync + 188 > p.nbytes || p.data[probe] == 0x47u || p.data[next_sync] != 0x47u)
                    regular = false;
                else
                    quirk = true;
            }
            // (indices that keep growing over the span's blocks: a thread that is already in the next block cannot
            // disturb what a slower one still reads of this block — the one barrier per block is enough)
            if (mine && !regular)
                atomicMin(&s_stop, units_seen + t);
            // (a damaged stream is the slot form's: some span has said so — seen by ONE thread, published in front of the barrier)
            if (p.abort_walks && t == 0 && __hip_atomic_load(p.event_count + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
                s_abort = 1;
            __syncthreads();
            if (p.abort_walks && s_abort)
                break;  // (workgroup-uniform; nothing of this scan will be used)
            const uint32_t stop = s_stop - units_seen;  // (kNone - units_seen >= nblk: a span looks at < 2^32 - 1024 units)
            const uint32_t take = stop < nblk ? stop : nblk;  // units in front of the first irregular one
            if (first_take == kNone)
                first_take = walks == 0 && packets == 0 ? take : 0u;
            // A whole block taken: the next block's header words are asked for NOW — they fly while this block's packets are
            // committed and its continuity counters go through the workgroup's table in turn (below), which is then not
            // on the scan's critical path
            if (take == (uint32_t)kScanBlock) {
                const uint64_t npos = st.pos + (uint64_t)kScanBlock * p.stride;
                const uint64_t nroom = npos < B1 ? B1 - npos : 0;
                const uint32_t nn = nroom > (uint64_t)(kScanBlock - 1) * p.stride ? (uint32_t)kScanBlock
                                                                                 : (uint32_t)((nroom + p.stride - 1) / p.stride);
                const uint64_t nsy = npos + (uint64_t)t * p.stride + p.sync_offset;
                const uint64_t na = (t < nn && nsy + 188 <= p.nbytes) ? (nsy & ~3ull) : 0ull;
                pre_w0 = *reinterpret_cast<const uint32_t *>(p.data + na);
                pre_w1 = *reinterpret_cast<const uint32_t *>(p.data + na + 4);
                pre_pos = npos;
            }
            // A partial block: the walker is next, and behind it a block whose position only the walker knows.  Two dependent trips
            // to memory with nothing to hide them — unless they are asked for NOW, while this block is committed: the walker's
            // window (it starts at the irregular unit), and this lane's header in the block behind, guessed one unit further on
            // with room for the grid to have moved (bytes inserted: up to 3; bytes missing: up to kSpecBefore).  A guess that
            // does not hold costs nothing but its bytes: the block loads as before.
            if (p.lookahead && take < nblk && p.nbytes >= kWalkWindow) {  // (workgroup-uniform; a stream of a few bytes has nothing to look ahead at)
                const uint64_t P = st.pos + (uint64_t)take * p.stride;
                spec_pos = P + p.stride;
                const uint64_t q = spec_pos + (uint64_t)t * p.stride + p.sync_offset;
                const uint64_t a = (q - kSpecBefore) & ~3ull;
                spec_ok = a + 16u <= p.nbytes;
                const uint32_t *src = reinterpret_cast<const uint32_t *>(p.data + (spec_ok ? a : 0ull));
#pragma unroll
                for (int k = 0; k < 4; k++)
                    spec_w[k] = src[k];
                if (wave == 0) {
                    const uint32_t mis = (uint32_t)(((uintptr_t)p.data + P) & 15u);
                    win_pre_len = 0;
                    if (P >= mis && P < p.nbytes) {
                        win_pre_base = P - mis;
                        const uint64_t room = p.nbytes - win_pre_base;
                        win_pre_len = room < kWinAhead ? (uint32_t)room & ~15u : kWinAhead;  // (whole 16-byte pieces only)
                        win_pre = *reinterpret_cast<const uint4 *>(p.data + win_pre_base + (16u * lane < win_pre_len ? 16u * lane : 0u));
                    }
                }
            }
            // (the tables' entry of this lane's PID — a slot, in the slot form — wanted by the count and by the continuity check)
            const uint32_t ix = (t < take && (tei == 0 || ((b3 & 0x10u) != 0 && pid != 0u))) ? idx(pid) : 0u;
            if (t < take) {
                const uint32_t rel = (uint32_t)packets + t;  // packet number within the span (a span counts < 2^32)
                if (tei == 0) {
                    atomicAdd(&s_count[ix], 1u);
                    atomicMin(&s_first[ix], rel);
                    atomicMax(&s_last[ix], rel);
                }
            }
            // ---- continuity counters (xport.c:2872-2889) of the block's committed packets: header byte 3 is loaded already ----
            // A payload-carrying packet (adaptation_field_control & 1) of a PID other than 0 is compared with the PID's previous
            // such packet.  Inside a wave the previous one is found with ballots (one round per PID the wave holds: a handful).
            // Across the waves of the block nothing is serial either: every wave PUBLISHES, per PID it holds, that it does
            // and the counter of its last packet of it (one bit and one nibble per wave in two words the PID gets for the
            // length of this block); behind a barrier the first packet of a PID in a wave finds the nearest earlier wave
            // that holds the PID in those words — or, if there is none, the workgroup's table s_cc (the state at the block's
            // start) — and behind a second barrier the block's last packet of every PID writes the table and gives the
            // words back.  (A first form passed a ticket from wave to wave: 16 dependent LDS round trips per block, which
            // the damaged stream's many short blocks paid in full.)  A PID's first such packet in the SPAN has nothing to
            // be compared with here: it goes on the span's list, and the host links the spans (ts_runtime.cpp).
            {
                const uint32_t cc4 = b3 & 0xfu;
                const bool ccv = t < take && (b3 & 0x10u) != 0 && pid != 0u;
                uint32_t prev = 0;  // the previous packet's counter + 1; 0: not known (yet)
                bool first_in_wave = ccv, last_in_wave = false;
                unsigned long long todo = __ballot(ccv);
                while (todo) {  // (wave-uniform)
                    const int leader = __ffsll((long long)todo) - 1;
                    const uint32_t lp = (uint32_t)__builtin_amdgcn_readlane((int)pid, leader);
                    const bool in = ccv && pid == lp;
                    const unsigned long long m = __ballot(in);
                    const unsigned long long below = m & ((1ull << lane) - 1ull);
                    const int src = below ? 63 - __clzll((long long)below) : (int)lane;
                    const uint32_t pc = (uint32_t)__shfl((int)cc4, src);
                    if (in) {
                        if (below) {
                            prev = pc + 1u;
                            first_in_wave = false;
                        }
                        last_in_wave = ((m >> lane) >> 1) == 0;
                    }
                    todo &= ~m;
                }
                // publish (one lane per wave and PID): the PID's words for this block are found through s_bid
                uint32_t bid = 0;
                if (ccv && last_in_wave) {
                    bid = s_bid[ix];
                    if (!bid) {
                        const uint32_t n = atomicAdd(&s_nid, 1u) + 1u;  // (at most one per publishing lane: <= 1024 a block)
                        const uint32_t old = atomicCAS(&s_bid[ix], 0u, n);
                        bid = old ? old : n;
                    }
                    atomicOr(&s_waves[bid], 1u << wave);
                    atomicOr(&s_lastcc[bid], (unsigned long long)cc4 << (4u * wave));
                }
                __syncthreads();
                bool block_last = false;
                if (ccv && first_in_wave) {
                    const uint32_t id = s_bid[ix];
                    const uint32_t earlier = s_waves[id] & ((1u << wave) - 1u);
                    if (earlier) {
                        const uint32_t wp = 31u - (uint32_t)__clz((int)earlier);
                        prev = (uint32_t)((s_lastcc[id] >> (4u * wp)) & 0xfull) + 1u;
                    } else {
                        const uint32_t last = s_cc[ix];
                        if (last == 0) {
                            ts_cc_entry e;
                            e.pid = (uint16_t)pid;
                            e.first_cc = (uint8_t)cc4;
                            e.last_cc = 0;
                            e.first_rel = (uint32_t)packets + t;
                            cc_list[atomicAdd(&s_ncc, 1u)] = e;
                        } else {
                            prev = last;
                        }
                    }
                }
                if (ccv && last_in_wave)
                    block_last = (s_waves[bid] >> (wave + 1u)) == 0;
                // The block's lines — a discontinuity, and behind it the `skipped 1 bytes` of a read-boundary quirk (about one
                // packet in 4096 of a stream whose packets sit at odd offsets; printed when the stream locks again, i.e. with
                // this packet counted) — take their slots of the event list in stream order: the waves' line counts meet in
                // LDS, ONE atomic reserves the block's slots, every wave starts behind the waves in front of it and its
                // lanes follow in lane order.  A span's lines then sit in the list in the order the reference prints them
                // and the host has nothing to sort.
                const bool ev_disc = ccv && prev != 0 && pid != 0x1fffu && (prev & 0xfu) != cc4;
                const bool ev_quirk = t < take && quirk;
                const unsigned long long md = __ballot(ev_disc), mq = __ballot(ev_quirk);
                if (lane == 0)
                    s_evn[wave] = (uint32_t)(__popcll(md) + __popcll(mq));
                __syncthreads();
                if (block_last) {  // the table moves on; the PID's words are free again
                    s_cc[ix] = (unsigned char)(cc4 + 1u);
                    s_bid[ix] = 0;
                    s_waves[bid] = 0;
                    s_lastcc[bid] = 0;
                }
                if (t == 0)
                    s_nid = 0;
                uint32_t ev_before = 0, ev_total = 0;
#pragma unroll
                for (uint32_t wq = 0; wq < kScanBlock / 64; wq++) {
                    const uint32_t n = s_evn[wq];
                    ev_before += wq < wave ? n : 0u;
                    ev_total += n;
                }
                if (ev_total) {  // (workgroup-uniform; rare)
                    if (wave == 0) {
                        const unsigned int base = ev_take(s_ev, ev_total, lane, p.events, p.event_count, p.event_cap);
                        if (lane == 0)
                            s_evbase = base;
                    }
                    __syncthreads();
                    const unsigned long long lt = (1ull << lane) - 1ull;
                    unsigned int slot = s_evbase + ev_before + (unsigned int)(__popcll(md & lt) + __popcll(mq & lt));
                    const uint64_t number = (uint64_t)((uint32_t)packets + t) + 1;
                    auto put = [&](uint32_t kind, uint64_t skipped, uint32_t info) {
                        if (slot < p.event_cap) {
                            ts_event e;
                            e.skipped = skipped;
                            e.at_rel = number;
                            e.span = span;
                            e.attempt = p.attempt;
                            e.kind = kind;
                            e.info = info;
                            p.events[slot] = e;
                        }
                        slot++;
                    };
                    if (ev_disc)
                        put(TS_EV_DISC, 0, (pid << 8) | (cc4 << 4) | (prev & 0xfu));
                    if (ev_quirk)
                        put(TS_EV_SYNC, 1, 0u);
                }
            }
            packets += take;
            block_packets += take;
            units_seen += nblk;
            st.pos += (uint64_t)take * p.stride;
            walk_next = take < nblk;  // the unit at the new position is the walker's
            continue;
        }
        // ---- the walker: wave 0 across whatever is not on the grid, until the stream is clean again — and the byte
        // where the next sync is due is one (else the next block would only find that out) ----
        walk_next = false;
        walks++;
        // The full-table form gives a damaged stream up: one workgroup per CU has nothing to run while a damaged spot's three
        // dependent trips to memory are under way; the slot form (two spans per CU) has.  More than one walk in 6144
        // packets, abort_walks walks into the span: every span stops, the host scans again in the slot form.
        if (p.abort_walks && walks >= p.abort_walks && (uint64_t)walks * 6144u > packets) {  // (workgroup-uniform)
            if (t == 0)
                atomicOr(p.event_count + 2, 1u);
            break;
        }
        __syncthreads();  // every thread has committed its packets and read s_stop
        if (wave == 0) {
            DevWalk w;
            w.data = p.data;
            w.s_count = s_count;
            w.s_first = s_first;
            w.s_last = s_last;
            w.packets = packets;
            w.events = p.events;
            w.event_count = p.event_count;
            w.event_cap = p.event_cap;
            w.span = span;
            w.attempt = p.attempt;
            w.lane = lane;
            w.win = s_window;
            w.nbytes = p.nbytes;
            w.win_base = 0;
            w.win_len = 0;  // (nothing in the window yet: the first byte asked for brings it in)
            if (win_pre_len && st.pos >= win_pre_base && st.pos < win_pre_base + win_pre_len) {  // ... unless it was asked for already
                if (16u * lane < win_pre_len)
                    *reinterpret_cast<uint4 *>(s_window + 16u * lane) = win_pre;
                __builtin_amdgcn_wave_barrier();
                w.win_base = win_pre_base;
                w.win_len = win_pre_len;
            }
            win_pre_len = 0;
            w.g_count = nullptr;
            w.g_first = w.g_last = nullptr;
            w.abs0 = 0;
            w.stop_at = TS_NO_ENTRY;
            w.quiet = 0;
            w.s_cc = s_cc;
            w.s_ncc = &s_ncc;
            w.cc_list = cc_list;
            w.s_ev = s_ev;
            w.s_slot = SLOTS ? s_slot : nullptr;
            w.s_nslots = &s_nslots;
            w.s_over = &s_over;
            w.s_slot_pid = s_slot_pid;
            w.slot_limit = p.slot_limit;
            ts_walk_state s2 = st;
            uint64_t rs = run_start;
            for (;;) {
                const uint64_t before = s2.pos;
                const bool owed = s2.skipped != 0;
                if (!dev_walk_step(&s2, &w, p.nbytes, 1))
                    break;  // the stream ended in front of the next packet (s2.pos == nbytes)
                if (owed || s2.pos != before + p.stride)  // (bytes skipped in front of this packet, or it ended early: a new run begins
                    rs = s2.pos >= p.stride ? s2.pos - p.stride : s2.pos;  // with it — at the unit it fills if it is a whole one)
                if (s2.pos >= p.nbytes)
                    break;
                if (walk_is_clean(s2) && (s2.pos >= B1 || s2.pos + p.sync_offset >= p.nbytes || dev_byte(&w, s2.pos + p.sync_offset) == 0x47u))
                    break;  // (behind the span's end the next span takes over, grid or not: the merge sorts that out)
            }
            if (lane == 0) {
                s_st = s2;
                s_packets = w.packets;
                s_run_start = rs;
                s_stop = kNone;
            }
        }
        __syncthreads();
        st = s_st;
        packets = s_packets;
        run_start = s_run_start;
    }
    __syncthreads();
    if (t == 0) {
        s_st = st;
        s_packets = packets;
        s_block_packets = block_packets;
        s_walks = walks;
    }
    __syncthreads();
    // ---- what the span leaves behind: its PIDs as a list, and its record ----
    ts_wg_entry *list = p.lists + (size_t)span * TS_PIDS;
    const uint32_t nused = SLOTS ? (s_nslots < p.slot_limit ? s_nslots : p.slot_limit) : (uint32_t)TS_PIDS;
    for (uint32_t k = t; k < nused; k += kScanBlock) {
        if (s_count[k]) {
            const uint32_t at = atomicAdd(&s_entries, 1u);
            ts_wg_entry e;
            e.pid = SLOTS ? (uint32_t)s_slot_pid[k] : k;
            e.count = s_count[k];
            e.first = s_first[k];
            e.last = s_last[k];
            list[at] = e;
        }
    }
    for (uint32_t k = t; k < s_ncc; k += kScanBlock)  // the list's PIDs: their last counter in this span
        cc_list[k].last_cc = (uint8_t)(s_cc[SLOTS ? slot_peek(s_slot, cc_list[k].pid) - 1u : (uint32_t)cc_list[k].pid] - 1u);
    __syncthreads();
    if (t == 0) {
        ts_span_rec r;
        r.ncc = s_ncc;
        r.pad = SLOTS ? s_over : 0u;  // 1: more PIDs than slots — this span's numbers are not to be used
        if (SLOTS && s_over)
            atomicOr(p.event_count + 1, 1u);  // (the word behind the event counter: the scan's "a span overflowed")
        r.entry = entry_pos;
        r.exit_pos = s_st.pos;
        r.exit_skipped = s_st.skipped;
        r.exit_stale_af = s_st.stale_af;
        r.exit_extra = s_st.extra_pending;
        r.packets = s_packets;
        r.block_packets = s_block_packets;
        r.walks = s_walks;
        r.nlist = s_entries;
        r.attempt = p.attempt;
        r.explicit_entry = p.explicit_entry;
        r.first_take = first_take == kNone ? 0u : first_take;  // (the same in every thread)
        r.pad3 = 0;
        r.exit_run_start = run_start;
        p.recs[span] = r;
    }
}

// The chain arrives BEHIND the entry a span assumed — by whole packets, on the span's grid, owing nothing: the span in front
// ended un-clean on the boundary (an adaptation field that ran over, bytes skipped) and, by the reference's rules, took the
// first packet(s) that start in this span along before it was clean.  The span counted the same packets, plain ones its first
// block committed one lane each (none of them a read-boundary quirk: those print a line of their own): they are counted once
// — this span's numbering starts that many packets earlier, the merge takes one off their PIDs' counts, the host drops their
// lines and continuity checks here — instead of the whole span being scanned again by one workgroup (0.7-1.0 ms; 20 of the 26
// spans scanned again in 40 damaged streams were this case, all of them by one packet).  Returns that number of packets, or 0.
// `run_start`: ts_span_rec::exit_run_start of the span the chain comes out of (TS_NO_ENTRY: not known — a state the host handed in).
constexpr uint32_t kOverlapMax = 8;
__device__ uint32_t ts_overlap_packets(const ts_scan_params &p, const ts_span_rec &r, const ts_walk_state &cur, uint64_t run_start, uint64_t B1)
{
    if (!p.overlap || r.explicit_entry || r.entry == TS_NO_ENTRY || cur.pos <= r.entry || cur.pos >= B1)
        return 0u;
    // (the chain's last packets must BE this span's first: consumed back to back from a unit start on the span's grid at or in
    // front of its entry — a state machine that came through here on a bogus packet's grid and merely ended on this one has
    // counted other packets)
    if (run_start == TS_NO_ENTRY || run_start > r.entry || (r.entry - run_start) % p.stride != 0)
        return 0u;
    if (cur.skipped != 0 || cur.stale_af != 0 || (cur.hdmv && cur.extra_pending != 4u))
        return 0u;
    const uint64_t d = cur.pos - r.entry;
    if (d % p.stride != 0 || d / p.stride > kOverlapMax)
        return 0u;
    const uint32_t m = (uint32_t)(d / p.stride);
    if (m > r.first_take || (uint64_t)m > r.packets)
        return 0u;
    for (uint32_t i = 0; i < m; i++) {
        const uint64_t sy = r.entry + (uint64_t)i * p.stride + p.sync_offset;
        if (((sy + 187) & (TS_READ_CHUNK - 1)) == 0)
            return 0u;  // (a read-boundary quirk candidate: its own line, its own rules)
    }
    return m;
}

// The walker from the chain's state `cur` up to span k's entry (wave 0, all lanes in step): does it get there — on the very
// byte, owing nothing?  quiet: nothing is applied (a dry run: what a bridge counts and reports must not be applied unless it
// arrives); otherwise its packets go into the stream-wide tables, numbered from `base`, and its lines into the event list.
__device__ bool bridge_walk(const ts_scan_params &p, uint32_t k, const ts_span_rec &r, const ts_walk_state &cur, uint64_t base,
                            uint32_t lane, unsigned char *win, bool quiet, uint32_t *g_count, unsigned long long *g_first,
                            unsigned long long *g_last, uint64_t *packets_out)
{
    DevWalk w;
    w.data = p.data;
    w.s_count = w.s_first = w.s_last = nullptr;
    w.events = p.events;
    w.event_count = p.event_count;
    w.event_cap = p.event_cap;
    w.span = k;
    w.attempt = r.attempt | kBridgeEvent;
    w.lane = lane;
    w.win = win;
    w.nbytes = p.nbytes;
    w.g_count = g_count;
    w.g_first = g_first;
    w.g_last = g_last;
    w.abs0 = base;
    w.s_cc = nullptr;  // (a bridge's packets are reported one by one: the host checks their continuity)
    w.s_ncc = nullptr;
    w.cc_list = nullptr;
    w.s_ev = nullptr;
    w.s_slot = nullptr;
    w.s_nslots = w.s_over = nullptr;
    w.s_slot_pid = nullptr;
    w.slot_limit = 0;
    w.stop_at = r.entry + p.sync_offset;  // the span's first sync byte: where its own findings begin
    w.quiet = quiet ? 1u : 0u;
    w.packets = 0;
    w.win_base = 0;
    w.win_len = 0;
    ts_walk_state s2 = cur;
    bool arrived = false;
    for (uint32_t steps = 0; steps < kBridgeSteps; steps++) {
        const int rc = dev_walk_step(&s2, &w, p.nbytes, 1);
        if (rc == 2) {  // the search ended on that very byte; what an earlier packet still owed would
            arrived = s2.stale_af == 0;  // change how the span's first packet is taken: not the same stream
            break;
        }
        if (rc == 0 || s2.pos > w.stop_at)
            break;
    }
    // the line the reference prints when it locks there (xport.c:4324-4327): the span, which started clean, did not know of
    // the bytes skipped in front of it
    if (arrived && !quiet && s2.skipped)
        dev_event(&w, s2.skipped, w.packets);
    *packets_out = w.packets;
    return arrived;
}

// Every span's bridge at once, one wave per span: the chain coming from the exit of the span in front — the usual case; the
// merge kernel falls back to walking where it comes from somewhere else.  Dry runs only.
__global__ __launch_bounds__(64) void ts_bridge_kernel(const ts_scan_params p, uint32_t from_span, ts_walk_state cur0)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_window[kWalkWindow];
    const uint32_t k = from_span + blockIdx.x;
    const ts_span_rec r = p.recs[k];
    ts_walk_state cur = cur0;  // (span from_span: the state the chain arrived with)
    bool explicit_ok = r.explicit_entry != 0;
    if (k > from_span) {
        const ts_span_rec q = p.recs[k - 1];
        cur.pos = q.exit_pos;
        cur.skipped = q.exit_skipped;
        cur.stale_af = q.exit_stale_af;
        cur.extra_pending = q.exit_extra;
        explicit_ok = false;
    }
    const uint64_t B1 = (k + 1 == p.nspans_total || (uint64_t)(k + 1) * p.span_bytes > p.nbytes) ? p.nbytes : (uint64_t)(k + 1) * p.span_bytes;
    uint32_t state = 0;
    uint64_t packets = 0;
    if (cur.pos < B1) {  // (else: the chain is past this span already — the merge's business)
        const bool clean = cur.skipped == 0 && cur.stale_af == 0 && (!cur.hdmv || cur.extra_pending == 4u);
        if (r.entry == cur.pos && (clean || explicit_ok))
            state = 1;
        else if (!r.explicit_entry && r.entry != TS_NO_ENTRY && cur.pos <= r.entry && r.entry - cur.pos <= kBridgeMax)
            state = bridge_walk(p, k, r, cur, 0, threadIdx.x, s_window, true, nullptr, nullptr, nullptr, &packets) ? 2u : 3u;
        else if ((packets = ts_overlap_packets(p, r, cur, k > from_span ? p.recs[k - 1].exit_run_start : TS_NO_ENTRY, B1)) != 0)
            state = 4;
        else
            state = 3;
    }
    if (threadIdx.x == 0) {
        ts_bridge_rec b;
        b.packets = packets;
        b.state = state;
        b.pad = 0;
        p.bridges[k] = b;
    }
}

// One workgroup per span from `from_span` on.  Every workgroup walks the chain of records for itself (a few hundred
// entries): a span is taken if the chain arrives, clean, exactly where the span started (or the span was launched from
// the very state the chain arrived with); a span the chain has already passed (the span in front ran on across it: a
// stretch without a grid) is skipped; the first span that fits neither ends the valid part.  Taken spans fold their
// lists into the stream-wide tables with their first packet's stream-wide number as base (count: add, first: min over
// a table that starts at all-ones, last: max — order-independent, so the spans go in parallel).
__global__ __launch_bounds__(kMergeBlock) void ts_merge_kernel(const ts_scan_params p, uint32_t from_span, uint64_t packet_base,
                                                               ts_walk_state cur0, uint32_t *__restrict__ g_count,
                                                               unsigned long long *__restrict__ g_first,
                                                               unsigned long long *__restrict__ g_last,
                                                               unsigned long long *__restrict__ span_base,
                                                               unsigned long long *__restrict__ span_bridge_base,
                                                               uint32_t *__restrict__ span_attempt,
                                                               ts_merge_out *__restrict__ out, ts_span_out *__restrict__ span_out)
{
    __shared__ uint32_t s_taken;
    __shared__ unsigned long long s_base, s_bridge_base;
    __shared__ __attribute__((aligned(16))) unsigned char s_window[kWalkWindow];  // (the bridges' view of the stream)
    __shared__ ts_span_rec s_recs[TS_MAX_SPANS];  // (the chain walk is a serial loop: out of LDS, not out of HBM)
    const uint32_t t = threadIdx.x;
    const uint32_t me = from_span + blockIdx.x;
    __shared__ uint32_t s_broken, s_bad, s_dup;
    __shared__ unsigned long long s_sum_before, s_sum_all, s_sum_block, s_sum_walks, s_my_bridge;
    for (uint32_t k = from_span + t; k < p.nspans_total; k += kMergeBlock)
        s_recs[k] = p.recs[k];
    if (t == 0) {
        s_broken = 0;
        s_bad = p.nspans_total;
        s_dup = 0;
        s_my_bridge = 0;
        s_sum_before = s_sum_all = s_sum_block = s_sum_walks = 0;
    }
    __syncthreads();
    // In parallel, link by link: span k is reached if it started exactly where — clean — the one in front of it ended,
    // inside its own range (the common case: nothing has to be walked), or if ts_bridge_kernel found a bridge from there to
    // its entry (a damaged stream: damage at a span's beginning, which its speculated entry skipped); the first span that is
    // reached neither way ends the valid part.  A span's base is then a sum over the spans in front of it.  Only a link the
    // bridge kernel did not work out (the chain already past a span: a stretch without a grid) sends wave 0 along the chain.
    auto link = [&](uint32_t k, ts_walk_state &prev, unsigned long long &bridge_packets) -> uint32_t {
        const ts_span_rec r = s_recs[k];
        prev = cur0;
        bool explicit_ok = r.explicit_entry != 0;
        if (k > from_span) {
            const ts_span_rec q = s_recs[k - 1];
            prev.pos = q.exit_pos;
            prev.skipped = q.exit_skipped;
            prev.stale_af = q.exit_stale_af;
            prev.extra_pending = q.exit_extra;
            explicit_ok = false;
        }
        const uint64_t B1 = (k + 1 == p.nspans_total || (uint64_t)(k + 1) * p.span_bytes > p.nbytes) ? p.nbytes
                                                                                                    : (uint64_t)(k + 1) * p.span_bytes;
        const bool clean = prev.skipped == 0 && prev.stale_af == 0 && (!prev.hdmv || prev.extra_pending == 4u);
        bridge_packets = 0;
        if (prev.pos < B1 && r.entry == prev.pos && (clean || explicit_ok))
            return 1u;
        if (!p.bridges) {  // (no bridge kernel in front: only what needs no walk is decided here)
            const uint32_t m = ts_overlap_packets(p, r, prev, k > from_span ? s_recs[k - 1].exit_run_start : TS_NO_ENTRY, B1);
            if (m)
                bridge_packets = 0ull - m;  // (the span's numbering starts m packets EARLIER: the chain has counted them)
            return m ? 4u : 0u;
        }
        const ts_bridge_rec b = p.bridges[k];
        if (b.state == 2u)
            bridge_packets = b.packets;
        else if (b.state == 4u)
            bridge_packets = 0ull - b.packets;
        return b.state == 1u ? 0u : b.state;  // (1 cannot be: the bridge kernel saw the same records)
    };
    for (uint32_t k = from_span + t; k < p.nspans_total; k += kMergeBlock) {
        ts_walk_state prev;
        unsigned long long bp;
        const uint32_t st = link(k, prev, bp);
        if (st == 0u)
            s_broken = 1;
        else if (st == 3u)
            atomicMin(&s_bad, k);
    }
    __syncthreads();
    if (!s_broken) {
        const uint32_t valid = s_bad;  // spans [from_span, valid) are the chain
        unsigned long long before = 0, all = 0, blk = 0, wk = 0;
        for (uint32_t k = from_span + t; k < valid; k += kMergeBlock) {
            const ts_span_rec r = s_recs[k];
            ts_walk_state prev;
            unsigned long long bp;
            const uint32_t st = link(k, prev, bp);
            all += r.packets + bp;
            blk += r.block_packets;
            wk += r.walks + (st == 2u ? 1u : 0u);
            if (k < me)
                before += r.packets + bp;
            if (k == me) {
                s_my_bridge = bp;
                s_dup = st == 4u ? (uint32_t)(0ull - bp) : 0u;
            }
        }
        atomicAdd(&s_sum_before, before);
        atomicAdd(&s_sum_all, all);
        atomicAdd(&s_sum_block, blk);
        atomicAdd(&s_sum_walks, wk);
        __syncthreads();
        if (t == 0) {
            s_taken = me < valid ? s_recs[me].attempt : 0u;
            s_bridge_base = packet_base + s_sum_before;
            s_base = s_bridge_base + s_my_bridge;
            if (blockIdx.x == 0) {
                out->valid_upto = valid;
                out->pad = p.event_count[2];  // the full-table form gave a damaged stream up (ts_scan_params::abort_walks)
                out->packets = packet_base + s_sum_all;
                ts_walk_state cur = cur0;
                if (valid > from_span) {
                    const ts_span_rec last = s_recs[valid - 1];
                    cur.pos = last.exit_pos;
                    cur.skipped = last.exit_skipped;
                    cur.stale_af = last.exit_stale_af;
                    cur.extra_pending = last.exit_extra;
                }
                out->cur = cur;
                out->block_packets = s_sum_block;
                out->walks = s_sum_walks;
                out->events = *p.event_count;
                out->pad2 = p.event_count[1];  // a span of the slot form met more PIDs than it has slots
            }
        }
        __syncthreads();
        // the bridge in front of THIS span, for real: its packets into the stream-wide tables, its lines into the event list
        if (t < 64u && me < valid) {
            ts_walk_state prev;
            unsigned long long bp;
            if (link(me, prev, bp) == 2u) {
                uint64_t walked = 0;
                (void)bridge_walk(p, me, s_recs[me], prev, s_bridge_base, t, s_window, false, g_count, g_first, g_last, &walked);
            }
        }
    } else if (t < 64u) {
        // The chain does not hold everywhere: wave 0 goes along it, all lanes in step.  Where it arrives IN FRONT of the
        // place a span assumed — damage at the span's beginning, which its speculated entry skipped — the walker BRIDGES
        // the gap: from the chain's state up to the span's entry; if it gets there clean, the span's own findings stand,
        // numbered behind the bridge's packets, and only the bridge was walked (a dry run first: what a bridge counts and
        // reports must not be applied unless it arrives).  Every workgroup walks the bridges in front of its span (it
        // needs its base); the one that owns the span applies them.  Only where that fails too (the chain arrives BEHIND
        // the span's entry, or dirty, or nowhere near) does the host launch the span again.
        ts_walk_state cur = cur0;
        uint64_t base = packet_base, blockp = 0, walks = 0;
        uint32_t k = from_span, taken_me = 0, dup_me = 0;
        uint64_t base_me = 0, bridge_me = 0;
        uint64_t chain_run = TS_NO_ENTRY;  // ts_span_rec::exit_run_start of the span the chain last came out of (ts_overlap_packets)
        for (; k < p.nspans_total; k++) {
            const uint64_t B1 = (k + 1 == p.nspans_total || (uint64_t)(k + 1) * p.span_bytes > p.nbytes) ? p.nbytes
                                                                                                        : (uint64_t)(k + 1) * p.span_bytes;
            if (cur.pos >= B1) {  // the chain is past this span already
                if (k == me)
                    taken_me = 0;
                continue;
            }
            const ts_span_rec r = s_recs[k];
            const bool clean = cur.skipped == 0 && cur.stale_af == 0 && (!cur.hdmv || cur.extra_pending == 4u);
            bool fits = r.entry == cur.pos && (clean || (r.explicit_entry && k == from_span));
            uint64_t bridge_base = base, w_packets = 0;
            if (!fits && !r.explicit_entry && r.entry != TS_NO_ENTRY && cur.pos <= r.entry && r.entry - cur.pos <= kBridgeMax) {
                // (a dry run first: what a bridge counts and reports must not be applied unless it arrives)
                uint64_t bridge_packets = 0;
                if (bridge_walk(p, k, r, cur, base, t, s_window, true, g_count, g_first, g_last, &bridge_packets)) {
                    fits = true;
                    if (k == me)
                        (void)bridge_walk(p, k, r, cur, base, t, s_window, false, g_count, g_first, g_last, &bridge_packets);
                }
                w_packets = bridge_packets;
                if (fits) {
                    base += w_packets;
                    walks++;
                }
            }
            if (!fits) {  // (... or behind it by whole packets the span in front took along: counted once)
                const uint32_t m = ts_overlap_packets(p, r, cur, chain_run, B1);
                if (m) {
                    fits = true;
                    base -= m;
                    if (k == me)
                        dup_me = m;
                }
            }
            if (!fits)
                break;
            if (k == me) {
                taken_me = r.attempt;
                base_me = base;
                bridge_me = bridge_base;
            }
            base += r.packets;
            blockp += r.block_packets;
            walks += r.walks;
            cur.pos = r.exit_pos;
            cur.skipped = r.exit_skipped;
            cur.stale_af = r.exit_stale_af;
            cur.extra_pending = r.exit_extra;
            chain_run = r.exit_run_start;
        }
        if (t == 0) {
            s_taken = k > me ? taken_me : 0u;  // (k <= me: the chain broke in front of this span)
            s_base = base_me;
            s_bridge_base = bridge_me;
            s_dup = dup_me;
            if (blockIdx.x == 0) {
                out->valid_upto = k;
                out->pad = p.event_count[2];
                out->packets = base;
                out->cur = cur;
                out->block_packets = blockp;
                out->walks = walks;
                out->pad2 = p.event_count[1];
            }
        }
    }
    __syncthreads();
    const uint32_t taken = s_taken;
    const uint64_t base = s_base;
    if (t == 0) {
        span_attempt[me] = taken;
        span_base[me] = base;
        span_bridge_base[me] = s_bridge_base;
    }
    // what the host needs of this span — where its packets are numbered from, and the head of its continuity list — in
    // the one block the scan's single wait brings over
    {
        const uint32_t ncc = taken ? s_recs[me].ncc : 0u;
        ts_span_out *so = span_out + me;
        if (t == 0) {
            so->base = base;
            so->bridge_base = s_bridge_base;
            so->attempt = taken;
            so->ncc = ncc;
            so->dup = taken ? s_dup : 0u;
            so->pad = 0;
        }
        const ts_cc_entry *cl = p.cc_lists + (size_t)me * TS_PIDS;
        for (uint32_t k = t; k < ncc && k < TS_CC_OUT; k += kMergeBlock)
            so->cc[k] = cl[k];
    }
    if (!taken)
        return;  // (workgroup-uniform)
    const ts_wg_entry *list = p.lists + (size_t)me * TS_PIDS;
    const uint32_t n = s_recs[me].nlist;
    for (uint32_t k = t; k < n; k += kMergeBlock) {
        const ts_wg_entry e = list[k];
        atomicAdd(&g_count[e.pid], e.count);
        atomicMin(&g_first[e.pid], base + e.first + 1);
        atomicMax(&g_last[e.pid], base + e.last + 1);
    }
    // (packets the span in front counted already: once is enough — their numbers are the same from either side, so first / last stand)
    if (t < s_dup) {
        const uint64_t sy = s_recs[me].entry + (uint64_t)t * p.stride + p.sync_offset;
        const uint32_t b1 = p.data[sy + 1], b2 = p.data[sy + 2];
        if ((b1 & 0x80u) == 0)
            atomicSub(&g_count[((b1 & 0x1fu) << 8) | b2], 1u);
    }
}

__global__ __launch_bounds__(256) void ts_generate_kernel(unsigned char *__restrict__ out, uint64_t nunits, uint32_t unit,
                                                           uint64_t seed, int hdmv)
{
    // one thread per 4 bytes (units are multiples of 4 bytes)
    const uint64_t words = nunits * (unit / 4);
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < words; w += (uint64_t)gridDim.x * 256) {
        const uint64_t k = w / (unit / 4);
        const uint32_t i = (uint32_t)(w % (unit / 4)) * 4;
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++)
            v |= (uint32_t)ts_synth_byte(seed, k, i + b, hdmv) << (8 * b);
        reinterpret_cast<uint32_t *>(out)[w] = v;
    }
}

__global__ __launch_bounds__(256) void ts_generate_damaged_kernel(unsigned char *__restrict__ out, uint64_t nbytes, uint64_t period,
                                                                   uint64_t seed)
{
    const uint64_t words = (nbytes + 3) / 4;  // (the buffer has slack behind nbytes)
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < words; w += (uint64_t)gridDim.x * 256) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++)
            v |= (uint32_t)(4 * w + b < nbytes ? ts_synth_damaged_byte(seed, period, 4 * w + b) : 0) << (8 * b);
        reinterpret_cast<uint32_t *>(out)[w] = v;
    }
}

void ts_launch_generate_damaged(hipStream_t st, void *out, uint64_t nbytes, uint64_t period, uint64_t seed)
{
    const uint64_t words = (nbytes + 3) / 4;
    const int blocks = (int)((words + 255) / 256 < 16384 ? (words + 255) / 256 : 16384);
    if (blocks > 0)
        hipLaunchKernelGGL(ts_generate_damaged_kernel, dim3(blocks), dim3(256), 0, st, (unsigned char *)out, nbytes, period, seed);
}

// what a scan starts from, in ONE launch (four memsets were four dispatches): empty tables — `first` is a min table and
// starts at all-ones — no events, no span taken
__global__ __launch_bounds__(256) void ts_reset_kernel(uint32_t *__restrict__ g_count, unsigned long long *__restrict__ g_first,
                                                        unsigned long long *__restrict__ g_last, unsigned int *__restrict__ event_count,
                                                        uint32_t *__restrict__ span_attempt, uint32_t nspans)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < TS_PIDS) {
        g_count[i] = 0;
        g_first[i] = ~0ull;
        g_last[i] = 0;
    }
    if (i < nspans)
        span_attempt[i] = 0;
    if (i == 0)
        event_count[0] = event_count[1] = event_count[2] = 0;  // (events wanted; "a span overflowed its PID slots"; "a damaged stream")
}

void ts_launch_reset(hipStream_t st, uint32_t *g_count, unsigned long long *g_first, unsigned long long *g_last,
                     unsigned int *event_count, uint32_t *span_attempt, uint32_t nspans)
{
    const uint32_t n = nspans > TS_PIDS ? nspans : TS_PIDS;
    hipLaunchKernelGGL(ts_reset_kernel, dim3((n + 255) / 256), dim3(256), 0, st, g_count, g_first, g_last, event_count, span_attempt,
                       nspans);
}

static size_t scan_lds(bool slots)
{
    return 3 * (slots ? (size_t)kSlots + 1 : (size_t)TS_PIDS) * sizeof(uint32_t);
}

void ts_kernels_prepare_device(void)  // function attributes belong to the current device
{
    (void)hipFuncSetAttribute((const void *)ts_scan_kernel<kScanBlock, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scan_lds(false));
    (void)hipFuncSetAttribute((const void *)ts_scan_kernel<kSlotBlock, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scan_lds(true));
}

void ts_launch_scan(hipStream_t st, int blocks, const ts_scan_params &p)
{
    if (p.slots)
        hipLaunchKernelGGL((ts_scan_kernel<kSlotBlock, true>), dim3(blocks), dim3(kSlotBlock), scan_lds(true), st, p);
    else
        hipLaunchKernelGGL((ts_scan_kernel<kScanBlock, false>), dim3(blocks), dim3(kScanBlock), scan_lds(false), st, p);
}

void ts_launch_merge(hipStream_t st, const ts_scan_params &p, uint32_t from_span, uint64_t packet_base, const ts_walk_state &cur,
                     uint32_t *g_count, unsigned long long *g_first, unsigned long long *g_last, unsigned long long *span_base,
                     unsigned long long *span_bridge_base, uint32_t *span_attempt, ts_merge_out *out, ts_span_out *span_out)
{
    hipLaunchKernelGGL(ts_merge_kernel, dim3(p.nspans_total - from_span), dim3(kMergeBlock), 0, st, p, from_span, packet_base, cur,
                       g_count, g_first, g_last, span_base, span_bridge_base, span_attempt, out, span_out);
}

void ts_launch_bridges(hipStream_t st, const ts_scan_params &p, uint32_t from_span, const ts_walk_state &cur)
{
    if (p.bridges && p.nspans_total > from_span)
        hipLaunchKernelGGL(ts_bridge_kernel, dim3(p.nspans_total - from_span), dim3(64), 0, st, p, from_span, cur);
}

void ts_launch_generate(hipStream_t st, void *out, uint64_t nunits, uint32_t unit, uint64_t seed, int hdmv)
{
    const uint64_t words = nunits * (unit / 4);
    const int blocks = (int)((words + 255) / 256 < 16384 ? (words + 255) / 256 : 16384);
    if (blocks > 0)
        hipLaunchKernelGGL(ts_generate_kernel, dim3(blocks), dim3(256), 0, st, (unsigned char *)out, nunits, unit, seed, hdmv);
}
