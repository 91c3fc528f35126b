// ts_kernels.h — shared between ts_kernels.hip and ts_runtime.cpp.  Internal: not part of the C ABI.
#ifndef TS_KERNELS_H
#define TS_KERNELS_H

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "ts_hip.h"

// one PID of one span: packets, first and last packet number (relative to the span's first packet)
struct ts_wg_entry {
    uint32_t pid, count, first, last;
};

// What a span leaves behind: where it started, the walker's state where it ended, what it counted.  The chain of spans
// is valid where every span started exactly where the one in front of it ended (ts_merge_kernel).
struct ts_span_rec {
    uint64_t entry;          // file offset the span started from (TS_NO_ENTRY: no packet grid found at its start)
    uint64_t exit_pos;       // walker state behind its last packet
    uint64_t exit_skipped;
    uint32_t exit_stale_af, exit_extra;
    uint64_t packets;        // packets it counted ...
    uint64_t block_packets;  // ... of them by the one-lane-per-packet blocks (the rest: the walker)
    uint32_t walks;          // times the walker took over
    uint32_t nlist;          // PIDs in its list
    uint32_t attempt;        // id of the launch that wrote this record (its events carry the same)
    uint32_t explicit_entry; // 1: started from a walker state handed in by the host, clean or not
    uint32_t ncc, pad;       // entries in its continuity list
    uint32_t first_take;     // packets its very first block committed (0: it began with the walker): as many of its first packets are
    uint32_t pad3;           // plain packets on its grid, one lane each — what ts_overlap_packets may hand back to the span in front
    uint64_t exit_run_start; // where the run of packets it consumed back to back (each starting where the one in front ended, whole)
                             // up to its exit began: a span behind whose entry lies on that run was entered by THIS span's packets
};
#define TS_NO_ENTRY 0xFFFFFFFFFFFFFFFFull
#define TS_EVENT_BRIDGE 0x80000000u /* ts_event::attempt: written by a bridge of ts_merge_kernel */
#define TS_MAX_SPANS 1024 /* spans per scan (one to three per CU; ts_merge_kernel keeps their records in LDS) */

// a line of the report, before the span's packets have their stream-wide numbers
//   kind 0  `Transport Sync Error`: at_rel = packets the span had counted at that moment
//   kind 1  `Discontinuity!` found inside a span: at_rel = the packet's own number within the span (1-based); info = pid << 8 |
//           received << 4 | expected
//   kind 2  a payload-carrying packet a BRIDGE of the merge kernel walked (info = pid << 8 | counter << 4): its continuity is
//           the host's to check, which links the spans' first and last counters in stream order anyway (ts_runtime.cpp)
struct ts_event {
    uint64_t skipped;
    uint64_t at_rel;
    uint32_t span, attempt;
    uint32_t kind, info;
};
#define TS_EV_SYNC 0u
#define TS_EV_DISC 1u
#define TS_EV_BRIDGE_CC 2u

// continuity counters (xport.c:2872-2889) per span: one entry per PID that had a payload-carrying packet in the span
struct ts_cc_entry {
    uint16_t pid;
    uint8_t first_cc, last_cc;  // counter of the PID's first / last such packet in the span
    uint32_t first_rel;         // the first one's number within the span (0-based)
};
#define TS_CC_OUT 32u /* entries per span that travel to the host with the scan's one wait (more: a second copy) */

// what the host needs of every span once the chain is merged, in one block of memory (one D2H copy)
struct ts_span_out {
    unsigned long long base, bridge_base;  // stream-wide number of the span's (of its bridge's) first packet - 1
    uint32_t attempt;                      // the attempt whose record was taken (0: the span was not taken)
    uint32_t ncc;                          // entries in its continuity list
    uint32_t dup, pad;                     // its first `dup` packets were the span's in front already (that span ended un-clean on the
                                           // boundary and took them along): counted once, their lines and continuity checks dropped here
    ts_cc_entry cc[TS_CC_OUT];
};

// Does the chain, arriving from the exit of span k - 1, reach span k?  Worked out for every span at once (ts_bridge_kernel)
// in front of the merge, whose workgroups would otherwise each walk every bridge in front of their span themselves.
struct ts_bridge_rec {
    unsigned long long packets;  // state 2: packets the bridge counts in front of the span
    uint32_t state;              // 0 not worked out (the merge walks itself); 1 the span starts where the one in front ended;
                                 // 2 a bridge gets there; 3 nothing does (the chain ends in front of this span);
                                 // 4 the chain stands `packets` whole packets BEHIND the span's entry, on its grid (ts_overlap_packets)
    uint32_t pad;
};

struct ts_scan_params {
    const unsigned char *data;  // the stream, file offset 0 at data[0]
    uint64_t nbytes;
    uint64_t span_bytes;        // span k = the packets that start in bytes [k, k + 1) * span_bytes
    uint32_t first_span, nspans_total;
    uint32_t stride;            // 188, or 192 (HDMV)
    uint32_t sync_offset;       // 0, or 4 (HDMV: behind the tp_extra_header)
    uint32_t hdmv;
    uint32_t attempt;
    uint32_t explicit_entry;    // 1: a launch of ONE span from `entry` (the state the chain arrived with)
    uint32_t quirk_events;      // 0: every read-boundary quirk goes to the walker (tests)
    uint32_t slots;             // 1: the slot form of the scan kernel (512 threads, per-slot tables: two spans per CU); 0: full tables
    uint32_t slot_limit;        // slots a span may hand out (0: all it has; tests: few, to force the full-table scan)
    uint32_t abort_walks;       // full-table form: a span that has walked this often, more than once per 6144 packets, stops the
                                // scan (every span) — the stream is damaged and the slot form's (0: never)
    uint32_t lookahead;         // 1: at a partial block, ask for the walker's window and for the headers of the block behind the damage
                                // before the block is committed (TS_SCAN_LOOKAHEAD=0: afterwards, one trip to memory at a time)
    uint32_t overlap;           // 1: a span whose first packets the span in front took along is kept (ts_overlap_packets);
                                // TS_SCAN_OVERLAP=0: it is scanned again from the chain's state, as before round 5 (tests)
    ts_walk_state entry;
    ts_wg_entry *lists;         // per span: up to TS_PIDS entries
    ts_span_rec *recs;          // per span
    ts_cc_entry *cc_lists;      // per span: up to TS_PIDS entries
    ts_event *events;           // one list for the launch(es) of a scan, slots handed out by an atomic counter
    uint32_t event_cap;
    ts_bridge_rec *bridges;     // per span (may be null: the merge works every bridge out itself)
    unsigned int *event_count;  // [0] events wanted so far (may run past event_cap: the host then repeats the scan with more room);
                                // [1] a span of the slot form met more PIDs than it has slots (the host scans again, full tables)
                                // [2] the full-table form gave a damaged stream up (the host scans again, slot form)
};

// what ts_merge_kernel tells the host
struct ts_merge_out {
    uint32_t valid_upto;        // first span the chain did NOT reach validly (== nspans_total: done)
    uint32_t pad;               // != 0: the full-table form gave the (damaged) stream up — nothing of this scan is to be used
    uint64_t packets;           // packets of the valid chain (stream-wide packet_counter so far)
    ts_walk_state cur;          // the walker state the chain arrived with in front of span `valid_upto`
    uint64_t block_packets;
    uint64_t walks;
    uint32_t events;            // events the scan's launches have wanted so far (> event_cap: the list overflowed)
    uint32_t pad2;              // != 0: a span overflowed its PID slots — nothing of this scan is to be used
};

void ts_kernels_prepare_device(void);
void ts_launch_scan(hipStream_t st, int blocks, const ts_scan_params &p);
// folds spans [from_span, ...) as far as the chain holds, starting from state `cur` with `packet_base` packets counted;
// span_base / span_attempt (nspans_total each): per span its first packet's stream-wide number and the attempt whose
// record was taken (0: the span was not taken); span_bridge_base: the number of the first packet of the bridge the merge
// walked in front of it (== span_base where there was none) — events with TS_EVENT_BRIDGE in `attempt` count from there
void ts_launch_merge(hipStream_t st, const ts_scan_params &p, uint32_t from_span, uint64_t packet_base, const ts_walk_state &cur,
                     uint32_t *g_count, unsigned long long *g_first, unsigned long long *g_last, unsigned long long *span_base,
                     unsigned long long *span_bridge_base, uint32_t *span_attempt, ts_merge_out *out, ts_span_out *span_out);
// the bridges in front of spans [from_span, nspans_total) — from_span's from the state `cur` the chain arrived with — every
// span a workgroup of its own: in front of ts_launch_merge
void ts_launch_bridges(hipStream_t st, const ts_scan_params &p, uint32_t from_span, const ts_walk_state &cur);
void ts_launch_reset(hipStream_t st, uint32_t *g_count, unsigned long long *g_first, unsigned long long *g_last,
                     unsigned int *event_count, uint32_t *span_attempt, uint32_t nspans);
void ts_launch_generate(hipStream_t st, void *out, uint64_t nunits, uint32_t unit, uint64_t seed, int hdmv);
void ts_launch_generate_damaged(hipStream_t st, void *out, uint64_t nbytes, uint64_t period, uint64_t seed);

#endif
