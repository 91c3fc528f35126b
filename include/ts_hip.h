/*
 * ts_hip.h — C ABI of the MI355X (gfx950) transport-stream packet scan in libpaprhip.so: sync lock, per-PID packet
 * count and first / last packet number, and the continuity-counter check of an MPEG-2 transport stream (188-byte packets,
 * or 192-byte HDMV packets).
 *
 * It replaces the scan inside drmpeg/dtv-utils xport.c — which, like papr.c, has no library surface: the scan is
 * inline in main() and demux_mpeg2_transport() — for the report `xport -p` ends with:
 *
 *   reference xport.c                                            this ABI
 *   -----------------------------------------------------------  ------------------------------------------
 *   :241-244   while (!feof) fread 16384 -> demux (the read loop)  ts_hip_load_file / _upload / _adopt / _generate
 *   :4317-4373 sync acquisition, `Transport Sync Error` events     ts_hip_scan -> ts_scan_result.sync_errors
 *   :2844-2867 header parse, packet_counter, pid_counter[pid]++,   ts_hip_scan -> .packets, .count, .first, .last
 *              pid_first_packet / pid_last_packet
 *   :2872-2889 header byte 3: the continuity counter against the   ts_hip_scan -> .discontinuities
 *              PID's last one, `Discontinuity!` lines
 *   :245-250   printf("packets for pid ...")                       stays in the caller (ts_format_report is the
 *                                                                  reference's format, for tests and tools)
 *
 * Scope: the lines above as `xport -ps[h] <file> <program no PAT announces> <v> <a>` prints them — no demultiplexing,
 * no PSI / PES parsing, no PCR output.  Within that scope the result is the reference's bit for
 * bit, including what its 16384-byte read loop does to a packet that ends exactly one byte past a read
 * (xport.c:4302 `>=`): positions are therefore FILE offsets, and a stream must be scanned from its first byte.
 * Streams that carry a complete ATSC Master Guide Table on PID 0x1ffb are outside the domain (the reference starts
 * parsing further PIDs then).
 *
 * How: packets at a fixed stride from a known sync position are independent, so the stream is cut into one byte range
 * per CU and every range is scanned in parallel: stretches of "regular" packets (sync byte in place, whole, legal
 * adaptation field, not on the read-boundary quirk) one lane per packet header into per-workgroup LDS tables of count /
 * first / last; everything else — sync loss, false sync bytes, malformed adaptation fields, the truncated tail, the
 * read-boundary quirk — by the closed-form packet walker, ON THE DEVICE (ts_walk_core.h: the same step this library
 * exports as ts_walk, GPU-free, for the tests).  A range cannot know where the chain of packets enters it, so it
 * speculates (eight sync bytes in a row at the stride); a merge kernel checks that every range started exactly where
 * the one in front of it ended and folds the tables with stream-wide packet numbers; a range that guessed wrong is
 * scanned once more from the true state.  Speculation decides how many launches are made, never a number in the
 * result; damage costs its own bytes, once.
 *
 * Conventions as in papr_hip.h: plain C types, 0 or a negative PAPR_E_* code, no CPU fallback.
 */
#ifndef TS_HIP_H
#define TS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TS_PIDS 0x2000
#define TS_MAX_SYNC_ERRORS 4096
#define TS_MAX_DISCONTINUITIES 4096
#define TS_READ_CHUNK 16384u /* xport.c:70 `static unsigned char buffer[16384]` */

typedef struct ts_sync_error {
    uint64_t skipped;   /* bytes passed over before the stream locked again (xport.c prints it with %d) */
    uint64_t at_packet; /* packet_counter at that moment */
} ts_sync_error;

/* a `Discontinuity!` line (xport.c:2876-2884): header byte 3 of a payload-carrying packet (adaptation_field_control & 1) of
 * a PID other than the null PID does not continue that PID's last counter.  The two kinds of line interleave in the
 * reference's output: after_sync_errors = `Transport Sync Error` lines printed before this one. */
typedef struct ts_discontinuity {
    uint64_t at_packet;         /* packet_counter: the packet's own number */
    uint64_t after_sync_errors;
    uint32_t pid;
    uint8_t received, expected, pad[2];
} ts_discontinuity;

typedef struct ts_scan_result {
    uint64_t packets;          /* packet_counter (xport.c:34) */
    uint32_t count[TS_PIDS];   /* pid_counter: unsigned int, wraps like the reference's */
    uint64_t first[TS_PIDS];   /* pid_first_packet: 1-based packet number, 0 = PID never seen */
    uint64_t last[TS_PIDS];    /* pid_last_packet */
    uint64_t nsync_errors;     /* `Transport Sync Error` lines the reference prints; the first TS_MAX_SYNC_ERRORS are kept */
    ts_sync_error sync_errors[TS_MAX_SYNC_ERRORS];
    uint64_t ndiscontinuities; /* `Discontinuity!` lines the reference prints; the first TS_MAX_DISCONTINUITIES are kept */
    ts_discontinuity discontinuities[TS_MAX_DISCONTINUITIES];
    uint8_t cc_state[TS_PIDS]; /* continuity_counter[] as it stands (xport.c:2659): 0 = no payload packet of the PID yet,
                                  else its last counter + 1 (ts_walk keeps its state here between calls) */
    /* how the scan went (not part of the reference's output) */
    uint64_t bytes;            /* stream length */
    uint64_t gpu_packets;      /* packets counted one lane per packet (the rest: the device-side walker) */
    uint32_t launches;         /* scan-kernel launches (1 + the ranges that had to be scanned again) */
    uint32_t walks;            /* times the device-side walker took over from the one-lane-per-packet blocks */
    double kernel_ms;          /* sum of the scan kernels' durations (HIP events) */
    double merge_ms;           /* sum of the merge kernels' durations */
} ts_scan_result;

/* ---- the host walker: the scan's exact state between packets, and one step of it (GPU-free) ---------------------- */
typedef struct ts_walk_state {
    uint64_t pos;           /* file offset of the next byte to look at; the stream is out of sync there */
    uint64_t skipped;       /* bytes skipped so far in the current sync search (skipped_bytes) */
    uint32_t stale_af;      /* adaptation-field bytes a malformed field still owes: taken from the next packet's payload */
    uint32_t extra_pending; /* HDMV: tp_extra_header bytes still to swallow before a sync byte is looked for */
    int hdmv;
} ts_walk_state;

void ts_walk_init(ts_walk_state *st, int hdmv);
/* 1 if a GPU launch may take over at st->pos: nothing pending from earlier packets (a "clean" position) */
int ts_walk_is_clean(const ts_walk_state *st);
/* Walk `data` = file bytes [base, base + n) (base + n = end of what is available; `eof` says whether that is the
 * end of the stream) from st->pos, packet by packet, adding to `res`, until st->pos is clean again AND at least
 * `min_packets` packets were taken (or the data runs out).  Stops early — returning 0 — when the next packet might
 * reach past the window and eof == 0: the caller then supplies a window further on.  Returns the packets taken. */
uint64_t ts_walk(ts_walk_state *st, const unsigned char *data, uint64_t base, uint64_t n, int eof, uint64_t min_packets,
                 ts_scan_result *res);
/* the report lines of the reference for a result (xport.c:245-250, :4326 / :4364 and :2876-2884: sync errors and
 * discontinuities in the order they were printed, then the PIDs); returns the bytes written (excluding the terminating
 * NUL), at most cap - 1 — or 0, with an empty string in buf, when the result does not hold all of its lines inline
 * (res->nsync_errors > TS_MAX_SYNC_ERRORS or res->ndiscontinuities > TS_MAX_DISCONTINUITIES: the reference prints every one
 * of them, so a report from the inline lists would not be its report): fetch the lists with ts_hip_get_sync_errors /
 * ts_hip_get_discontinuities and use ts_format_report_all */
size_t ts_format_report(const ts_scan_result *res, char *buf, size_t cap);
/* the same with complete lists */
size_t ts_format_report_all(const ts_scan_result *res, const ts_sync_error *errors, uint64_t nerrors,
                            const ts_discontinuity *discs, uint64_t ndiscs, char *buf, size_t cap);

/* ---- GPU scan -------------------------------------------------------------------------------------------------- */
typedef struct ts_hip_ctx ts_hip_ctx;

int ts_hip_open(ts_hip_ctx **ctx, int device);
void ts_hip_close(ts_hip_ctx *ctx);
const char *ts_hip_last_error(const ts_hip_ctx *ctx);
/* the stream: copied from host memory, read from a file (whole file), caller-owned device memory, or synthetic */
int ts_hip_upload(ts_hip_ctx *ctx, const void *bytes, uint64_t nbytes);
int ts_hip_load_file(ts_hip_ctx *ctx, const char *path);
int ts_hip_adopt(ts_hip_ctx *ctx, void *device_bytes, uint64_t nbytes);
/* fill the stream with `npackets` synthetic packets of include/ts_synth.h (hdmv != 0: 192-byte units) */
int ts_hip_generate(ts_hip_ctx *ctx, uint64_t seed, uint64_t npackets, int hdmv);
/* the same stream with one damaged spot every `period` packets (include/ts_synth.h: ts_synth_damaged_byte — inserted
 * and missing bytes, an overwritten sync byte; npackets a multiple of 4 * period): the scan's unfriendly case */
int ts_hip_generate_damaged(ts_hip_ctx *ctx, uint64_t seed, uint64_t npackets, uint64_t period);
int ts_hip_download(ts_hip_ctx *ctx, void *bytes, uint64_t first, uint64_t nbytes);
/* One complete scan of the resident stream (what `xport -p[h]` reports: xport.c:241-250, 2842-2889, 4317-4373).  The kernel has
 * two forms with the same result: full per-PID tables, one span of the stream per CU — the faster one on a stream that is in
 * order — and per-slot tables, two spans per CU, for damaged streams; a scan starts in the first and is done again in the
 * second when its spans meet damage more than once per 6144 packets (out->launches and out->kernel_ms count both attempts;
 * TS_SCAN_FORM=auto|full|slots chooses, a context reads it when it is opened). */
int ts_hip_scan(ts_hip_ctx *ctx, int hdmv, ts_scan_result *out);
/* sizeof(ts_scan_result) as the LIBRARY was built (it has grown between ABI versions: papr_hip.h, PAPR_HIP_ABI_VERSION): a
 * caller compares it with its own before it hands ts_hip_scan or ts_walk a buffer */
size_t ts_hip_result_size(void);
/* EVERY sync error of the last ts_hip_scan, in the order the reference prints them (ts_scan_result holds the first
 * TS_MAX_SYNC_ERRORS inline; a stream that locks one byte off a 4-byte grid yields one `skipped 1 bytes` line per 4096
 * packets, i.e. more than that from ~3 GB on): their number, and a copy of entries [first, first + n) */
uint64_t ts_hip_sync_error_count(const ts_hip_ctx *ctx);
int ts_hip_get_sync_errors(const ts_hip_ctx *ctx, uint64_t first, uint64_t n, ts_sync_error *out);
/* ... and every discontinuity */
uint64_t ts_hip_discontinuity_count(const ts_hip_ctx *ctx);
int ts_hip_get_discontinuities(const ts_hip_ctx *ctx, uint64_t first, uint64_t n, ts_discontinuity *out);
/* The host threads that lay out the report's lines of a damaged stream, checked on their own (no GPU): `rounds` rounds of
 * 1 ... 64 jobs through `threads` threads (the caller's included), each job must run exactly once in its own round.  Returns 0,
 * or the number of the first round in which one did not (tests/test_ts_scan.py). */
int ts_host_pool_selftest(int threads, int rounds);

#ifdef __cplusplus
}
#endif
#endif /* TS_HIP_H */
