/* ts_oracle.h — TEST INFRASTRUCTURE (see ts_oracle.c): the packet scan of the reference's xport.c, restated. */
#ifndef TS_ORACLE_H
#define TS_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#define TS_ORACLE_MAX_SYNC_ERRORS (1u << 20) /* room for every line of any stream the tests build (the reference prints them all) */

typedef struct ts_oracle_sync_error {
    uint64_t skipped;   /* bytes passed over before the next sync byte (printed with %d) */
    uint64_t at_packet; /* packet_counter when the stream locked again */
} ts_oracle_sync_error;

typedef struct ts_oracle_discontinuity { /* xport.c:2876-2884: a `Discontinuity!` line */
    uint64_t at_packet;         /* packet_counter when it was printed */
    uint64_t after_sync_errors; /* `Transport Sync Error` lines printed before it: the two kinds of line interleave */
    uint32_t pid, received, expected, pad;
} ts_oracle_discontinuity;

typedef struct ts_oracle_result {
    uint64_t packets;        /* packet_counter (xport.c:34) */
    uint32_t count[0x2000];  /* pid_counter (unsigned int: wraps like the reference's) */
    uint64_t first[0x2000];  /* pid_first_packet, 1-based packet numbers; 0 = never seen */
    uint64_t last[0x2000];   /* pid_last_packet */
    uint64_t nsync_errors;   /* how many "Transport Sync Error" lines the reference prints */
    ts_oracle_sync_error sync_errors[TS_ORACLE_MAX_SYNC_ERRORS];
    uint64_t ndiscontinuities; /* how many `Discontinuity!` lines the reference prints */
    ts_oracle_discontinuity discontinuities[TS_ORACLE_MAX_SYNC_ERRORS];
} ts_oracle_result;

typedef struct ts_oracle_state {
    int hdmv;
    unsigned sync_state, packet_length, header_parse, af_state, af_parse, tei, pid, tp_extra_header_parse;
    uint64_t skipped_bytes;
    unsigned char continuity_counter[0x2000]; /* xport.c:2659; 0xff = PID not seen with a payload yet (xport.c:2721) */
    ts_oracle_result result;
} ts_oracle_state;

void ts_oracle_init(ts_oracle_state *s, int hdmv);
void ts_oracle_feed(ts_oracle_state *s, const unsigned char *buffer, unsigned int length);
void ts_oracle_scan_mem(const unsigned char *data, size_t n, int hdmv, ts_oracle_result *out);
int ts_oracle_scan_file(const char *path, int hdmv, ts_oracle_result *out);
void ts_oracle_print(const ts_oracle_result *r, FILE *fp);

#endif
