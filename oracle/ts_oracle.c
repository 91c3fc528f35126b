/*
 * ts_oracle.c — TEST INFRASTRUCTURE: CPU restatement of the packet scan of the reference's MPEG-2 transport stream
 * demultiplexer, drmpeg/dtv-utils xport.c, as far as it decides the per-PID report of
 *
 *     xport -ps <file> <program that is in no PAT> <v> <a>        (add -h for 192-byte HDMV packets)
 *
 * i.e. the lines `packets for pid %4d <0x%04x> = %d, first = %lld, last = %lld` (xport.c:245-250),
 * `Transport Sync Error, skipped %d bytes, at %lld` (xport.c:4325-4327 / 4363-4365) and
 * `Discontinuity!, pid = %d <0x%04x>, received = %2d, expected = %2d, at %lld` (xport.c:2876-2884).  Only tests/, smoke() and the
 * cpu_baseline leg of bench.py may use it, and only as the checker; nothing here is linked into the product.
 *
 * What is restated, byte for byte as the reference walks it:
 *   xport.c:241-244    while (!feof) fread 16384 -> demux_mpeg2_transport(length, buffer): the CHUNKING matters (below)
 *   xport.c:4317-4373  out of sync: a byte 0x47 starts a packet (187 more bytes), anything else is skipped and counted;
 *                      HDMV mode first swallows the 4 bytes of the tp_extra_header in front of every packet
 *   xport.c:2844-2906  the three header bytes: transport_error_indicator + 13-bit PID; packet_counter++ at the second
 *                      one; pid_counter / first / last only when the error indicator is clear (xport.c:2861-2867);
 *                      adaptation_field_control & 2 arms the adaptation-field state
 *   xport.c:2908-2984  adaptation field: one length byte, then that many bytes, each consumed singly; the packet ends
 *                      when its 188 bytes are used up even if the (malformed) field is not: the rest of the field is
 *                      then taken out of the NEXT packet's payload
 *   xport.c:2985-3112  PID 0 (PAT): consumed byte-wise / in bursts bounded by the chunk: plain accounting
 *   xport.c:3875-4295  PID 0x1ffb (ATSC PSIP base PID) is parsed whether or not -g was given, again byte-wise / in
 *                      bounded bursts: plain accounting.  (A complete Master Guide Table there would register
 *                      further PIDs for the same treatment; the scan's domain is streams without one — the
 *                      reference itself indexes unallocated tables for them.)
 *   xport.c:4296-4315  every other PID (with a program number no PAT announces, program_map_pid, video_pid, audio_pid
 *                      and pcr_pid keep 0xffff and never match a 13-bit PID): the rest of the packet is skipped in one
 *                      step — with `(length - i) >= xport_packet_length` where `>` was meant, so a packet that ends
 *                      exactly ONE byte past a 16384-byte read is declared finished one byte early and its last byte
 *                      goes to the sync search (skipped as a 1-byte sync error, or taken for a sync byte if it is 0x47)
 *
 *   xport.c:2872-2889  the continuity counter of header byte 3 against the PID's last one (0xff = none yet): a line when
 *                      it is not the successor, the packet carries a payload and the PID is not the null PID; the
 *                      counter is remembered for every payload-carrying packet of a PID other than 0
 *
 * Not restated (does not touch the lines above): PAT contents, PCR / rate output.
 */
#define _FILE_OFFSET_BITS 64
#include "ts_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void ts_oracle_init(ts_oracle_state *s, int hdmv)
{
    memset(s, 0, sizeof(*s));
    s->hdmv = hdmv != 0;
    s->tp_extra_header_parse = 4; /* xport.c:2662, 2726 */
    memset(s->continuity_counter, 0xff, sizeof(s->continuity_counter)); /* xport.c:2720-2721 */
}

/* demux_mpeg2_transport(length, buffer) for one fread chunk (xport.c:2729-4378) */
void ts_oracle_feed(ts_oracle_state *s, const unsigned char *buffer, unsigned int length)
{
    for (unsigned int i = 0; i < length; i++) {
        if (s->sync_state) {
            if (s->header_parse != 0) { /* xport.c:2845-2906 */
                --s->packet_length;
                --s->header_parse;
                switch (s->header_parse) {
                case 2:
                    s->tei = (buffer[i] >> 7) & 1;
                    s->pid = (unsigned)(buffer[i] & 0x1f) << 8;
                    break;
                case 1:
                    s->pid |= buffer[i];
                    s->result.packets++;
                    if (s->tei == 0) {
                        s->result.count[s->pid]++;
                        if (s->result.first[s->pid] == 0)
                            s->result.first[s->pid] = s->result.packets;
                        s->result.last[s->pid] = s->result.packets;
                    }
                    break;
                case 0: { /* xport.c:2872-2892 */
                    const unsigned temp = buffer[i];
                    const unsigned adaptation_field_control = (temp >> 4) & 0x3;
                    if (((s->continuity_counter[s->pid] + 1u) & 0xf) != (temp & 0xf)) {
                        if ((adaptation_field_control & 0x1) && s->pid != 0x1fff) {
                            if (s->continuity_counter[s->pid] != 0xff) {
                                if (s->result.ndiscontinuities < TS_ORACLE_MAX_SYNC_ERRORS) {
                                    ts_oracle_discontinuity *d = &s->result.discontinuities[s->result.ndiscontinuities];
                                    d->at_packet = s->result.packets;
                                    d->after_sync_errors = s->result.nsync_errors;
                                    d->pid = s->pid;
                                    d->received = temp & 0xf;
                                    d->expected = (s->continuity_counter[s->pid] + 1u) & 0xf;
                                    d->pad = 0;
                                }
                                s->result.ndiscontinuities++;
                            }
                        }
                    }
                    if ((adaptation_field_control & 0x1) && s->pid)
                        s->continuity_counter[s->pid] = (unsigned char)(temp & 0xf);
                    if ((adaptation_field_control & 0x2) == 0x2)
                        s->af_state = 1;
                    break;
                }
                }
            } else if (s->af_state) { /* xport.c:2908-2917 */
                --s->packet_length;
                s->af_parse = buffer[i];
                s->af_state = 0;
            } else if (s->af_parse != 0) { /* xport.c:2918-2984 */
                --s->packet_length;
                --s->af_parse;
                if (s->packet_length == 0)
                    s->sync_state = 0;
            } else if (s->pid == 0 || s->pid == 0x1ffb) { /* xport.c:2985-3112, 3875-4295: byte-wise accounting */
                --s->packet_length;
                if (s->packet_length == 0)
                    s->sync_state = 0;
            } else { /* xport.c:4296-4315 */
                --s->packet_length;
                if ((length - i) >= s->packet_length) { /* (sic) */
                    i = i + s->packet_length;
                    s->packet_length = 0;
                } else {
                    s->packet_length = s->packet_length - (length - i) + 1;
                    i = length;
                }
                if (s->packet_length == 0)
                    s->sync_state = 0;
            }
        } else { /* xport.c:4317-4373 */
            const unsigned char sync = buffer[i];
            if (s->hdmv && s->tp_extra_header_parse != 0) {
                --s->tp_extra_header_parse;
            } else if (sync == 0x47) {
                s->sync_state = 1;
                s->packet_length = 187;
                s->header_parse = 3;
                if (s->skipped_bytes != 0) {
                    if (s->result.nsync_errors < TS_ORACLE_MAX_SYNC_ERRORS) {
                        s->result.sync_errors[s->result.nsync_errors].skipped = s->skipped_bytes;
                        s->result.sync_errors[s->result.nsync_errors].at_packet = s->result.packets;
                    }
                    s->result.nsync_errors++;
                    s->skipped_bytes = 0;
                }
                if (s->hdmv)
                    s->tp_extra_header_parse = 4;
            } else {
                s->skipped_bytes++;
            }
        }
    }
}

void ts_oracle_scan_mem(const unsigned char *data, size_t n, int hdmv, ts_oracle_result *out)
{
    ts_oracle_state *s = (ts_oracle_state *)malloc(sizeof(*s));
    ts_oracle_init(s, hdmv);
    for (size_t off = 0; off < n; off += 16384) { /* xport.c:241-244 */
        const size_t len = n - off < 16384 ? n - off : 16384;
        ts_oracle_feed(s, data + off, (unsigned int)len);
    }
    *out = s->result;
    free(s);
}

int ts_oracle_scan_file(const char *path, int hdmv, ts_oracle_result *out)
{
    FILE *fp = fopen(path, "rb");
    if (!fp)
        return -1;
    static unsigned char buffer[16384];
    ts_oracle_state *s = (ts_oracle_state *)malloc(sizeof(*s));
    ts_oracle_init(s, hdmv);
    while (!feof(fp)) {
        const unsigned int length = (unsigned int)fread(buffer, 1, 16384, fp);
        ts_oracle_feed(s, buffer, length);
    }
    fclose(fp);
    *out = s->result;
    free(s);
    return 0;
}

/* the report lines this oracle is pinned on, in the reference's order and format */
void ts_oracle_print(const ts_oracle_result *r, FILE *fp)
{
    uint64_t d = 0;
    const uint64_t nd = r->ndiscontinuities < TS_ORACLE_MAX_SYNC_ERRORS ? r->ndiscontinuities : TS_ORACLE_MAX_SYNC_ERRORS;
    for (uint64_t k = 0;; k++) { /* the two kinds of line, in the order the reference prints them */
        for (; d < nd && r->discontinuities[d].after_sync_errors <= k; d++)
            fprintf(fp, "Discontinuity!, pid = %d <0x%04x>, received = %2d, expected = %2d, at %lld\n",
                    (int)r->discontinuities[d].pid, r->discontinuities[d].pid, (int)r->discontinuities[d].received,
                    (int)r->discontinuities[d].expected, (long long)r->discontinuities[d].at_packet);
        if (k >= r->nsync_errors || k >= TS_ORACLE_MAX_SYNC_ERRORS)
            break;
        fprintf(fp, "Transport Sync Error, skipped %d bytes, at %lld\n", (int)r->sync_errors[k].skipped,
                (long long)r->sync_errors[k].at_packet);
    }
    for (int i = 0; i < 0x2000; i++)
        if (r->count[i] != 0)
            fprintf(fp, "packets for pid %4d <0x%04x> = %d, first = %lld, last = %lld\n", i, i, (int)r->count[i],
                    (long long)r->first[i], (long long)r->last[i]);
}

#ifdef TS_ORACLE_MAIN
int main(int argc, char **argv)
{
    int hdmv = 0;
    const char *path = NULL;
    for (int a = 1; a < argc; a++) {
        if (!strcmp(argv[a], "-h"))
            hdmv = 1;
        else
            path = argv[a];
    }
    if (!path) {
        fprintf(stderr, "usage: ts_oracle [-h] <file.ts>\n");
        return 2;
    }
    ts_oracle_result *r = (ts_oracle_result *)malloc(sizeof(*r));
    if (ts_oracle_scan_file(path, hdmv, r) != 0) {
        fprintf(stderr, "Cannot open bitstream file <%s>\n", path);
        return 255;
    }
    ts_oracle_print(r, stdout);
    free(r);
    return 0;
}
#endif
