/*
 * ts_synth.h — index-addressable synthetic MPEG-2 transport stream (bench / test workload of the TS packet scan).
 *
 * Shared verbatim by the host tool (oracle/mkts.c), the tests and the HIP generator kernel, so byte i of packet k
 * of stream `seed` is the same everywhere: 64-bit integer arithmetic only (continuity counters included).  A unit is 188 bytes (or 192 in HDMV
 * form: four bytes of tp_extra_header in front).  The mix resembles a broadcast multiplex: one video PID carrying
 * half of the packets, two audio PIDs, a data PID, ~17 % null packets, a PAT on PID 0 every 64th packet, a legal
 * adaptation field on one packet in eight (every legal length class), the transport_error_indicator on about one
 * packet in 1000.  Every stream is regular (in sync, whole packets); damage is the tests' business.
 */
#ifndef TS_SYNTH_H
#define TS_SYNTH_H

#include <stdint.h>

#ifdef __HIPCC__
#define TS_SYNTH_FN __host__ __device__ static inline
#else
#define TS_SYNTH_FN static inline
#endif

#define TS_SYNTH_DEFAULT_SEED 0x7500001ull

TS_SYNTH_FN uint64_t ts_synth_mix(uint64_t x)
{
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

/* byte `i` (0 .. unit-1) of unit `k` */
TS_SYNTH_FN unsigned char ts_synth_byte(uint64_t seed, uint64_t k, uint32_t i, int hdmv)
{
    const uint64_t h = ts_synth_mix(seed ^ (k * 0x9E3779B97F4A7C15ull));
    if (hdmv) {
        if (i < 4)
            return (unsigned char)(ts_synth_mix(h + 0x1234u) >> (8 * i));
        i -= 4;
    }
    /* PIDs rotate through a fixed 16-slot frame, so a packet's rank within its PID — its continuity counter — is
     * index-addressable too; slot 12 of every fourth frame carries the PAT instead of a null packet */
    const uint32_t slot = (uint32_t)(k & 15u);
    const uint32_t pids[16] = {0x100, 0x101, 0x100, 0x102, 0x100, 0x120, 0x100, 0x1fff, 0x100, 0x101, 0x100, 0x102, 0x1fff,
                               0x100, 0x1fff, 0x100};
    const uint32_t prefix[16] = {0, 0, 1, 0, 2, 0, 3, 0, 4, 1, 5, 1, 1, 6, 2, 7}; /* earlier slots with the same PID */
    const uint32_t per_frame[16] = {8, 2, 8, 2, 8, 1, 8, 3, 8, 2, 8, 2, 3, 8, 3, 8};
    const int is_pat = slot == 12 && ((k >> 4) & 3u) == 0;
    const uint32_t pid = is_pat ? 0u : pids[slot];
    const uint32_t cc = is_pat ? (uint32_t)((k >> 6) & 15u) : (uint32_t)(((k >> 4) * per_frame[slot] + prefix[slot]) & 15u);
    const uint32_t tei = !is_pat && ((h >> 16) % 1009u) == 0 ? 1u : 0u;
    const uint32_t has_af = !is_pat && ((h >> 32) & 7u) == 0 ? 1u : 0u;
    const uint32_t af_lens[8] = {0, 1, 7, 20, 100, 181, 182, 183};
    const uint32_t af_len = af_lens[(h >> 36) & 7u];
    switch (i) {
    case 0: return 0x47;
    case 1: return (unsigned char)((tei << 7) | ((is_pat ? 1u : 0u) << 6) | (pid >> 8));
    case 2: return (unsigned char)(pid & 0xffu);
    case 3: return (unsigned char)(((has_af ? 3u : 1u) << 4) | cc);
    default: break;
    }
    if (has_af && i == 4)
        return (unsigned char)af_len;
    if (has_af && i == 5 && af_len > 0)
        return (unsigned char)((ts_synth_mix(h + 5) & 0xefu) & (af_len < 7 ? 0xffu : 0xffu)); /* flags: no PCR */
    if (is_pat) {
        /* pointer_field 0, table_id 0, a short section announcing program 1 on PID 0x30; the rest stuffing */
        const unsigned char pat[17] = {0x00, 0x00, 0xB0, 0x0D, 0x00, 0x01, 0xC1, 0x00, 0x00, 0x00, 0x01, 0xE0, 0x30, 0x2A, 0xB1, 0x04, 0xB2};
        return i - 4 < 17 ? pat[i - 4] : (unsigned char)0xff;
    }
    if (pid == 0x1fffu)
        return 0xff; /* null packets carry stuffing */
    return (unsigned char)(ts_synth_mix(h + (i >> 3)) >> (8 * (i & 7u)));
}

/* ---- the same stream with damage at a fixed period (the scan's unfriendly bench case) --------------------------
 * Every `period` packets something is wrong with the stream, in a cycle of four kinds: three garbage bytes inserted in
 * front of a packet (the stream leaves its grid), a sync byte overwritten (0x46), the first five bytes of a packet
 * missing (off the grid again), one byte inserted.  Four periods are 4 * period * 188 - 1 bytes, so byte `pos` of the
 * damaged stream is index-addressable like the clean one.  npackets must be a multiple of 4 * period. */
TS_SYNTH_FN uint64_t ts_synth_damaged_size(uint64_t npackets, uint64_t period)
{
    return npackets * 188u - npackets / (4u * period);
}

TS_SYNTH_FN unsigned char ts_synth_damaged_byte(uint64_t seed, uint64_t period, uint64_t pos)
{
    const uint64_t plain = period * 188u, block = 4u * plain - 1u;
    const uint64_t b = pos / block;
    uint64_t off = pos % block;
    /* group lengths within a block: +3 (insert), +0 (sync overwritten), -5 (bytes missing), +1 (insert) */
    const uint64_t len[4] = {plain + 3u, plain, plain - 5u, plain + 1u};
    uint32_t g = 0;
    while (off >= len[g]) {
        off -= len[g];
        g++;
    }
    const uint64_t k0 = (4u * b + g) * period; /* the group's first packet */
    if (g == 0 || g == 3) {
        const uint64_t ins = g == 0 ? 3u : 1u;
        if (off < ins) /* garbage; never a sync byte */
            return (unsigned char)((ts_synth_mix(seed ^ (pos * 0xD1B54A32D192ED03ull)) & 0xffu) == 0x47u
                                       ? 0x48u
                                       : (ts_synth_mix(seed ^ (pos * 0xD1B54A32D192ED03ull)) & 0xffu));
        off -= ins;
    } else if (g == 2) {
        off += 5u; /* packet k0 starts at its sixth byte */
    }
    const uint64_t k = k0 + off / 188u;
    const uint32_t i = (uint32_t)(off % 188u);
    if (g == 1 && k == k0 && i == 0)
        return 0x46;
    return ts_synth_byte(seed, k, i, 0);
}

#endif
