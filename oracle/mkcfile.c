/*
 * mkcfile — write a deterministic synthetic gr_complex .cfile.
 * TEST INFRASTRUCTURE (fixture generator); the sample definition lives in
 * include/papr_synth.h and is shared with the HIP generator kernel.
 *
 * usage: mkcfile <out> <nsamples> [--seed S] [--scale F] [--spike]
 *                [--set IDX I Q]... [--extra-floats K] [--extra-bytes B] [--envelope constant|bursty]
 *   --spike         the bench workload: two equal 30 dB spikes (papr_synth_spike_spec)
 *   --set           force sample IDX to (I,Q); accepts nan/inf; up to 8
 *   --extra-floats  append K more floats (continuing the I/Q stream) => odd tails
 *   --extra-bytes   append B (1..3) stray bytes 0xA5,0x5A,0xC3 => size % 4 != 0
 *   --part K M      write only samples [K, K+M) of the n-sample stream, at their place in <out> (the file is
 *                   created if need be and never truncated), so that several mkcfile processes can fill one
 *                   large file side by side; tails (--extra-*) belong to a run without --part
 */
#define _FILE_OFFSET_BITS 64
#include "papr_synth.h"

#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: mkcfile <out> <nsamples> [--seed S] [--scale F] [--spike] "
                        "[--set IDX I Q]... [--extra-floats K] [--extra-bytes B]\n");
        return 2;
    }
    const char *path = argv[1];
    uint64_t n = strtoull(argv[2], NULL, 0);
    papr_synth_spec sp;
    memset(&sp, 0, sizeof(sp));
    sp.seed = PAPR_SYNTH_DEFAULT_SEED;
    uint64_t extra_floats = 0, part_first = 0, part_count = 0;
    int extra_bytes = 0, spike = 0, part = 0;
    uint32_t envelope = 0;
    for (int a = 3; a < argc; a++) {
        if (!strcmp(argv[a], "--seed") && a + 1 < argc) {
            sp.seed = strtoull(argv[++a], NULL, 0);
        } else if (!strcmp(argv[a], "--scale") && a + 1 < argc) {
            sp.scale = strtof(argv[++a], NULL);
        } else if (!strcmp(argv[a], "--spike")) {
            spike = 1;
        } else if (!strcmp(argv[a], "--set") && a + 3 < argc) {
            if (PAPR_SYNTH_NOV(&sp) >= PAPR_SYNTH_MAX_OVERRIDES) {
                fprintf(stderr, "mkcfile: too many --set\n");
                return 2;
            }
            papr_synth_override *o = &sp.ov[sp.n_overrides++];
            o->index = strtoull(argv[++a], NULL, 0);
            o->i = strtof(argv[++a], NULL);
            o->q = strtof(argv[++a], NULL);
        } else if (!strcmp(argv[a], "--envelope") && a + 1 < argc) {
            ++a;
            envelope = !strcmp(argv[a], "constant") ? PAPR_SYNTH_ENV_CONSTANT : !strcmp(argv[a], "bursty") ? PAPR_SYNTH_ENV_BURSTY : 0u;
        } else if (!strcmp(argv[a], "--part") && a + 2 < argc) {
            part = 1;
            part_first = strtoull(argv[++a], NULL, 0);
            part_count = strtoull(argv[++a], NULL, 0);
        } else if (!strcmp(argv[a], "--extra-floats") && a + 1 < argc) {
            extra_floats = strtoull(argv[++a], NULL, 0);
        } else if (!strcmp(argv[a], "--extra-bytes") && a + 1 < argc) {
            extra_bytes = atoi(argv[++a]);
        } else {
            fprintf(stderr, "mkcfile: bad argument %s\n", argv[a]);
            return 2;
        }
    }
    if (spike) {
        uint64_t seed = sp.seed;
        float scale = sp.scale;
        papr_synth_spike_spec(&sp, seed, n);
        sp.scale = scale;
    }
    sp.n_overrides |= envelope << 8;
    if (part && (part_first > n || part_count > n - part_first || extra_floats || extra_bytes)) {
        fprintf(stderr, "mkcfile: --part outside the stream, or combined with a tail\n");
        return 2;
    }
    FILE *fp;
    if (part) { /* create without truncating: other writers may already have filled their parts */
        int fd = open(path, O_WRONLY | O_CREAT, 0644);
        fp = fd >= 0 ? fdopen(fd, "wb") : NULL;
    } else {
        fp = fopen(path, "wb");
    }
    if (!fp) {
        perror(path);
        return 1;
    }
    if (part && fseeko(fp, (off_t)(part_first * 8), SEEK_SET) != 0) {
        perror("fseeko");
        return 1;
    }
    enum { BLK = 8192 };
    static float buf[2 * BLK];
    uint64_t total_floats = part ? 2 * (part_first + part_count) : 2 * n + extra_floats, done = 2 * part_first;
    while (done < total_floats) {
        uint64_t want = total_floats - done;
        if (want > 2 * BLK)
            want = 2 * BLK;
        for (uint64_t k = 0; k < want; k += 2) {
            float i, q;
            papr_synth_sample(&sp, (done + k) / 2, &i, &q);
            buf[k] = i;
            if (k + 1 < want)
                buf[k + 1] = q;
        }
        if (fwrite(buf, sizeof(float), want, fp) != want) {
            perror("fwrite");
            return 1;
        }
        done += want;
    }
    static const unsigned char stray[3] = {0xA5, 0x5A, 0xC3};
    if (extra_bytes > 0 && extra_bytes < 4)
        fwrite(stray, 1, (size_t)extra_bytes, fp);
    fclose(fp);
    return 0;
}
