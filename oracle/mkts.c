/*
 * mkts — write a deterministic synthetic transport stream (include/ts_synth.h).  TEST INFRASTRUCTURE.
 * usage: mkts <out> <npackets> [--seed S] [--hdmv] [--damage PERIOD]
 *   --damage PERIOD   one damaged spot every PERIOD packets (ts_synth_damaged_byte; npackets a multiple of 4 * PERIOD)
 */
#define _FILE_OFFSET_BITS 64
#include "ts_synth.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: mkts <out> <npackets> [--seed S] [--hdmv] [--damage PERIOD]\n");
        return 2;
    }
    uint64_t seed = TS_SYNTH_DEFAULT_SEED, n = strtoull(argv[2], NULL, 0), damage = 0;
    int hdmv = 0;
    for (int a = 3; a < argc; a++) {
        if (!strcmp(argv[a], "--seed") && a + 1 < argc)
            seed = strtoull(argv[++a], NULL, 0);
        else if (!strcmp(argv[a], "--hdmv"))
            hdmv = 1;
        else if (!strcmp(argv[a], "--damage") && a + 1 < argc)
            damage = strtoull(argv[++a], NULL, 0);
        else {
            fprintf(stderr, "mkts: bad argument %s\n", argv[a]);
            return 2;
        }
    }
    FILE *fp = fopen(argv[1], "wb");
    if (!fp) {
        perror(argv[1]);
        return 1;
    }
    const uint32_t unit = hdmv ? 192 : 188;
    static unsigned char buf[192 * 256];
    if (damage) {
        if (hdmv || n % (4 * damage)) {
            fprintf(stderr, "mkts: --damage wants 188-byte packets and npackets a multiple of 4 * PERIOD\n");
            return 2;
        }
        const uint64_t size = ts_synth_damaged_size(n, damage);
        for (uint64_t pos = 0; pos < size; pos += sizeof(buf)) {
            const uint64_t m = size - pos < sizeof(buf) ? size - pos : sizeof(buf);
            for (uint64_t j = 0; j < m; j++)
                buf[j] = ts_synth_damaged_byte(seed, damage, pos + j);
            if (fwrite(buf, 1, m, fp) != m) {
                perror("fwrite");
                return 1;
            }
        }
        fclose(fp);
        return 0;
    }
    for (uint64_t k = 0; k < n; k += 256) {
        const uint64_t m = n - k < 256 ? n - k : 256;
        for (uint64_t j = 0; j < m; j++)
            for (uint32_t i = 0; i < unit; i++)
                buf[j * unit + i] = ts_synth_byte(seed, k + j, i, hdmv);
        if (fwrite(buf, unit, m, fp) != m) {
            perror("fwrite");
            return 1;
        }
    }
    fclose(fp);
    return 0;
}
