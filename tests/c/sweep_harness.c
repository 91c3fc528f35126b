/* sweep_harness — drives the host half of the one-sweep mode (papr_guess_levels, papr_sweep_bands,
 * papr_sweep_resolve in dtv-utils_amd/csrc/papr_host.c) with seeded random and hostile inputs, every array in an
 * exact-size heap block, for the AddressSanitizer / UBSan run in tests/test_sanitizers.py.
 * Usage: sweep_harness <seed> <rounds>; prints the number of resolved / refused / band-less rounds. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "papr_hip.h"
#include "papr_hip_measure.h"

static uint64_t state;
static uint32_t rnd(void)
{
    state = state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(state >> 33);
}
static float rnd_float(void)
{
    static const float special[] = {0.0f, -0.0f, 1e-45f, 1.17549435e-38f, 3.4e38f, -1.0f, 1.0f, 2.0f};
    const uint32_t r = rnd();
    if (r % 11 == 0)
        return special[rnd() % 8];
    if (r % 13 == 0)
        return NAN;
    if (r % 17 == 0)
        return INFINITY;
    uint32_t bits = rnd() & 0x7FFFFFFFu; /* any non-negative pattern, NaNs included */
    float f;
    memcpy(&f, &bits, 4);
    return (r & 1) ? f : (float)(rnd() % 100000) / 777.0f;
}

int main(int argc, char **argv)
{
    state = argc > 1 ? strtoull(argv[1], NULL, 0) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 100;
    int resolved = 0, refused = 0, bandless = 0;
    for (int r = 0; r < rounds; r++) {
        const int ng = (int)(rnd() % 700), nl = (int)(rnd() % 400), band = (int)(rnd() % 26);
        float *guess = (float *)malloc((size_t)(ng ? ng : 1) * sizeof(float));
        float *levels = (float *)malloc((size_t)(nl ? nl : 1) * sizeof(float));
        if (r % 3 == 0) { /* the real thing: a guess table from an estimate */
            papr_stats est;
            papr_stats_init(&est);
            est.sum = (double)rnd_float() * 1000.0;
            est.n = rnd() % 5 ? 1000 : 0;
            const int got = papr_guess_levels(&est, (int)(rnd() & 1), (double)(rnd() % 130) - 5.0, guess, ng);
            for (int j = got; j < ng; j++)
                guess[j] = rnd_float();
        } else {
            for (int j = 0; j < ng; j++)
                guess[j] = rnd_float();
        }
        for (int j = 0; j < nl; j++)
            levels[j] = (r % 2 && ng) ? guess[rnd() % (uint32_t)ng] * (1.0f + (float)((int)(rnd() % 2001) - 1000) * 1e-6f)
                                      : rnd_float();
        uint32_t *keys = (uint32_t *)malloc((size_t)(ng ? ng : 1) * sizeof(uint32_t));
        uint32_t *edges = (uint32_t *)malloc((size_t)(ng ? 2 * ng : 1) * sizeof(uint32_t));
        const int m = papr_sweep_bands(guess, ng, band, keys, edges);
        if (m <= 0) {
            bandless++;
        } else {
            uint64_t *above = (uint64_t *)malloc((size_t)m * sizeof(uint64_t));
            uint64_t *stash = (uint64_t *)malloc((size_t)(nl ? nl : 1) * sizeof(uint64_t));
            uint64_t *counts = (uint64_t *)malloc((size_t)(nl ? nl : 1) * sizeof(uint64_t));
            for (int j = 0; j < m; j++)
                above[j] = rnd();
            for (int j = 0; j < nl; j++)
                stash[j] = rnd();
            if (papr_sweep_resolve(keys, m, band, above, levels, nl, stash, counts))
                resolved++;
            else
                refused++;
            free(above);
            free(stash);
            free(counts);
        }
        free(guess);
        free(levels);
        free(keys);
        free(edges);
    }
    printf("%d %d %d\n", resolved, refused, bandless);
    return 0;
}
