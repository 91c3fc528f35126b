/*
 * papr_host.c — the GPU-free part of libpaprhip's C ABI: merging shard
 * statistics and the host scalar stage (mean, PAPR, level table).
 *
 * Plain C compiled with gcc -O2 -ffp-contract=off on purpose: these few
 * scalars are where libm (log10, pow) and double->float rounding decide the
 * printed digits, so they are evaluated with the same compiler family, libm
 * and rounding points as the reference program (papr.c:131-141, 164-173).
 */
#define _FILE_OFFSET_BITS 64
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include "papr_exact_format.h"
#include "papr_hip.h"
#include "papr_hip_measure.h"

int papr_hip_abi_version(void)
{
    return PAPR_HIP_ABI_VERSION;
}

void papr_stats_init(papr_stats *s)
{
    memset(s, 0, sizeof(*s));
    s->nan_first_idx = PAPR_NO_INDEX;
}

/* "more extreme value, else smaller index": what sequential strict compares
 * (papr.c:105-126) reduce to when the stream is cut into ordered shards */
static void merge_max(float *v, uint64_t *i, float nv, uint64_t ni)
{
    if (nv > *v || (nv == *v && ni < *i)) {
        *v = nv;
        *i = ni;
    }
}

static void merge_min(float *v, uint64_t *i, float nv, uint64_t ni)
{
    if (nv < *v || (nv == *v && ni < *i)) {
        *v = nv;
        *i = ni;
    }
}

void papr_stats_merge(papr_stats *acc, const papr_stats *next)
{
    /* papr.c:104 — once the running sum is NaN it stays that NaN */
    if (!(acc->flags & PAPR_FLAG_NAN)) {
        if (next->flags & PAPR_FLAG_NAN)
            acc->sum = next->sum;
        else
            acc->sum = acc->sum + next->sum;
    }
    if (next->nan_first_idx < acc->nan_first_idx) {
        acc->nan_first_idx = next->nan_first_idx;
        acc->nan_first_neg = next->nan_first_neg;
    }
    acc->n += next->n;
    acc->flags |= next->flags;
    merge_max(&acc->peak, &acc->peak_idx, next->peak, next->peak_idx);
    merge_max(&acc->re_pos, &acc->re_pos_idx, next->re_pos, next->re_pos_idx);
    merge_min(&acc->re_neg, &acc->re_neg_idx, next->re_neg, next->re_neg_idx);
    merge_max(&acc->im_pos, &acc->im_pos_idx, next->im_pos, next->im_pos_idx);
    merge_min(&acc->im_neg, &acc->im_neg_idx, next->im_neg, next->im_neg_idx);
}

/* float -> int as cvttss2si: NaN / out of range => INT_MIN (papr.c:136,138) */
static int to_int_x86(float x)
{
    if (!(x == x) || x >= 2147483648.0f || x < -2147483648.0f)
        return INT_MIN;
    return (int)x;
}

/* pow(10, x_j) of the level tables: the argument of level j depends on j and the mode alone (papr.c:138-141: (float)j / 10;
 * papr.c:168-173: the float that accumulated j times 0.1), so the libm call is made once per j and process — the SAME
 * call with the same argument the reference makes, its result kept; what changes from file to file is `* mean` and the
 * rounding to float.  (301 pow() calls were 10 us of every -g step, made three times.) */
#define PAPR_POW_CACHE 2048
static double pow_cache[2][PAPR_POW_CACHE];
static pthread_once_t pow_once = PTHREAD_ONCE_INIT;
static void pow_fill(void)
{
    float tenth_db = 0.0f;
    for (int j = 0; j < PAPR_POW_CACHE; j++) {
        pow_cache[0][j] = pow(10, (double)((float)j / 10));
        pow_cache[1][j] = pow(10, (double)(tenth_db / 10));
        tenth_db = (float)(tenth_db + 0.1);
    }
}

/* (internal: the runtime copies the tables to the device, where the kernels that speculate the level tables use the very
 * same values — their tables then ARE the host's, given the same sum) */
const double *papr_level_pow_table(int graph, int *count)
{
    pthread_once(&pow_once, pow_fill);
    if (count)
        *count = PAPR_POW_CACHE;
    return pow_cache[graph ? 1 : 0];
}

int papr_levels(const papr_stats *total, int graph, double *mean_out, float *papr_out, float *levels, int cap)
{
    const double mean = total->sum / (double)(long long)total->n;       /* papr.c:131 / 164 */
    const float papr = (float)(10 * log10((double)total->peak / mean)); /* papr.c:134 / 165 */
    if (mean_out)
        *mean_out = mean;
    if (papr_out)
        *papr_out = papr;
    const int top = graph ? to_int_x86(papr * 10) : to_int_x86(papr);   /* papr.c:166 / 136 */
    const int nl = top < 0 ? 0 : top + 1;
    if (!levels)
        return nl;
    pthread_once(&pow_once, pow_fill);
    if (graph) {
        float tenth_db = 0.0f;                                          /* papr.c:168-173 */
        for (int j = 0; j < nl && j < cap; j++) {
            levels[j] = (float)((j < PAPR_POW_CACHE ? pow_cache[1][j] : pow(10, (double)(tenth_db / 10))) * mean);
            tenth_db = (float)(tenth_db + 0.1);
        }
    } else {
        for (int j = 0; j < nl && j < cap; j++)                         /* papr.c:138-141 */
            levels[j] = (float)((j < PAPR_POW_CACHE ? pow_cache[0][j] : pow(10, (double)((float)j / 10))) * mean);
    }
    return nl;
}

/* one-sweep mode: the speculative table — papr_levels' thresholds for the estimated mean, carried on
 * to max_db above it (the true table stops at the true peak, which is not known yet) */
int papr_guess_levels(const papr_stats *est_total, int graph, double max_db, float *levels, int cap)
{
    if (!est_total || !levels || cap <= 0 || est_total->n == 0 || !(max_db >= 0))
        return 0;
    const double mean = est_total->sum / (double)(long long)est_total->n;
    if (!(mean > 0) || mean > 3e38)
        return 0;
    int nl = 0;
    pthread_once(&pow_once, pow_fill);
    if (graph) {
        float tenth_db = 0.0f;
        while (nl < cap && (double)tenth_db <= max_db) {
            levels[nl] = (float)((nl < PAPR_POW_CACHE ? pow_cache[1][nl] : pow(10, (double)(tenth_db / 10))) * mean);
            nl++;
            tenth_db = (float)(tenth_db + 0.1);
        }
    } else {
        while (nl < cap && (double)nl <= max_db) {
            levels[nl] = (float)((nl < PAPR_POW_CACHE ? pow_cache[0][nl] : pow(10, (double)((float)nl / 10))) * mean);
            nl++;
        }
    }
    return nl;
}

/* Band half-width for an estimate whose relative standard error is est_total->peak: a relative change d of a float
 * moves its bit pattern by d * 2^23 * m patterns (m = its mantissa, 1 <= m < 2), so a band of h patterns covers at
 * least h / 2^24; 4.5 standard errors, never below 2^10 (float rounding of the thresholds themselves). */
int papr_sweep_band_for(const papr_stats *est_total)
{
    if (!est_total || !(est_total->peak >= 0.0f) || !(est_total->peak < 1.0f))
        return 14;
    const double want = 4.5 * (double)est_total->peak * 16777216.0;
    int lg = 10;
    while (lg < 20 && (double)(1u << lg) < want)
        lg++;
    return lg;
}

/* ---- one-sweep mode: the host halves of the speculation (papr_sweep.hip) -------------------------- */

uint32_t papr_level_key(float t)
{
    uint32_t bits;
    if (t != t)
        return UINT32_MAX;
    if (t < 0.0f)
        return 0u;
    if (t == 0.0f)
        return 1u;
    memcpy(&bits, &t, 4);
    return bits >= 0x7F800000u ? UINT32_MAX : bits + 1u;
}

static int cmp_u32(const void *a, const void *b)
{
    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

int papr_sweep_bands(const float *guess_levels, int nlevels, int band_log2, uint32_t *keys, uint32_t *edges)
{
    if (!guess_levels || !keys || !edges || nlevels <= 0 || band_log2 < 1 || band_log2 > 24)
        return 0;
    const uint32_t half = 1u << band_log2;
    int m = 0;
    for (int j = 0; j < nlevels; j++) {
        const uint32_t key = papr_level_key(guess_levels[j]);
        if (key != UINT32_MAX)
            keys[m++] = key;
    }
    if (m == 0)
        return 0;
    qsort(keys, (size_t)m, sizeof(uint32_t), cmp_u32);
    int u = 1;
    for (int j = 1; j < m; j++)
        if (keys[j] != keys[u - 1])
            keys[u++] = keys[j];
    m = u;
    for (int j = 0; j < m; j++) {
        const uint32_t g = keys[j];
        /* normal floats only, and neighbouring bands must not touch */
        if (g < 0x00800000u + half || g >= 0x7F800000u - half || (j && g - half <= keys[j - 1] + half))
            return 0;
        edges[2 * j] = g - half;
        edges[2 * j + 1] = g + half;
    }
    return m;
}

int papr_sweep_resolve(const uint32_t *guess_keys, int nguess, int band_log2, const uint64_t *above_band,
                       const float *levels, int nlevels, const uint64_t *stash_above, uint64_t *counts_above)
{
    if (nlevels < 0 || nguess <= 0 || !guess_keys || !above_band || (nlevels && (!levels || !stash_above || !counts_above)))
        return 0;
    const uint32_t half = 1u << band_log2;
    /* pass 1: every true threshold must lie inside a band (matched by value: the true table may be longer or
     * shorter than the guess); levels nothing can exceed need no band */
    for (int pass = 0; pass < 2; pass++) {
        for (int l = 0; l < nlevels; l++) {
            const uint32_t t = papr_level_key(levels[l]);
            if (t == UINT32_MAX) {
                if (pass)
                    counts_above[l] = 0;
                continue;
            }
            int lo = 0, hi = nguess; /* first guess key >= t */
            while (lo < hi) {
                const int mid = (lo + hi) / 2;
                if (guess_keys[mid] < t)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            int band = -1;
            if (lo < nguess && guess_keys[lo] - t <= half)
                band = lo;
            else if (lo > 0 && t - guess_keys[lo - 1] <= half)
                band = lo - 1;
            if (band < 0)
                return 0;
            if (pass)
                counts_above[l] = above_band[band] + stash_above[l];
        }
    }
    return 1;
}

int papr_file_samples(const char *path, uint64_t *nsamples)
{
    struct stat sb;
    if (!path || !nsamples)
        return PAPR_E_ARG;
    if (stat(path, &sb) != 0 || !S_ISREG(sb.st_mode))
        return PAPR_E_IO;
    const uint64_t nfloats = (uint64_t)sb.st_size / 4;
    *nsamples = (nfloats + 1) / 2; /* an odd float count still yields a (phantom) sample, papr.c:102 */
    return PAPR_OK;
}

/* ---- exact sequential sum: the host end of papr_exact.hip -------------------
 * Replays papr.c:104 (`sum += value`, double, file order) from the per-shard sum
 * programs: whole groups / segments of additions that provably stay inside one
 * binade are applied as one exact increment chosen by the parity of the running
 * sum's last mantissa bit; everything else is added sample by sample, exactly as
 * the reference does it. */

static int binade_of(double s)
{
    uint64_t b;
    memcpy(&b, &s, 8);
    const int biased = (int)((b >> 52) & 0x7ff);
    return (biased == 0 || biased == 0x7ff) ? INT_MIN : biased - 1023;
}

/* S += D[parity of S's last mantissa bit]; both S and the result must lie in binade E */
static int apply_pair(double *S, int E, double D0, double D1)
{
    uint64_t b;
    if (binade_of(*S) != E)
        return -1;
    memcpy(&b, S, 8);
    *S += (b & 1u) ? D1 : D0;
    return binade_of(*S) == E ? 0 : -1;
}

static void add_raw(double *S, const float *iq, uint64_t nsamples)
{
    double acc = *S;
    for (uint64_t k = 0; k < nsamples; k++) {
        const float re = iq[2 * k], im = iq[2 * k + 1];
        const float re2 = re * re, im2 = im * im; /* separate roundings, no FMA (papr.c:103) */
        const float pw = re2 + im2;
        acc += pw;
    }
    *S = acc;
}

/* a tile the sum changes binade in: run by run — the run's pair where the sum is in the pair's binade before and after
 * (all terms are >= 0: every intermediate sum then is, too), sample by sample where it is not (the run the sum crosses
 * in, a run the device's approximate prefix placed in the wrong binade, a run with no pair at all): ~128 steps and one
 * or two 16-sample runs per tile instead of 2048 dependent additions. */
/* The program was written by the GPU: none of it is in this core's caches, and a raw record's run table (2.5 KB, 11 KB
 * from the next one) is too short a stream for the hardware prefetcher — walked cold it cost 40 dependent trips to
 * memory per tile, more than its arithmetic.  So the table of the NEXT raw record is requested, all lines at once, while
 * the chain is busy in front of it. */
static void prefetch_run_table(const papr_exact_raw_rec *r)
{
    const char *p = (const char *)r->run_E, *e = (const char *)(r->run_D + PAPR_XF_TILE_RUNS);
    for (; p < e; p += 64)
        __builtin_prefetch(p, 0, 3);
}

static void add_raw_powers(double *S, const float *pw, uint64_t nsamples)
{
    double acc = *S;
    for (uint64_t k = 0; k < nsamples; k++)
        acc += pw[k]; /* the device's fl(fl(I*I) + fl(Q*Q)), as papr.c:103 forms it */
    *S = acc;
}

static void add_raw_tile(double *S, const papr_exact_raw_rec *r)
{
    for (int k = 0; k < PAPR_XF_TILE_RUNS; k++) {
        const int32_t e = r->run_E[k];
        if (e == PAPR_XF_ZERO)
            continue; /* sixteen times + 0.0: nothing changes (the sum is never -0) */
        if (e != PAPR_XF_AMBIG && binade_of(*S) == e) {
            uint64_t b;
            memcpy(&b, S, 8);
            const double S2 = *S + r->run_D[k][b & 1u];
            if (binade_of(S2) == e) {
                *S = S2;
                continue;
            }
        }
        add_raw_powers(S, r->pw + PAPR_XF_RUN_SAMPLES * k, PAPR_XF_RUN_SAMPLES);
    }
}

int papr_exact_chain(const void *const *programs, const size_t *bytes, int nprograms, double *sum_out)
{
    if (!sum_out)
        return PAPR_E_ARG;
    double S = 0.0;
    const int rc = papr_exact_chain_continue(&S, programs, bytes, nprograms);
    if (rc == PAPR_OK)
        *sum_out = S;
    return rc;
}

/* the same replay from a running sum the caller carries: a stream reduced window by window (papr_hip_stream_stats) replays each
 * window's program as soon as it exists, from the accumulator the windows before it left */
int papr_exact_chain_continue(double *sum_inout, const void *const *programs, const size_t *bytes, int nprograms)
{
    double *sum_out = sum_inout;
    if (!programs || !bytes || !sum_inout || nprograms < 0 || !(*sum_inout >= 0.0))
        return PAPR_E_ARG;
    double S = *sum_inout;
    for (int k = 0; k < nprograms; k++) {
        const unsigned char *p = (const unsigned char *)programs[k], *end = p + bytes[k];
        papr_exact_header h;
        if (!p || bytes[k] < sizeof(h))
            return PAPR_E_ARG;
        memcpy(&h, p, sizeof(h));
        p += sizeof(h);
        if (h.magic != PAPR_EXACT_MAGIC || h.version != PAPR_EXACT_VERSION)
            return PAPR_E_ARG;
        /* sizes come from another process: every count is checked against what is left before it is used */
        uint64_t left = (uint64_t)(end - p);
        if (h.ngroups > left / sizeof(papr_exact_group_rec))
            return PAPR_E_ARG;
        left -= h.ngroups * sizeof(papr_exact_group_rec);
        if (h.nmixed > left / sizeof(papr_exact_mixed_rec))
            return PAPR_E_ARG;
        left -= (uint64_t)h.nmixed * sizeof(papr_exact_mixed_rec);
        if (h.nraw > left / sizeof(papr_exact_raw_rec))
            return PAPR_E_ARG;
        left -= (uint64_t)h.nraw * sizeof(papr_exact_raw_rec);
        if (h.tail_samples >= PAPR_XF_TILE_SAMPLES || (uint64_t)h.tail_samples * 8 > left)
            return PAPR_E_ARG;
        if (h.ngroups != (h.ntiles + PAPR_XF_GROUP_TILES - 1) / PAPR_XF_GROUP_TILES || h.reserved != 0)
            return PAPR_E_ARG;
        const papr_exact_group_rec *groups = (const papr_exact_group_rec *)p;
        const papr_exact_mixed_rec *mixed = (const papr_exact_mixed_rec *)(groups + h.ngroups);
        const papr_exact_raw_rec *raw = (const papr_exact_raw_rec *)(mixed + h.nmixed);
        const float *tail = (const float *)(raw + h.nraw);
        uint32_t mi = 0, ri = 0;
        if (h.nraw)
            prefetch_run_table(&raw[0]);
        for (uint64_t g = 0; g < h.ngroups; g++) {
            const papr_exact_group_rec *gr = &groups[g];
            if (gr->E == PAPR_XF_ZERO)
                continue;
            if (gr->E != PAPR_XF_AMBIG) {
                if (apply_pair(&S, gr->E, gr->D0, gr->D1))
                    return PAPR_E_INTERNAL;
                continue;
            }
            if (mi >= h.nmixed || mixed[mi].group != g)
                return PAPR_E_ARG;
            const papr_exact_mixed_rec *m = &mixed[mi++];
            for (uint64_t j = 0; j < PAPR_XF_GROUP_TILES; j++) {
                const uint64_t tile = g * PAPR_XF_GROUP_TILES + j;
                if (tile >= h.ntiles)
                    break;
                const int32_t e = m->tile_E[j];
                if (e == PAPR_XF_ZERO)
                    continue;
                if (e == PAPR_XF_AMBIG) {
                    if (ri >= h.nraw || raw[ri].tile != tile)
                        return PAPR_E_ARG;
                    if (ri + 1 < h.nraw)
                        prefetch_run_table(&raw[ri + 1]);
                    add_raw_tile(&S, &raw[ri++]);
                } else if (apply_pair(&S, e, m->seg_D[2 * j][0], m->seg_D[2 * j][1]) ||
                           apply_pair(&S, e, m->seg_D[2 * j + 1][0], m->seg_D[2 * j + 1][1])) {
                    return PAPR_E_INTERNAL;
                }
            }
        }
        if (mi != h.nmixed || ri != h.nraw)
            return PAPR_E_ARG;
        add_raw(&S, tail, h.tail_samples);
    }
    *sum_out = S;
    return PAPR_OK;
}
