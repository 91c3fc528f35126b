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
#include <string.h>
#include <sys/stat.h>

#include "papr_hip.h"

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
    if (graph) {
        float tenth_db = 0.0f;                                          /* papr.c:168-173 */
        for (int j = 0; j < nl && j < cap; j++) {
            levels[j] = (float)(pow(10, (double)(tenth_db / 10)) * mean);
            tenth_db = (float)(tenth_db + 0.1);
        }
    } else {
        for (int j = 0; j < nl && j < cap; j++)                         /* papr.c:138-141 */
            levels[j] = (float)(pow(10, (double)((float)j / 10)) * mean);
    }
    return nl;
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
