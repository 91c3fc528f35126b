/*
 * papr_oracle.c — CPU restatement of the reference `papr` (drmpeg/dtv-utils
 * papr.c).  TEST INFRASTRUCTURE ONLY — see papr_oracle.h for who may use it
 * and how its parity with the reference program is pinned.
 *
 * Written from the behavioural spec in SURVEY.md section 8 / appendix A; each
 * function cites the reference lines whose behaviour it restates.  It keeps
 * the reference's cost structure on purpose (two sequential passes, serial
 * double accumulator, O(N*L) threshold loop) so that it is an honest timing
 * proxy when used as bench.py's cpu_baseline "port".
 */
#define _FILE_OFFSET_BITS 64
#include "papr_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- chunk feeder -------------------------------------------------------
 * The reference reads `fread(buffer, 4, 16384, fp)` into one static buffer
 * (papr.c:35, 101, 144, 176) and never clears it, so an odd trailing float is
 * paired with whatever the slot after it held before (previous chunk, stray
 * tail bytes, or the initial zeros).  One persistent window reproduces that
 * for both passes. */
typedef struct feeder {
    FILE *fp;           /* file source, or NULL */
    const float *mem;   /* memory source */
    uint64_t mem_left;  /* floats left in the memory source */
    int exhausted;
    float window[PAPR_ORACLE_CHUNK_FLOATS];
} feeder;

static void feeder_rewind(feeder *f, const float *mem, uint64_t nfloats)
{
    if (f->fp) {
        fseeko(f->fp, 0, SEEK_SET); /* papr.c:142, 174 */
    } else {
        f->mem = mem;
        f->mem_left = nfloats;
    }
    f->exhausted = 0;
}

/* returns the number of whole floats now at the front of f->window, or -1
 * when the source had already signalled end-of-data (papr.c:100 `!feof`) */
static int feeder_next(feeder *f)
{
    if (f->fp) {
        if (feof(f->fp))
            return -1;
        return (int)fread(f->window, sizeof(float), PAPR_ORACLE_CHUNK_FLOATS, f->fp);
    }
    if (f->exhausted)
        return -1;
    uint64_t take = f->mem_left < PAPR_ORACLE_CHUNK_FLOATS ? f->mem_left : PAPR_ORACLE_CHUNK_FLOATS;
    if (take)
        memcpy(f->window, f->mem, take * sizeof(float));
    f->mem += take;
    f->mem_left -= take;
    if (take < PAPR_ORACLE_CHUNK_FLOATS)
        f->exhausted = 1; /* a short read is what sets EOF on a stream */
    return (int)take;
}

/* ---- pass 1: power, mean accumulator, peak and component extrema --------
 * papr.c:102-128.  Float products rounded separately, float add, then the
 * double accumulator in file order; strict compares against 0.0-initialised
 * trackers so the first occurrence wins and NaN never does. */
static void pass1(feeder *f, papr_oracle_result *r)
{
    double acc = 0.0;
    int64_t k = 0;
    float peak = 0.0f, rp = 0.0f, rn = 0.0f, ip = 0.0f, in = 0.0f;
    int64_t peak_k = 0, rp_k = 0, rn_k = 0, ip_k = 0, in_k = 0;
    int got;

    while ((got = feeder_next(f)) >= 0) {
        const float *w = f->window;
        for (int p = 0; p < got; p += 2, k++) {
            const float re = w[p], im = w[p + 1];
            const float re2 = re * re, im2 = im * im;
            const float pw = re2 + im2;
            acc += pw;
            if (pw > peak) { peak = pw; peak_k = k; }
            if (re > rp) { rp = re; rp_k = k; }
            if (re < rn) { rn = re; rn_k = k; }
            if (im > ip) { ip = im; ip_k = k; }
            if (im < in) { in = im; in_k = k; }
        }
    }
    r->sum = acc;
    r->n = k;
    r->peak = peak;       r->peak_idx = peak_k;
    r->re_pos = rp;       r->re_pos_idx = rp_k;
    r->re_neg = rn;       r->re_neg_idx = rn_k;
    r->im_pos = ip;       r->im_pos_idx = ip_k;
    r->im_neg = in;       r->im_neg_idx = in_k;
}

/* (int) of a float the way x86-64 cvttss2si does it for the values that can
 * reach here: NaN and out-of-range give INT_MIN (papr.c:136, 138, 166, 168) */
static int trunc_like_x86(float x)
{
    if (!(x == x) || x >= 2147483648.0f || x < -2147483648.0f)
        return INT_MIN;
    return (int)x;
}

/* ---- host scalars: mean, PAPR, level table -------------------------------
 * papr.c:131,134,136-141 (default) and 164-173 (graph). */
void papr_oracle_levels(papr_oracle_result *r, int graph)
{
    r->mean = r->sum / (double)r->n;                          /* 0/0 -> NaN */
    r->papr = (float)(10 * log10((double)r->peak / r->mean));
    int top = graph ? trunc_like_x86(r->papr * 10) : trunc_like_x86(r->papr);
    int nl = top < 0 ? 0 : top + 1;
    r->nlevels = nl;
    r->level = nl ? (float *)malloc((size_t)nl * sizeof(float)) : NULL;
    r->count = nl ? (int64_t *)calloc((size_t)nl, sizeof(int64_t)) : NULL;
    if (graph) {
        float step = 0.0f;                                    /* float accumulator */
        for (int j = 0; j < nl; j++) {
            r->level[j] = (float)(pow(10, (double)(step / 10)) * r->mean);
            step = (float)(step + 0.1);
        }
    } else {
        for (int j = 0; j < nl; j++)
            r->level[j] = (float)(pow(10, (double)((float)j / 10)) * r->mean);
    }
}

/* ---- pass 2: samples above each level ------------------------------------
 * papr.c:143-153 / 175-185: every sample against every level, strict `>`. */
static void pass2(feeder *f, papr_oracle_result *r)
{
    const int nl = r->nlevels;
    const float *lv = r->level;
    int64_t *cnt = r->count;
    int got;
    while ((got = feeder_next(f)) >= 0) {
        const float *w = f->window;
        for (int p = 0; p < got; p += 2) {
            const float re2 = w[p] * w[p], im2 = w[p + 1] * w[p + 1];
            const float pw = re2 + im2;
            for (int j = 0; j < nl; j++)
                if (pw > lv[j])
                    cnt[j]++;
        }
    }
}

static int run(feeder *f, const float *mem, uint64_t nfloats, int graph, papr_oracle_result *out)
{
    memset(out, 0, sizeof(*out));
    feeder_rewind(f, mem, nfloats);
    pass1(f, out);
    papr_oracle_levels(out, graph);
    feeder_rewind(f, mem, nfloats);
    pass2(f, out);
    return 0;
}

int papr_oracle_run_file(const char *path, int graph, papr_oracle_result *out)
{
    feeder *f = (feeder *)calloc(1, sizeof(feeder));
    f->fp = fopen(path, "r");
    if (!f->fp) {
        free(f);
        return -1;
    }
    int rc = run(f, NULL, 0, graph, out);
    fclose(f->fp);
    free(f);
    return rc;
}

int papr_oracle_run_mem(const float *data, uint64_t nfloats, int graph, papr_oracle_result *out)
{
    feeder *f = (feeder *)calloc(1, sizeof(feeder));
    int rc = run(f, data, nfloats, graph, out);
    free(f);
    return rc;
}

/* pass 2 alone against a caller-supplied level table (papr.c:145-152) */
int papr_oracle_count_mem(const float *data, uint64_t nfloats, const float *levels, int nlevels, int64_t *counts)
{
    feeder *f = (feeder *)calloc(1, sizeof(feeder));
    papr_oracle_result r;
    memset(&r, 0, sizeof(r));
    r.nlevels = nlevels;
    r.level = (float *)levels;
    r.count = counts;
    for (int j = 0; j < nlevels; j++)
        counts[j] = 0;
    feeder_rewind(f, data, nfloats);
    pass2(f, &r);
    free(f);
    return 0;
}

static double percent_above(int64_t count, int64_t n)
{
    return ((float)count / (float)n) * 100.0;                 /* papr.c:155, 188 */
}

void papr_oracle_print(const papr_oracle_result *r, int graph, FILE *fp)
{
    if (graph) {                                              /* papr.c:186-190 */
        for (int j = 0; j < r->nlevels; j++)
            fprintf(fp, "%0.8f\n", percent_above(r->count[j], r->n));
        return;
    }
    /* papr.c:132-135 */
    fprintf(fp, "Peak magnitude = %f\n", sqrt((double)r->peak));
    fprintf(fp, "average power = %lf, peak power = %f @ %lld\n\n", r->mean, r->peak,
            (long long)(r->peak_idx * 8));
    fprintf(fp, "Maximum PAPR = %f\n", r->papr);
    /* papr.c:154-161 */
    for (int j = 0; j < r->nlevels; j++)
        fprintf(fp, "percentage above %d dB = %0.8f\n", j, percent_above(r->count[j], r->n));
    fprintf(fp, "\n");
    fprintf(fp, "peak real positive = %f, peak imaginary positive = %f\n", r->re_pos, r->im_pos);
    fprintf(fp, "peak real negative = %f, peak imaginary negative = %f\n\n", r->re_neg, r->im_neg);
    fprintf(fp, "peak real positive @ %lld, peak imaginary positive @ %lld\n",
            (long long)(r->re_pos_idx * 8), (long long)(r->im_pos_idx * 8 + 1));
    fprintf(fp, "peak real negative @ %lld, peak imaginary negative @ %lld\n",
            (long long)(r->re_neg_idx * 8), (long long)(r->im_neg_idx * 8 + 1));
}

void papr_oracle_free(papr_oracle_result *r)
{
    free(r->level);
    free(r->count);
    r->level = NULL;
    r->count = NULL;
}

static void usage(void)
{
    fprintf(stderr, "usage: papr -g <infile>\n");
    fprintf(stderr, "Options:\n");
    fprintf(stderr, "\tg = graph suitable output\n");
}

/* argv grammar, messages and exit status of the reference (papr.c:53-98) */
int papr_oracle_main(int argc, char **argv)
{
    int graph = 0;
    const char *path;
    if (argc != 2 && argc != 3) {
        usage();
        return 255;
    }
    if (argc == 2) {
        path = argv[1];
    } else {
        if (argv[1][0] != '-') {
            usage();
            return 255;
        }
        for (const char *c = argv[1] + 1; *c; c++) {
            if (*c == 'g' || *c == 'G')
                graph = 1;
            else
                fprintf(stderr, "Unsupported Option: %c\n", *c);
        }
        path = argv[2];
    }
    papr_oracle_result r;
    if (papr_oracle_run_file(path, graph, &r) != 0) {
        fprintf(stderr, "Cannot open bitstream file <%s>\n", path);
        return 255;
    }
    papr_oracle_print(&r, graph, stdout);
    papr_oracle_free(&r);
    return 0;
}

#ifdef PAPR_ORACLE_MAIN
int main(int argc, char **argv)
{
    return papr_oracle_main(argc, argv);
}
#endif
