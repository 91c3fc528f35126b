/*
 * papr — PAPR + CCDF calculator for gr_complex .cfile IQ captures, MI355X build.
 *
 * Drop-in for the `papr` tool of drmpeg/dtv-utils: same command line
 * (`papr <infile>` / `papr -g <infile>`, reference papr.c:53-98), same stdout
 * (papr.c:132-135,154-161 / 186-190), same stderr messages and exit codes.
 * What differs is where the two passes over the samples run: the file is cut
 * into one contiguous shard per GPU, each shard is streamed into HBM and
 * reduced there by libpaprhip (include/papr_hip.h), and this program only folds
 * the per-GPU records, evaluates the libm scalars and prints.
 *
 * Environment (argv grammar is left untouched on purpose):
 *   PAPR_GPUS=N        use N GPUs (default: one per 2 GiB of input, at most all visible)
 *   PAPR_OVERSUBSCRIBE=1  let PAPR_GPUS exceed the visible GPUs (shard g runs on GPU g mod visible);
 *                      for exercising the multi-shard path on a small machine
 *   PAPR_STATS=1       one JSON line with sizes and timings on stderr
 *   PAPR_EXACT_SUM=0   skip the bit-exact emulation of the reference's sequential double
 *                      sum (papr.c:104) and print the mean from the parallel tree sum, which
 *                      differs from the reference's value by ~1e-13 relative (default: exact).
 *                      In that mode a file too large to stay in HBM is read ONCE instead of twice
 *                      (papr_hip_load_file_sweep: both passes ride along with the ingest);
 *                      PAPR_ONE_SWEEP=0 turns that off, =1 also uses it for shards that fit
 *                      (no gain there: pass 2 over a resident shard is 1.5 ms, the sample costs 40)
 * There is no CPU fallback: without a usable GPU the program exits 254.
 */
#define _FILE_OFFSET_BITS 64
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "papr_hip.h"

#define MAX_GPUS 64
#define SHARD_ALIGN 8192ull /* samples: one reference fread chunk (papr.c:30), a multiple of the kernel tile */

typedef struct shard {
    papr_hip_ctx *ctx;
    int device;
    double before;       /* accurate sum of the shards before this one */
    uint64_t n_total;
    const void *program; /* this shard's exact-sum program */
    size_t program_bytes;
    const char *path;
    uint64_t first, count;
    papr_stats stats;
    const float *levels;
    int nlevels;
    uint64_t *counts;
    papr_stats estimate;   /* one-sweep ingest: sampled mean of this shard's file range */
    const float *guess;    /* ... and the speculative level table (NULL: plain ingest) */
    int nguess;
    int rc;
} shard;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void usage(void)
{
    fprintf(stderr, "usage: papr -g <infile>\n");
    fprintf(stderr, "Options:\n");
    fprintf(stderr, "\tg = graph suitable output\n");
}

static void *estimate_thread(void *arg)
{
    shard *s = (shard *)arg;
    s->rc = papr_hip_estimate_file(s->ctx, s->path, s->first, s->count, &s->estimate);
    return NULL;
}

static void *pass1_thread(void *arg)
{
    shard *s = (shard *)arg;
    if (s->guess)
        s->rc = papr_hip_load_file_sweep(s->ctx, s->path, s->first, s->count, s->guess, s->nguess);
    else
        s->rc = papr_hip_load_file(s->ctx, s->path, s->first, s->count);
    if (s->rc == PAPR_OK)
        s->rc = papr_hip_stats(s->ctx, &s->stats);
    return NULL;
}

static void *exact_thread(void *arg)
{
    shard *s = (shard *)arg;
    s->rc = papr_hip_ccdf_exact(s->ctx, s->levels, s->nlevels, s->counts, s->before, s->n_total, &s->program,
                                &s->program_bytes);
    return NULL;
}

static void *pass2_thread(void *arg)
{
    shard *s = (shard *)arg;
    s->rc = papr_hip_ccdf(s->ctx, s->levels, s->nlevels, s->counts);
    return NULL;
}

/* run fn on every shard, one thread per GPU (a shard whose thread cannot be created runs inline) */
static void run_shards(shard *sh, int n, void *(*fn)(void *))
{
    pthread_t th[MAX_GPUS];
    int started[MAX_GPUS];
    for (int g = 1; g < n; g++)
        started[g] = pthread_create(&th[g], NULL, fn, &sh[g]) == 0;
    fn(&sh[0]);
    for (int g = 1; g < n; g++) {
        if (started[g])
            pthread_join(th[g], NULL);
        else
            fn(&sh[g]);
    }
}

/* like run_all below, but a failure is not fatal (the caller has another way): no message here */
static int run_all_quiet(shard *sh, int n, void *(*fn)(void *))
{
    run_shards(sh, n, fn);
    for (int g = 0; g < n; g++)
        if (sh[g].rc != PAPR_OK)
            return sh[g].rc;
    return PAPR_OK;
}

static int run_all(shard *sh, int n, void *(*fn)(void *))
{
    run_shards(sh, n, fn);
    for (int g = 0; g < n; g++)
        if (sh[g].rc != PAPR_OK) {
            fprintf(stderr, "papr: GPU %d: %s (code %d)\n", sh[g].device, papr_hip_last_error(sh[g].ctx), sh[g].rc);
            return sh[g].rc;
        }
    return PAPR_OK;
}

int main(int argc, char **argv)
{
    int graph = 0;
    const char *path;

    /* ---- command line, as the reference parses it (papr.c:53-98) ---- */
    if (argc != 2 && argc != 3) {
        usage();
        exit(-1);
    }
    if (argc == 2) {
        path = argv[1];
    } else {
        if (argv[1][0] != '-') {
            usage();
            exit(-1);
        }
        for (size_t i = 1; i < strlen(argv[1]); i++) {
            if (argv[1][i] == 'g' || argv[1][i] == 'G')
                graph = 1;
            else
                fprintf(stderr, "Unsupported Option: %c\n", argv[1][i]);
        }
        path = argv[2];
    }
    FILE *probe = fopen(path, "r");
    if (probe == NULL) {
        fprintf(stderr, "Cannot open bitstream file <%s>\n", path);
        exit(-1);
    }
    fclose(probe);

    const double t0 = now_s();
    double t_open = t0;
    uint64_t nsamples = 0;
    if (papr_file_samples(path, &nsamples) != PAPR_OK) {
        fprintf(stderr, "Cannot open bitstream file <%s>\n", path);
        exit(-1);
    }

    /* ---- one shard per GPU ---- */
    int visible = papr_hip_device_count();
    if (visible <= 0) {
        fprintf(stderr, "papr: no usable GPU (%s); this build has no CPU path\n", papr_hip_last_error(NULL));
        return 254;
    }
    if (visible > MAX_GPUS)
        visible = MAX_GPUS;
    int ngpu;
    const char *env = getenv("PAPR_GPUS");
    if (env && atoi(env) > 0) {
        ngpu = atoi(env);
    } else {
        const uint64_t per_gpu = (2ull << 30) / 8; /* samples in 2 GiB */
        ngpu = (int)((nsamples + per_gpu - 1) / per_gpu);
        if (ngpu < 1)
            ngpu = 1;
    }
    env = getenv("PAPR_OVERSUBSCRIBE");
    const int oversubscribe = env && atoi(env) > 0;
    if (ngpu > visible && !oversubscribe)
        ngpu = visible;
    if (ngpu > MAX_GPUS)
        ngpu = MAX_GPUS;

    env = getenv("PAPR_EXACT_SUM");
    int exact = !(env && atoi(env) == 0 && env[0] != '\0');
    shard sh[MAX_GPUS];
    memset(sh, 0, sizeof(sh));
    uint64_t per = (nsamples + (uint64_t)ngpu - 1) / (uint64_t)ngpu;
    per = (per + SHARD_ALIGN - 1) / SHARD_ALIGN * SHARD_ALIGN;
    int used = 0;
    for (int g = 0; g < ngpu; g++) {
        const uint64_t first = (uint64_t)g * per;
        if (g > 0 && first >= nsamples)
            break;
        sh[g].device = g % visible;
        sh[g].path = path;
        sh[g].first = first;
        sh[g].count = first + per > nsamples ? nsamples - first : per;
        int rc = papr_hip_open(&sh[g].ctx, sh[g].device);
        if (rc != PAPR_OK) {
            fprintf(stderr, "papr: cannot open GPU %d: %s\n", sh[g].device, papr_hip_last_error(NULL));
            return 254;
        }
        papr_hip_set_exact(sh[g].ctx, exact);
        used++;
    }
    ngpu = used;
    t_open = now_s();

    /* ---- one-sweep ingest (tree-sum mode only: the exact-sum sweep needs pass 1's per-tile sums first):
     * a 1-in-64 tile sample of every shard gives the mean to ~1e-4, the level table it implies is widened
     * into bands, and pass 2 then rides along with pass 1 on the one read of the file ---- */
    float *guess = NULL;
    env = getenv("PAPR_ONE_SWEEP");
    int one_sweep = env && env[0] != '\0' ? (atoi(env) > 0 ? 2 : 0) : 1; /* 2 = forced, 1 = when the file is streamed */
    if (one_sweep == 1) {
        one_sweep = 0;
        for (int g = 0; g < ngpu; g++)
            if (papr_hip_shard_fits(sh[g].ctx, sh[g].count) == 0)
                one_sweep = 1;
    }
    if (!exact && one_sweep && run_all_quiet(sh, ngpu, estimate_thread) == PAPR_OK) {
        papr_stats est_total;
        papr_stats_init(&est_total);
        for (int g = 0; g < ngpu; g++)
            papr_stats_merge(&est_total, &sh[g].estimate);
        guess = (float *)malloc(PAPR_HIP_MAX_LEVELS * sizeof(float));
        const int nguess = guess ? papr_guess_levels(&est_total, graph, graph ? 48.0 : 60.0, guess, PAPR_HIP_MAX_LEVELS) : 0;
        for (int g = 0; g < ngpu && nguess > 0; g++) {
            sh[g].guess = guess;
            sh[g].nguess = nguess;
        }
    }
    const double t_est = now_s();

    /* ---- pass 1 on every shard, then fold in file order (papr.c:100-129) ---- */
    if (run_all(sh, ngpu, pass1_thread) != PAPR_OK)
        return 253;
    const double t1 = now_s();
    papr_stats total;
    papr_stats_init(&total);
    for (int g = 0; g < ngpu; g++) {
        sh[g].before = total.sum; /* accurate sum of everything before shard g */
        sh[g].n_total = nsamples;
        papr_stats_merge(&total, &sh[g].stats);
    }
    /* ---- host scalars (papr.c:131-141 / 164-173), here from the tree sum ---- */
    double mean;
    float papr;
    int nlevels = papr_levels(&total, graph, &mean, &papr, NULL, 0);
    if (nlevels > PAPR_HIP_MAX_LEVELS) {
        fprintf(stderr, "papr: %d levels exceed the supported maximum of %d\n", nlevels, PAPR_HIP_MAX_LEVELS);
        return 253;
    }
    float *level = (float *)malloc((size_t)(nlevels + 1) * sizeof(float));
    uint64_t *count = (uint64_t *)calloc((size_t)(nlevels + 1), sizeof(uint64_t));
    papr_levels(&total, graph, NULL, NULL, level, nlevels);
    for (int g = 0; g < ngpu; g++) {
        sh[g].levels = level;
        sh[g].nlevels = nlevels;
        sh[g].counts = (uint64_t *)calloc((size_t)(nlevels + 1), sizeof(uint64_t));
    }

    /* ---- pass 2 on every shard (papr.c:142-153 / 174-185).  papr.c:104 adds in file order in double; by
     * default that rounding sequence is reproduced exactly from the same sweep (papr_hip_ccdf_exact); with
     * NaN/Inf present the merged record already carries the reference's value. ---- */
    int exact_done = 0, need_pass2 = nlevels > 0;
    double t1x = now_s();
    int exact_rc = PAPR_OK;
    if (exact && isfinite(total.sum) && (exact_rc = run_all_quiet(sh, ngpu, exact_thread)) == PAPR_OK) {
        const void *progs[MAX_GPUS];
        size_t sizes[MAX_GPUS];
        double seq;
        for (int g = 0; g < ngpu; g++) {
            progs[g] = sh[g].program;
            sizes[g] = sh[g].program_bytes;
        }
        exact_rc = papr_exact_chain(progs, sizes, ngpu, &seq);
        if (exact_rc == PAPR_OK) {
            total.sum = seq;
            exact_done = 1;
            need_pass2 = 0;
            /* the exact sum almost never moves a float threshold; when it does, pass 2 runs again */
            float *level2 = (float *)malloc((size_t)(nlevels + 1) * sizeof(float));
            const int nlevels2 = papr_levels(&total, graph, &mean, &papr, level2, nlevels);
            if (nlevels2 != nlevels || memcmp(level, level2, (size_t)nlevels * sizeof(float)) != 0) {
                free(level2);
                if (nlevels2 > PAPR_HIP_MAX_LEVELS)
                    return 253;
                nlevels = nlevels2;
                level = (float *)realloc(level, (size_t)(nlevels + 1) * sizeof(float));
                count = (uint64_t *)realloc(count, (size_t)(nlevels + 1) * sizeof(uint64_t));
                papr_levels(&total, graph, NULL, NULL, level, nlevels);
                for (int g = 0; g < ngpu; g++) {
                    sh[g].levels = level;
                    sh[g].nlevels = nlevels;
                    sh[g].counts = (uint64_t *)realloc(sh[g].counts, (size_t)(nlevels + 1) * sizeof(uint64_t));
                }
                need_pass2 = nlevels > 0;
            } else {
                free(level2);
            }
        }
        t1x = now_s();
    }
    if (exact_rc != PAPR_OK) {
        /* stdout stays the reference's format; the mean (and, rarely, a threshold) now comes from the parallel
         * tree sum, which may differ from the reference's sequential sum in the last printed digit: say so */
        const char *why = "";
        for (int g = 0; g < ngpu; g++)
            if (sh[g].rc != PAPR_OK)
                why = papr_hip_last_error(sh[g].ctx);
        fprintf(stderr, "papr: warning: bit-exact sequential sum abandoned (code %d%s%s); using the parallel sum\n",
                exact_rc, why[0] ? ": " : "", why);
    }
    if (need_pass2 && run_all(sh, ngpu, pass2_thread) != PAPR_OK)
        return 253;
    memset(count, 0, (size_t)(nlevels + 1) * sizeof(uint64_t));
    for (int g = 0; g < ngpu; g++)
        for (int j = 0; j < nlevels; j++)
            count[j] += sh[g].counts[j];
    const double t2 = now_s();

    /* ---- output, byte for byte the reference's (papr.c:132-135,154-161 / 186-190) ---- */
    const long long offset = (long long)total.n;
    if (!graph) {
        printf("Peak magnitude = %f\n", sqrt(total.peak));
        printf("average power = %lf, peak power = %f @ %lld\n\n", mean, total.peak, (long long)total.peak_idx * 8);
        printf("Maximum PAPR = %f\n", papr);
        for (int j = 0; j < nlevels; j++)
            printf("percentage above %d dB = %0.8f\n", j, ((float)(long long)count[j] / (float)offset) * 100.0);
        printf("\n");
        printf("peak real positive = %f, peak imaginary positive = %f\n", total.re_pos, total.im_pos);
        printf("peak real negative = %f, peak imaginary negative = %f\n\n", total.re_neg, total.im_neg);
        printf("peak real positive @ %lld, peak imaginary positive @ %lld\n", (long long)total.re_pos_idx * 8,
               ((long long)total.im_pos_idx * 8) + 1);
        printf("peak real negative @ %lld, peak imaginary negative @ %lld\n", (long long)total.re_neg_idx * 8,
               ((long long)total.im_neg_idx * 8) + 1);
    } else {
        for (int j = 0; j < nlevels; j++)
            printf("%0.8f\n", ((float)(long long)count[j] / (float)offset) * 100.0);
    }
    fflush(stdout);

    env = getenv("PAPR_STATS");
    if (env && atoi(env) > 0) {
        const double t3 = now_s();
        papr_hip_ingest_timing it;
        memset(&it, 0, sizeof(it));
        papr_hip_get_ingest_timing(sh[0].ctx, &it);
        int swept = 0, resolved = 0;
        for (int g = 0; g < ngpu; g++) {
            papr_hip_sweep_info si;
            memset(&si, 0, sizeof(si));
            papr_hip_get_sweep_info(sh[g].ctx, &si);
            swept += si.swept;
            resolved += si.resolved;
        }
        fprintf(stderr,
                "{\"samples\": %llu, \"bytes\": %llu, \"gpus\": %d, \"levels\": %d, \"open_s\": %.6f, "
                "\"estimate_s\": %.6f, \"shards_swept\": %d, \"shards_resolved_from_sweep\": %d, "
                "\"ingest_pass1_s\": %.6f, \"exact_sum\": %d, \"exact_sum_s\": %.6f, \"pass2_s\": %.6f, \"total_s\": %.6f, \"msamples_per_s\": %.3f, "
                "\"ingest_GBps\": %.2f, \"gpu0_ingest\": {\"setup_s\": %.4f, \"read_s\": %.4f, \"buffer_wait_s\": %.4f, "
                "\"issue_s\": %.4f, \"drain_s\": %.4f, \"chunks\": %llu, \"reader_threads\": %d, \"resident\": %d, \"o_direct\": %d, \"numa_bound\": %d}}\n",
                (unsigned long long)nsamples, (unsigned long long)nsamples * 8, ngpu, nlevels, t_open - t0,
                t_est - t_open, swept, resolved, t1 - t_est,
                exact_done, t1x - t1, t2 - t1x, t3 - t0, (double)nsamples / (t3 - t0) / 1e6, (double)nsamples * 8 / (t1 - t_est) / 1e9,
                it.setup_s, it.read_s, it.buffer_wait_s, it.issue_s, it.drain_s, (unsigned long long)it.chunks,
                it.reader_threads, it.resident, it.o_direct, it.numa_bound);
    }

    for (int g = 0; g < ngpu; g++) {
        free(sh[g].counts);
        papr_hip_close(sh[g].ctx);
    }
    free(count);
    free(level);
    free(guess);
    return 0;
}
