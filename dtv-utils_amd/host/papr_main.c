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
 *   PAPR_XCH=rccl|auto|threads  how the shards' partial results meet: RCCL collectives on device buffers over xGMI, queued
 *                      on each GPU's stream (the default with more than one GPU when every shard has a GPU of its own;
 *                      =rccl also runs them for a single shard), or the plain in-process hub (=threads).  The communicators
 *                      come up in threads of their own BESIDE the ingest and are taken when the shards are loaded: waited
 *                      for up to PAPR_XCH_BIND_TIMEOUT_S (30) seconds, or — =auto — only if they are up by then; librccl
 *                      missing, a set-up that fails or is late: the hub, same stdout (and, when RCCL was asked for by name, one line on stderr)
 *   PAPR_STATS=1       one JSON line with sizes and timings on stderr
 *   PAPR_TEARDOWN=1    close the contexts and let the runtime's exit handlers run (default: _exit once the answer
 *                      is printed — the orderly way costs ~90 ms for a 10 GiB shard)
 *   PAPR_EXACT_SUM=0   skip the bit-exact emulation of the reference's sequential double
 *                      sum (papr.c:104) and print the mean from the parallel tree sum, which
 *                      differs from the reference's value by ~1e-13 relative (default: exact).
 *                      In that mode a file too large to stay in HBM is read ONCE instead of twice
 *                      (papr_hip_load_file_sweep: both passes ride along with the ingest);
 *                      PAPR_ONE_SWEEP=0 turns that off, =1 also uses it for shards that fit
 *                      (no gain there: pass 2 over a resident shard is 1.5 ms, the sample costs 40)
 * Structure: one thread per GPU shard opens its context, ingests its range of the file (pass 1 rides along
 * with the copy) and calls papr_hip_analyze — the sequence bench.py times — with the in-process transport of
 * papr_exchange between the threads; thread 0's result is printed.
 * There is no CPU fallback: without a usable GPU the program exits 254.
 */
#define _FILE_OFFSET_BITS 64
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "papr_hip.h"
#include "papr_exchange.h"
#include "papr_hip_measure.h" /* (PAPR_STATS=1: the ingest and sweep read-outs) */

#define MAX_GPUS 64
#define SHARD_ALIGN 8192ull /* samples: one reference fread chunk (papr.c:30), a multiple of the kernel tile */

typedef struct shard {
    papr_hip_ctx *ctx;
    papr_exchange *xch;
    int device, index, graph, exact, ingest_sweep;
    int stream_fd;        /* >= 0: the input cannot be positioned (a FIFO, a pipe): read once from this descriptor */
    const char *path;
    uint64_t first, count;
    int say_fallback;     /* PAPR_XCH=rccl|auto: a set-up that fails or is late is said on stderr; by default stderr stays the reference's */
    double bind_timeout_s, xch_setup_s, xch_waited_s; /* RCCL set-up beside the ingest: how long it may be waited for; what it took */
    uint64_t stream_windows; /* stream_fd >= 0: windows of HBM the stream crossed (papr_hip_stream_stats) */
    float *levels;        /* PAPR_HIP_MAX_LEVELS each */
    uint64_t *counts;
    papr_result res;
    papr_hip_sweep_info sweep;
    papr_hip_ingest_timing ingest;
    double t_open, t_loaded, t_done;
    int rc;
    char err[300];
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

static int shard_fail(shard *s, int rc, const char *what)
{
    s->rc = rc;
    snprintf(s->err, sizeof(s->err), "%s: %s (code %d)", what, s->ctx ? papr_hip_last_error(s->ctx) : papr_hip_last_error(NULL), rc);
    papr_exchange_abort(s->xch); /* the other shards' threads must not wait for this one */
    return rc;
}

/* everything one GPU does for its shard: context, ingest (papr.c:100-101), analysis (papr.c:102-153 / 164-185) */
static void *shard_thread(void *arg)
{
    shard *s = (shard *)arg;
    int rc = papr_hip_open(&s->ctx, s->device);
    if (rc != PAPR_OK) {
        shard_fail(s, rc, "cannot open the GPU");
        return NULL;
    }
    papr_hip_set_exact(s->ctx, s->exact);
    s->t_open = now_s();
    if (s->ingest_sweep == 1 && s->stream_fd < 0) {
        /* "when the file is streamed": does any shard exceed its GPU's HBM budget?  (decided together: every thread
         * must take the same path through the exchanges) */
        uint64_t nofit = papr_hip_shard_fits(s->ctx, s->count) == 0 ? 1u : 0u;
        rc = papr_exchange_counts(s->xch, &nofit, 1);
        if (rc != PAPR_OK) {
            shard_fail(s, rc, "exchange");
            return NULL;
        }
        s->ingest_sweep = nofit > 0 ? 2 : 0;
    }
    if (s->stream_fd >= 0) {
        /* an input that cannot be rewound, of any length (papr.c:100-129 reads it in 64 KiB of memory): pass 1 over windows
         * of HBM, the sequential sum carried exactly from window to window; the reference's pass 2 finds such a stream at its
         * end and counts nothing (papr.c:142-143 / 174-175), so nothing is kept */
        rc = papr_hip_stream_stats(s->ctx, s->stream_fd, &s->res.total, &s->res.exact_sum, &s->stream_windows);
        if (rc != PAPR_OK) {
            shard_fail(s, rc, "ingest");
            return NULL;
        }
        s->count = s->res.total.n;
        papr_hip_get_ingest_timing(s->ctx, &s->ingest);
        s->t_loaded = now_s();
        s->res.nlevels = papr_levels(&s->res.total, s->graph, &s->res.mean, &s->res.papr, s->levels, PAPR_HIP_MAX_LEVELS);
        if (s->res.nlevels > PAPR_HIP_MAX_LEVELS) {
            s->rc = PAPR_E_LIMIT;
            snprintf(s->err, sizeof(s->err), "analysis: %d levels exceed the table (code %d)", s->res.nlevels, PAPR_E_LIMIT);
            return NULL;
        }
        s->t_done = now_s();
        return NULL;
    } else if (s->ingest_sweep) {
        /* one read of the FILE for both passes: a 1-in-64 sample of every shard gives the mean to ~1e-4, the level
         * table it implies is widened into bands, and pass 2 rides along with pass 1 on the ingest */
        papr_stats est, est_total;
        double before = 0.0;
        rc = papr_hip_estimate_file(s->ctx, s->path, s->first, s->count, &est);
        if (rc != PAPR_OK) {
            shard_fail(s, rc, "estimate");
            return NULL;
        }
        rc = papr_exchange_stats(s->xch, &est, &est_total, &before, NULL);
        if (rc != PAPR_OK) {
            shard_fail(s, rc, "exchange");
            return NULL;
        }
        const int nguess = est_total.n ? papr_guess_levels(&est_total, s->graph, s->graph ? 48.0 : 60.0, s->levels, PAPR_HIP_MAX_LEVELS) : 0;
        papr_hip_set_band(s->ctx, papr_sweep_band_for(&est_total));
        if (s->exact) /* the running sum in front of this shard, about: the sweep speculates the sum's binades from it */
            (void)papr_hip_set_exact_hint(s->ctx, isfinite(before) && before >= 0.0 ? before : 0.0);
        rc = papr_hip_load_file_sweep(s->ctx, s->path, s->first, s->count, s->levels, nguess);
    } else {
        rc = papr_hip_load_file(s->ctx, s->path, s->first, s->count);
    }
    if (rc != PAPR_OK) {
        shard_fail(s, rc, "ingest");
        return NULL;
    }
    papr_hip_get_ingest_timing(s->ctx, &s->ingest);
    s->t_loaded = now_s();
    /* The RCCL communicators have been coming up in threads of their own since main() (ncclCommInitRank is seconds around a
     * step of milliseconds; every exchange up to here met at the in-process hub): take them now, all shards or none —
     * librccl missing, a failed or unfinished set-up is one line on stderr and the hub, never an error */
    rc = papr_exchange_adopt_rccl(s->xch, s->ctx, s->bind_timeout_s, &s->xch_setup_s, &s->xch_waited_s);
    if (rc != PAPR_OK) {
        s->rc = rc;
        snprintf(s->err, sizeof(s->err), "exchange: %s (code %d)", papr_exchange_last_error(s->xch), rc);
        papr_exchange_abort(s->xch);
        return NULL;
    }
    if (s->say_fallback && s->index == 0 && !papr_exchange_is_rccl(s->xch) && papr_exchange_last_error(s->xch)[0])
        fprintf(stderr, "papr: %s\n", papr_exchange_last_error(s->xch)); /* (only when PAPR_XCH asked for RCCL by name) */
    rc = papr_hip_analyze(s->ctx, s->xch, s->graph, 0, &s->res, s->levels, s->counts, PAPR_HIP_MAX_LEVELS);
    if (rc != PAPR_OK) {
        shard_fail(s, rc, "analysis");
        return NULL;
    }
    papr_hip_get_sweep_info(s->ctx, &s->sweep);
    {
        papr_hip_ingest_timing after;
        if (papr_hip_get_ingest_timing(s->ctx, &after) == PAPR_OK)
            s->ingest.file_passes = after.file_passes; /* (the analysis may have had to stream the file again) */
    }
    s->t_done = now_s();
    return NULL;
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
    /* The reference reads whatever fopen() gave it (papr.c:93, 100-101).  An input that cannot be positioned — a FIFO, a
     * pipe (`papr <(producer)`), a socket — is read ONCE, here through the descriptor that was just opened (a second open
     * of a FIFO would be a second reader); the reference's own second pass finds such a stream at its end (fseeko fails and
     * the EOF flag stays set: papr.c:142-143 / 174-175) and counts nothing, which is printed below as it prints it. */
    int stream_fd = -1;
    {
        struct stat sb;
        if (fstat(fileno(probe), &sb) == 0 && !S_ISREG(sb.st_mode) && lseek(fileno(probe), 0, SEEK_CUR) == (off_t)-1 && errno == ESPIPE)
            stream_fd = fileno(probe);
    }
    if (stream_fd < 0)
        fclose(probe);

    const double t0 = now_s();
    uint64_t nsamples = 0;
    if (stream_fd < 0 && papr_file_samples(path, &nsamples) != PAPR_OK) {
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
    if (stream_fd >= 0)
        ngpu = 1; /* (one reader, one shard: the length is known when the stream ends) */
    env = getenv("PAPR_OVERSUBSCRIBE");
    const int oversubscribe = env && atoi(env) > 0;
    if (ngpu > visible && !oversubscribe)
        ngpu = visible;
    if (ngpu > MAX_GPUS)
        ngpu = MAX_GPUS;
    env = getenv("PAPR_EXACT_SUM");
    const int exact = !(env && atoi(env) == 0 && env[0] != '\0');

    static shard sh[MAX_GPUS];
    memset(sh, 0, sizeof(sh));
    uint64_t per = (nsamples + (uint64_t)ngpu - 1) / (uint64_t)ngpu;
    per = (per + SHARD_ALIGN - 1) / SHARD_ALIGN * SHARD_ALIGN;
    int used = 0;
    for (int g = 0; g < ngpu; g++) {
        const uint64_t first = (uint64_t)g * per;
        if (g > 0 && first >= nsamples)
            break;
        sh[g].index = g;
        sh[g].device = g % visible;
        sh[g].path = path;
        sh[g].graph = graph;
        sh[g].exact = exact;
        sh[g].stream_fd = stream_fd;
        sh[g].first = first;
        sh[g].count = first + per > nsamples ? nsamples - first : per;
        used++;
    }
    ngpu = used;

    /* one-sweep ingest: by default for shards that will not stay in HBM, where it saves a whole second read of the
     * file — in exact-sum mode too: the few tiles the sum program needs again are read back from the file (the shard
     * threads settle that between them once their contexts know the budget) */
    env = getenv("PAPR_ONE_SWEEP");
    const int one_sweep = env && env[0] != '\0' ? (atoi(env) > 0 ? 2 : 0) : 1; /* 2 = always, 1 = when the file is streamed */
    papr_exchange *xs[MAX_GPUS];
    /* the shards' partial peak / mean / histogram meet over RCCL (collectives on device buffers, on each GPU's stream) when
     * there is more than one GPU — or when asked to; otherwise, or when librccl cannot be loaded, at the in-process hub */
    env = getenv("PAPR_XCH");
    const int rccl_forced = env && strcmp(env, "rccl") == 0;
    const int rccl_auto = env && strcmp(env, "auto") == 0; /* RCCL only if it is up when the shards are loaded: no wait at all */
    const int rccl_wanted = (rccl_forced || rccl_auto || (ngpu > 1 && !(env && strcmp(env, "threads") == 0))) && stream_fd < 0;
    int devices[MAX_GPUS];
    for (int g = 0; g < ngpu; g++)
        devices[g] = sh[g].device;
    /* (the set-up threads start here: librccl, the id and ncclCommInitRank run beside papr_hip_open and the ingest) */
    const int xrc = rccl_wanted ? papr_exchange_open_rccl_local_async(xs, ngpu, devices) : papr_exchange_open_local(xs, ngpu);
    if (xrc != PAPR_OK) {
        fprintf(stderr, "papr: out of memory\n");
        return 253;
    }
    env = getenv("PAPR_XCH_BIND_TIMEOUT_S");
    const double bind_timeout_s = rccl_auto ? 0.0 : (env && atof(env) > 0 ? atof(env) : 30.0);
    for (int g = 0; g < ngpu; g++) {
        sh[g].xch = xs[g];
        sh[g].bind_timeout_s = bind_timeout_s;
        sh[g].say_fallback = rccl_forced || rccl_auto;
        sh[g].ingest_sweep = one_sweep;
        sh[g].levels = (float *)malloc(PAPR_HIP_MAX_LEVELS * sizeof(float));
        sh[g].counts = (uint64_t *)calloc(PAPR_HIP_MAX_LEVELS, sizeof(uint64_t));
        if (!sh[g].levels || !sh[g].counts) {
            fprintf(stderr, "papr: out of memory\n");
            return 253;
        }
    }

    /* stdout carries the reference's text and nothing else: RCCL prints a version banner there when a communicator is
     * created, so while the shards work (nothing of ours is printed before they are done) fd 1 points at /dev/null */
    int saved_stdout = -1;
    if (rccl_wanted) {
        fflush(stdout);
        saved_stdout = dup(1);
        const int nul = open("/dev/null", O_WRONLY);
        if (saved_stdout >= 0 && nul >= 0)
            dup2(nul, 1);
        if (nul >= 0)
            close(nul);
    }

    /* ---- one thread per shard (a shard whose thread cannot be created runs inline, last) ---- */
    pthread_t th[MAX_GPUS];
    int started[MAX_GPUS];
    for (int g = 1; g < ngpu; g++)
        started[g] = pthread_create(&th[g], NULL, shard_thread, &sh[g]) == 0;
    int inline_late = 0;
    for (int g = 1; g < ngpu; g++)
        inline_late |= !started[g];
    if (inline_late) { /* cannot meet the others at the exchange from one thread: give up cleanly */
        for (int g = 0; g < ngpu; g++)
            papr_exchange_abort(xs[g]);
        for (int g = 1; g < ngpu; g++)
            if (started[g])
                pthread_join(th[g], NULL);
        fprintf(stderr, "papr: cannot start a thread per GPU shard\n");
        return 253;
    }
    shard_thread(&sh[0]);
    for (int g = 1; g < ngpu; g++)
        pthread_join(th[g], NULL);
    /* (fd 1 stays on /dev/null: a set-up thread that was not waited for may still be inside RCCL; the report goes out through
     * the descriptor that was saved) */
    FILE *out = stdout;
    if (saved_stdout >= 0) {
        fflush(stdout);
        out = fdopen(saved_stdout, "w");
        if (!out) {
            dup2(saved_stdout, 1);
            out = stdout;
        }
    }
    for (int g = 0; g < ngpu; g++)
        if (sh[g].rc != PAPR_OK && sh[g].rc != PAPR_E_STATE) { /* (E_STATE: cancelled because another shard failed) */
            fprintf(stderr, "papr: GPU %d: %s\n", sh[g].device, sh[g].err);
            return sh[g].ctx ? 253 : 254;
        }
    for (int g = 0; g < ngpu; g++)
        if (sh[g].rc != PAPR_OK) {
            fprintf(stderr, "papr: GPU %d: %s\n", sh[g].device, sh[g].err);
            return 253;
        }
    const double t2 = now_s();

    /* every thread holds the same result; print shard 0's */
    const papr_result *r = &sh[0].res;
    const papr_stats total = r->total;
    const double mean = r->mean;
    const float papr = r->papr;
    const int nlevels = r->nlevels;
    uint64_t *count = sh[0].counts;
    if (stream_fd >= 0) {
        nsamples = total.n;
        for (int j = 0; j < nlevels; j++) /* the reference's pass 2 over a stream it cannot rewind: at EOF at once */
            count[j] = 0;
    }
    if (exact && !r->exact_sum && isfinite(total.sum))
        /* stdout stays the reference's format; the mean (and, rarely, a threshold) now comes from the parallel tree
         * sum, which may differ from the reference's sequential sum in the last printed digit: say so */
        fprintf(stderr, "papr: warning: bit-exact sequential sum abandoned; using the parallel sum\n");

    /* ---- output, byte for byte the reference's (papr.c:132-135,154-161 / 186-190) ---- */
    const long long offset = (long long)total.n;
    if (!graph) {
        fprintf(out, "Peak magnitude = %f\n", sqrt(total.peak));
        fprintf(out, "average power = %lf, peak power = %f @ %lld\n\n", mean, total.peak, (long long)total.peak_idx * 8);
        fprintf(out, "Maximum PAPR = %f\n", papr);
        for (int j = 0; j < nlevels; j++)
            fprintf(out, "percentage above %d dB = %0.8f\n", j, ((float)(long long)count[j] / (float)offset) * 100.0);
        fprintf(out, "\n");
        fprintf(out, "peak real positive = %f, peak imaginary positive = %f\n", total.re_pos, total.im_pos);
        fprintf(out, "peak real negative = %f, peak imaginary negative = %f\n\n", total.re_neg, total.im_neg);
        fprintf(out, "peak real positive @ %lld, peak imaginary positive @ %lld\n", (long long)total.re_pos_idx * 8,
               ((long long)total.im_pos_idx * 8) + 1);
        fprintf(out, "peak real negative @ %lld, peak imaginary negative @ %lld\n", (long long)total.re_neg_idx * 8,
               ((long long)total.im_neg_idx * 8) + 1);
    } else {
        for (int j = 0; j < nlevels; j++)
            fprintf(out, "%0.8f\n", ((float)(long long)count[j] / (float)offset) * 100.0);
    }
    fflush(out);

    env = getenv("PAPR_STATS");
    if (env && atoi(env) > 0) {
        const double t3 = now_s();
        int swept = 0, resolved = 0;
        double t_open = t0, t_loaded = t0;
        for (int g = 0; g < ngpu; g++) {
            swept += sh[g].sweep.swept;
            resolved += sh[g].sweep.resolved;
            if (sh[g].t_open > t_open) t_open = sh[g].t_open;
            if (sh[g].t_loaded > t_loaded) t_loaded = sh[g].t_loaded;
        }
        double xch_setup_s = 0.0, xch_waited_s = 0.0; /* the slowest shard's RCCL set-up, the longest wait for one */
        for (int g = 0; g < ngpu; g++) {
            if (sh[g].xch_setup_s > xch_setup_s) xch_setup_s = sh[g].xch_setup_s;
            if (sh[g].xch_waited_s > xch_waited_s) xch_waited_s = sh[g].xch_waited_s;
        }
        const papr_hip_ingest_timing *it = &sh[0].ingest;
        fprintf(stderr,
                "{\"samples\": %llu, \"bytes\": %llu, \"gpus\": %d, \"exchange\": \"%s\", \"levels\": %d, \"open_s\": %.6f, "
                "\"shards_swept\": %d, \"shards_resolved_from_sweep\": %d, \"stream_windows\": %llu, \"exchange_setup_s\": %.6f, \"exchange_wait_s\": %.6f, "
                "\"ingest_pass1_s\": %.6f, \"exact_sum\": %d, \"exact_redo_tiles\": %u, \"analysis_s\": %.6f, \"total_s\": %.6f, "
                "\"msamples_per_s\": %.3f, \"ingest_GBps\": %.2f, \"gpu0_ingest\": {\"setup_s\": %.4f, \"read_s\": %.4f, "
                "\"buffer_wait_s\": %.4f, \"issue_s\": %.4f, \"drain_s\": %.4f, \"chunks\": %llu, \"reader_threads\": %d, "
                "\"resident\": %d, \"o_direct\": %d, \"numa_bound\": %d, \"io_uring\": %d, \"file_passes\": %d}}\n",
                (unsigned long long)nsamples, (unsigned long long)nsamples * 8, ngpu,
                papr_exchange_is_rccl(xs[0]) ? "rccl" : (ngpu > 1 ? "threads" : "none"), nlevels, t_open - t0, swept, resolved,
                (unsigned long long)sh[0].stream_windows, xch_setup_s, xch_waited_s,
                t_loaded - t_open, r->exact_sum, r->exact_redo_tiles, t2 - t_loaded, t3 - t0, (double)nsamples / (t3 - t0) / 1e6,
                (double)nsamples * 8 / (t_loaded - t_open) / 1e9, it->setup_s, it->read_s, it->buffer_wait_s, it->issue_s,
                it->drain_s, (unsigned long long)it->chunks, it->reader_threads, it->resident, it->o_direct, it->numa_bound, it->io_uring, it->file_passes);
    }

    /* The answer is out.  Taking the contexts down in order — hipFree of the shard, unpinning the staging ring, the
     * runtime's own exit handlers — costs ~90 ms for a 10 GiB shard, a fifth of the whole run, and buys nothing the kernel
     * does not do for a dead process anyway: leave at once.  PAPR_TEARDOWN=1 keeps the orderly way (profilers and leak
     * checkers want it). */
    env = getenv("PAPR_TEARDOWN");
    if (!(env && atoi(env) > 0)) {
        fflush(NULL);
        _exit(0);
    }
    for (int g = 0; g < ngpu; g++) {
        free(sh[g].counts);
        free(sh[g].levels);
        papr_exchange_close(xs[g]); /* (the exchange's communicator lives on the context's device: the exchange goes first) */
        papr_hip_close(sh[g].ctx);
    }
    return 0;
}
