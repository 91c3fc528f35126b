/*
 * papr_hip.h — C ABI of libpaprhip.so, the MI355X (gfx950) implementation of
 * the `papr` IQ power-statistics hot path.
 *
 * The reference (drmpeg/dtv-utils papr.c) has no library or FFI surface: the
 * whole path is inline in main().  Each entry point below therefore names the
 * block of reference lines it replaces; INTEGRATION.md shows the edit a
 * maintainer of papr.c would make to call them.
 *
 *   reference papr.c                          this ABI
 *   ----------------------------------------  ---------------------------------
 *   :100-101 fread loop (pass 1 ingest)       papr_hip_load_file / _upload / _adopt
 *   :102-128 power, sum, peak, extrema        papr_hip_stats  (+ papr_stats_merge
 *                                             across shards / GPUs)
 *   :131,134,136-141 / :164-173 mean, PAPR,   papr_levels      (host scalar, libm)
 *            level table
 *   :142-153 / :174-185 rewind + O(N*L)       papr_hip_ccdf
 *            threshold counting
 *   (the second read of the file itself)      papr_hip_estimate + papr_hip_stats_sweep:
 *                                             both passes in ONE read of the shard
 *   :132-135,154-161 / :186-190 printing      stays in the caller (host/papr_main.c)
 *
 * Conventions: plain C types only (no HIP/torch types); every int-returning
 * call returns PAPR_OK (0) or a negative PAPR_E_* code and never throws or
 * aborts; papr_hip_last_error() gives the detail text.  One context drives one
 * GPU and owns one "shard": a contiguous range of the file's sample axis,
 * resident in HBM (or re-streamed from the file when it exceeds the HBM
 * budget).  A context is not thread-safe; different contexts may be driven
 * from different threads or processes.  Sample indices are global 64-bit
 * sample numbers (byte offset / 8), exactly the `offset` of papr.c:37.
 *
 * There is no CPU fallback: without a usable GPU papr_hip_open fails.
 */
#ifndef PAPR_HIP_H
#define PAPR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 6: papr_hip_stream_stats, papr_exact_chain_continue, papr_exchange_open_rccl_local_async / _adopt_rccl were added (nothing changed shape).
 * 5: ts_scan_result grew (continuity-counter lines), ts_format_report_all took two more arguments, the sum programs are
 * version 3, papr_exchange_selftest / ts_hip_result_size were added: a caller built against version 4 must be rebuilt */
#define PAPR_HIP_ABI_VERSION 6

enum {
    PAPR_OK = 0,
    PAPR_E_NO_DEVICE = -1, /* no usable GPU / bad device ordinal */
    PAPR_E_HIP = -2,       /* a HIP runtime call failed (see last_error) */
    PAPR_E_ARG = -3,       /* bad argument */
    PAPR_E_NOMEM = -4,     /* host or device allocation failed */
    PAPR_E_IO = -5,        /* file could not be opened / read */
    PAPR_E_STATE = -6,     /* call made before a shard was loaded */
    PAPR_E_LIMIT = -7,     /* level table larger than PAPR_HIP_MAX_LEVELS */
    PAPR_E_INTERNAL = -8   /* an exact-sum invariant did not hold (never expected; results are not used) */
};

#define PAPR_HIP_MAX_LEVELS 16384
#define PAPR_NO_INDEX UINT64_MAX

/* Result of pass 1 over one shard, or the merge of several (papr.c:37-49).
 * Trackers follow the reference exactly: all start at 0.0f with strict
 * compares, so the first occurrence of an extreme wins, NaN never wins, and a
 * tracker that never fires reports value 0 at index 0. */
typedef struct papr_stats {
    double sum;            /* sum of float powers, accumulated in double */
    uint64_t n;            /* samples in the shard (incl. the odd-tail phantom sample) */
    uint64_t peak_idx;     /* global index of the first max-power sample */
    uint64_t re_pos_idx, re_neg_idx, im_pos_idx, im_neg_idx;
    uint64_t nan_first_idx;/* first sample whose power is NaN, or PAPR_NO_INDEX */
    float peak;            /* max power (I*I + Q*Q in float, products rounded separately) */
    float re_pos, re_neg, im_pos, im_neg;
    uint32_t nan_first_neg;/* sign bit of that first NaN power as x86 would produce it */
    uint32_t flags;        /* PAPR_FLAG_* */
    uint32_t reserved;
} papr_stats;

#define PAPR_FLAG_NAN 1u      /* some power value was NaN */
#define PAPR_FLAG_ODD_TAIL 2u /* the shard ends with the reference's phantom sample */


typedef struct papr_hip_ctx papr_hip_ctx;

/* ---- context ------------------------------------------------------------ */
int papr_hip_device_count(void);                       /* >= 0, or a PAPR_E_* code */
int papr_hip_open(papr_hip_ctx **ctx, int device);
void papr_hip_close(papr_hip_ctx *ctx);
const char *papr_hip_last_error(const papr_hip_ctx *ctx); /* ctx may be NULL: last open error */

/* ---- shard residency (replaces the fread ingest, papr.c:100-101,143-144) -- */

/* Number of samples the reference would count for this file (papr.c:127
 * `offset`), including the phantom sample of an odd float count. */
int papr_file_samples(const char *path, uint64_t *nsamples);

/* Load samples [first_sample, first_sample + nsamples) of the file into this
 * context's shard; nsamples = UINT64_MAX means "to the end".  The reference's
 * odd-float / stray-byte tail behaviour (papr.c:102-103 with the stale static
 * buffer) is reproduced when the range includes the file's last sample.  The
 * copy to HBM is double-buffered and overlapped with the pass-1 kernel. */
int papr_hip_load_file(papr_hip_ctx *ctx, const char *path, uint64_t first_sample, uint64_t nsamples);
/* The whole of a stream that cannot be positioned — a FIFO, a pipe, a socket: the reference fopen()s and reads those like
 * any file (papr.c:62, 93, 100-101) — read ONCE from the open descriptor `fd` to its end into this context's shard, which
 * grows in HBM as the bytes arrive (one reader, pinned staging, the copies overlapped with the reads); the sample count
 * and the odd-float / stray-byte tail follow from the length as for a file.  *nsamples receives the count (may be NULL).
 * A stream longer than the HBM budget is PAPR_E_NOMEM here (the shard has to stay resident for further passes; pass 1 alone
 * of a stream of any length: papr_hip_stream_stats).  What the reference makes of
 * such an input in pass 2 — fseeko fails, EOF stays set, every level counts zero (papr.c:142-143 / 174-175) — is the
 * caller's to reproduce: bin/papr prints zero counts. */
int papr_hip_load_stream(papr_hip_ctx *ctx, int fd, uint64_t *nsamples);
/* Pass 1 (papr.c:100-129) over such a stream of ANY length: the reference reads a FIFO in 64 KiB of memory and keeps a
 * handful of scalars, and pass 2 counts nothing for an input it cannot rewind (papr.c:142-143 / 174-175) — no sample is needed
 * twice.  The stream crosses ONE window of HBM (min(budget, PAPR_STREAM_WINDOW_MB = 256 MiB)); every full window is reduced as a
 * resident shard — papr_hip_stats and, in exact-sum mode (papr_hip_set_exact), the window's sum program built from the exact
 * accumulator in front of it and replayed at once — and overwritten by the bytes that follow.  *total = the whole stream's
 * pass-1 record (indices global; n includes the phantom sample), *exact_sum = 1 if total->sum is the reference's accumulator
 * bit for bit (0: the parallel sum — exact mode off, or the stream holds NaN / Inf, for which every order gives the same
 * non-finite sum), *windows = how many windows were reduced.  The context holds no shard afterwards. */
int papr_hip_stream_stats(papr_hip_ctx *ctx, int fd, papr_stats *total, int *exact_sum, uint64_t *windows);
/* 1 if a shard of nsamples would be kept resident in HBM by papr_hip_load_file (it fits the context's HBM
 * budget: 90 % of the free memory at open, or PAPR_HBM_BUDGET_MB), 0 if it would be re-streamed from the file
 * for every further pass. */
int papr_hip_shard_fits(const papr_hip_ctx *ctx, uint64_t nsamples);


/* ---- pass 1 (papr.c:102-128) -------------------------------------------- */
int papr_hip_stats(papr_hip_ctx *ctx, papr_stats *out);

/* Host-side, GPU-free helpers. `merge` folds the stats of the NEXT shard in
 * file order into `acc` (sum added in call order; extrema by value, then by
 * smaller index). */
void papr_stats_init(papr_stats *s);
void papr_stats_merge(papr_stats *acc, const papr_stats *next);

/* Mean, PAPR and the level table exactly as papr.c:131,134,136-141 (graph=0)
 * or :164-173 (graph!=0) compute them.  Returns the number of levels L (>= 0)
 * and writes min(L, cap) of them; a NaN PAPR gives 0 levels (papr.c:138 with
 * (int)NaN = INT_MIN on x86-64). */
int papr_levels(const papr_stats *total, int graph, double *mean, float *papr, float *levels, int cap);

/* ---- bit-exact mean (papr.c:104: `sum += value`, double, strictly in file order) --
 * papr_hip_stats sums in a parallel tree, accurate to ~1e-15 but not the
 * reference's rounding sequence (which itself drifts 1e-13 .. 1e-11 from the
 * true sum as the file grows to 1e9 samples).  Exact mode reproduces the
 * reference's value bit for bit (see papr_exact.hip):
 *
 *   papr_hip_set_exact(ctx, 1)               before load/upload/adopt/generate + stats
 *   papr_hip_stats(ctx, &st)                 as usual (also leaves per-tile sums on the GPU)
 *   papr_hip_exact_program(ctx, before, n_total, &prog, &bytes)
 *        `before` = accurate sum of all samples of EARLIER shards (0 for the first),
 *        `n_total` = samples in the whole file; prog points at context-owned memory
 *        (valid until the next call) holding this shard's serialisable "sum program"
 *   papr_exact_chain(programs, sizes, nshards, &sum)   host, shards in file order
 *
 * then use `sum` as papr_stats.sum of the merged record.  Needs a finite sum (with
 * NaN/Inf present papr_hip_stats is already exact); papr_hip_exact_program needs a
 * shard that is resident in HBM, papr_hip_ccdf_exact below also serves shards
 * that are re-streamed from their file. */
int papr_hip_set_exact(papr_hip_ctx *ctx, int enabled);
int papr_hip_exact_program(papr_hip_ctx *ctx, double before, uint64_t n_total, const void **program, size_t *bytes);
/* The same with pass 2 fused into the one sweep over the samples (no extra HBM pass): counts against
 * `levels` — normally the table derived from the tree sum — AND the sum program.  If the exact sum then
 * yields a different table (rare: the two sums differ by ~1e-11), call papr_hip_ccdf with the new one. */
int papr_hip_ccdf_exact(papr_hip_ctx *ctx, const float *levels, int nlevels, uint64_t *counts_above, double before,
                        uint64_t n_total, const void **program, size_t *bytes);
int papr_exact_chain(const void *const *programs, const size_t *bytes, int nprograms, double *sum_out);
/* The same replay continued from *sum_inout (the accumulator every earlier program left; 0.0 in front of the first): on
 * PAPR_OK *sum_inout is the accumulator behind the last program given, otherwise it is unchanged. */
int papr_exact_chain_continue(double *sum_inout, const void *const *programs, const size_t *bytes, int nprograms);

/* ---- one-sweep mode: pass 1 and pass 2 in ONE read of the shard ----------------------
 * papr.c reads the file twice because its thresholds are mean * 10^(dB/10) (papr.c:131-141)
 * and the mean needs a full pass.  Here both passes are HBM-bound, so the second read is half
 * of the job; speculation removes it without changing one count (papr_sweep.hip):
 *
 *   papr_hip_estimate(ctx, &est)      est.sum / est.n = mean power over a pseudo-random 1/64 sample of the
 *                                     shard (1/64 of a pass); est.n = samples in the shard, est.sum = the
 *                                     sampled sum scaled to them, so shards merge with the right weights
 *   (merge the shards' estimates with papr_stats_merge; guess table = papr_guess_levels(&est_total, ...))
 *   papr_hip_stats_sweep(ctx, guess, L, &st)
 *                                     st is what papr_hip_stats returns (same trackers; the double sum
 *                                     from another, equally accurate summation tree); in the same read
 *                                     every power is binned against BANDS of +-2^14 bit patterns
 *                                     (+-0.1..0.2 %) around the guessed thresholds, and the few
 *                                     per cent that fall inside a band are stashed in HBM
 *   (true table from the merged stats, as always)
 *   papr_hip_ccdf(ctx, levels, L, counts)
 *                                     if every true threshold lies inside one of the bands,
 *                                     only the stash is re-examined (exact counts, ~1-6 % of a
 *                                     pass); otherwise — bad guess, stash overflow, unusual table —
 *                                     the shard is read again as before.  Same counts either way.
 *
 * papr_hip_stats_sweep falls back to plain papr_hip_stats by itself (exact-sum mode, shards that are
 * not resident, guess tables without a band form); papr_hip_get_sweep_info (papr_hip_measure.h) tells what
 * happened. */
enum {
    PAPR_SWEEP_OK = 0,
    PAPR_SWEEP_NONE = 1,        /* no sweep was made for the current shard */
    PAPR_SWEEP_MODE = 2,        /* exact-sum mode or a shard that is not resident: plain pass 1 */
    PAPR_SWEEP_NO_BANDS = 3,    /* the guess table has no band form (NaN/zero/denormal/crowded thresholds) */
    PAPR_SWEEP_OUT_OF_BAND = 4, /* a true threshold fell outside every band of the guess */
    PAPR_SWEEP_STASH_FULL = 5   /* more in-band samples than the stash holds (1/8 of the shard) */
};
int papr_hip_estimate(papr_hip_ctx *ctx, papr_stats *est);
/* Exact-sum mode (papr_hip_set_exact(ctx, 1)) is served by the same single read: papr_hip_estimate then also keeps
 * one sampled sum per 1 MiB group on the device, papr_hip_stats_sweep speculates from them in which binade the
 * reference's running double sum (papr.c:104) is while it crosses each 2048-sample tile and builds the tiles'
 * rounding functions for THAT binade in the sweep itself, and papr_hip_ccdf_exact / papr_hip_exact_program — given
 * the accurate sum of the earlier shards as before — only re-examine the tiles whose speculated binade turns out
 * wrong (papr_hip_sweep_info.exact_redo_tiles: a fraction of a per cent) instead of reading the shard again.
 * A shard that is not the first of its file passes the ESTIMATED sum of everything before it (the sum of the
 * earlier shards' estimate records) before the sweep; a poor hint costs re-examined tiles, never a result. */
int papr_hip_set_exact_hint(papr_hip_ctx *ctx, double estimated_sum_before_shard);
/* The same estimate for a range of a FILE that is not loaded yet (the 1-in-64 tiles are read by the
 * ingest's reader threads and summed on the GPU): what a one-sweep ingest needs before it starts.  In exact-sum
 * mode the sampled tiles' sums stay on the device for papr_hip_load_file_sweep of the same range. */
int papr_hip_estimate_file(papr_hip_ctx *ctx, const char *path, uint64_t first_sample, uint64_t nsamples, papr_stats *est);
/* Host helper: the table papr_levels would build for the (merged) estimate's mean, carried on to max_db
 * dB above the mean whatever the peak turns out to be (the real table's length depends on the true
 * peak; bands above it cost nothing).  Returns the number of levels written (<= cap). */
int papr_guess_levels(const papr_stats *est_total, int graph, double max_db, float *levels, int cap);
/* How wide the bands must be for THIS estimate: papr_hip_estimate also measures the scatter of its sample pieces and
 * leaves the estimate's relative standard error in est->peak (papr_stats_merge keeps the largest of the shards').
 * papr_sweep_band_for turns it into a band half-width (log2 of float bit patterns, 10..20) that the true
 * thresholds miss with a probability of a few in a million: 2^14 for a stationary capture sampled 1 in 64, wider
 * for a bursty one — which then stashes more samples but is still answered from one read — narrower when the
 * sample was (nearly) everything.  papr_hip_set_band hands it to the next papr_hip_stats_sweep /
 * papr_hip_load_file_sweep of the context (0 = the built-in default; PAPR_HIP_TUNE band= overrides both). */
int papr_sweep_band_for(const papr_stats *est_total);
int papr_hip_set_band(papr_hip_ctx *ctx, int band_log2);
int papr_hip_stats_sweep(papr_hip_ctx *ctx, const float *guess_levels, int nlevels, papr_stats *out);
/* papr_hip_load_file as a one-sweep ingest: the kernel that runs on every chunk as it lands also bins against the
 * guessed bands and stashes, so papr_hip_stats returns the file's pass-1 record as usual and papr_hip_ccdf needs
 * no second pass — neither over a resident shard nor, for a shard larger than the HBM budget, over the FILE
 * (which papr.c:142-144 and the plain path read twice).  Same fall-backs as papr_hip_stats_sweep.
 * Exact-sum mode rides along when papr_hip_estimate_file (in that mode) was called for the SAME range before: it
 * keeps the sampled tiles' sums, the ingest speculates the running sum's binades from them (papr_hip_set_exact_hint
 * for shards that are not the first) and builds the rounding functions in the same pass; papr_hip_ccdf_exact then
 * reads back from the file just the tiles it needs again (a few hundred of a 10 GiB shard's 655 360) — or, if the
 * speculation missed on more than 32 768 of them, streams the file once more.  papr_hip_ingest_timing.file_passes
 * tells which.  Without a matching estimate the ingest in exact-sum mode is the plain one (pass 1 only). */
int papr_hip_load_file_sweep(papr_hip_ctx *ctx, const char *path, uint64_t first_sample, uint64_t nsamples,
                             const float *guess_levels, int nlevels);


typedef struct papr_exchange papr_exchange; /* include/papr_exchange.h: the exchange between the shards of one file */

/* ---- the whole result in one call --------------------------------------------------------------------
 * papr_hip_analyze runs, for the shard loaded in `ctx` and — through `x` — together with the other shards'
 * processes / threads, everything papr.c:100-153 / 164-185 computes: the mean estimate and its exchange, the
 * guessed bands, ONE sweep over the samples (pass 1 + banded pass 2, in exact-sum mode also the rounding functions of
 * the sequential sum), the exchange of the pass-1 records and their ordered merge, mean / PAPR / level table exactly
 * as the reference derives them, the stash recount (or, speculation missed, pass 2), in exact-sum mode the chained
 * sequential sum (papr_hip_set_exact), and the exchange of the counters.  It is the sequence bench.py times and
 * bin/papr prints from; every step is one of the calls above and can be made separately.  `x` may be NULL for a
 * single shard.  levels / counts_above: caller's arrays of `cap` entries (PAPR_HIP_MAX_LEVELS always suffices);
 * all ranks receive the same result.
 * With the RCCL transports (papr_exchange_open_rccl, papr_exchange_open_rccl_local + papr_exchange_bind) the step's exchanges
 * are collectives queued on the context's stream between the kernels that produce and consume them — the estimate records,
 * the pass-1 records, in exact-sum mode every rank's sum program (a slot per rank, enlarged when a program outgrows it), the
 * counters — with no host staging: the whole step is one wait.  With the callback and in-process transports they are host
 * calls between three waits.
 * FAILURE WITH PEERS: the ranks first AGREE (a host all-reduce, repeated whenever a shard, a mode or the exchange changed)
 * that every one of them can take the single-wait step — buffers are allocated in front of that agreement and a rank that
 * cannot says so there, and all of them take the host path.  A rank whose call fails locally BEHIND the agreement, or
 * anywhere on the host path (a HIP error, out of memory, a shard that cannot be read), returns its error at once and does
 * NOT enter the collectives that were still to come; so that the other ranks do not wait in them for ever, the library
 * cancels the exchange on its way out of the single-wait step (papr_exchange_abort: the in-process hub's waiters return
 * PAPR_E_STATE, RCCL communicators are aborted with ncclCommAbort), and on the host path whoever gets a non-zero return
 * must do the same: papr_exchange_abort(x) (bin/papr does), or — with caller-supplied collectives — leave the process
 * group, as with any collective program. */
#define PAPR_ANALYZE_TWO_PASS 1u /* no speculation: pass 1, then pass 2 (two reads of the shard) */
#define PAPR_ANALYZE_SPOIL_GUESS 2u /* diagnostics: feed the sweep a guess that is 3 % off (what a missed speculation costs) */
typedef struct papr_result {
    papr_stats total;       /* whole file; .sum is the reference's sequential sum when exact_sum != 0 */
    double mean;            /* papr.c:131 / 164 */
    float papr;             /* papr.c:134 / 165 */
    int nlevels;            /* papr.c:136 / 166 */
    int exact_sum;          /* 1 = total.sum reproduces papr.c:104 bit for bit; 0 = parallel tree sum */
    int swept;              /* 1 = this shard went through the one-sweep kernel */
    int resolved;           /* 1 = this shard's counts came from the sweep (no second read) */
    int reason;             /* PAPR_SWEEP_* when not */
    int pass2_reruns;       /* exact mode: 1 if the exact sum moved a float threshold and the counts were redone */
    uint32_t exact_redo_tiles;
    int band_log2;
    int reserved;
} papr_result;
int papr_hip_analyze(papr_hip_ctx *ctx, papr_exchange *x, int graph, unsigned flags, papr_result *res, float *levels,
                     uint64_t *counts_above, int cap);

/* ---- pass 2 (papr.c:143-153 / 175-185) ---------------------------------- */
/* counts_above[j] = number of shard samples whose power is > levels[j]
 * (float compare, strict, NaN never counts).  Levels may be any floats. */
int papr_hip_ccdf(papr_hip_ctx *ctx, const float *levels, int nlevels, uint64_t *counts_above);

#ifdef __cplusplus
}
#endif
#endif /* PAPR_HIP_H */
