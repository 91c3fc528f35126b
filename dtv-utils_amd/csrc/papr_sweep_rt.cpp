// papr_sweep_rt.cpp — host side of the one-sweep mode (papr_sweep.hip): band planning, buffers, launches, the
// bookkeeping a sweep leaves behind and how papr_hip_ccdf is answered from it.

#include "papr_runtime_internal.h"

using namespace papr_rt;

namespace papr_rt {

// Plan the bands for `guess_levels`, size and clear the buffers, upload the LUT.  *reason != PAPR_SWEEP_OK: the guess
// has no band form (or memory is short) and the caller runs the plain pass instead.  `n_shard` sizes the stash,
// `n_launch` (a whole resident shard, or one ingest chunk) the grid.
int sweep_prepare(papr_hip_ctx *ctx, const float *guess_levels, int nlevels, uint64_t n_shard, uint64_t n_launch,
                  SweepRun *run, int *reason)
{
    papr_hip_sweep_info &info = ctx->sweep_info;
    info.band_log2 = ctx->tune.sweep_band_log2 > 0 ? ctx->tune.sweep_band_log2 : kSweepBandLog2;
    *reason = PAPR_SWEEP_NO_BANDS;
    if (nlevels <= 0 || nlevels > PAPR_HIP_MAX_LEVELS)
        return PAPR_OK;
    // widest band (<= the configured width) that has a band form (papr_sweep_bands) and whose edges have a LUT form
    int vblock = 512;
    run->variant = variant_of(ctx, SWEEP);
    (void)papr_sweep_geometry(run->variant, &vblock, &run->tile, &run->stash_lds);
    std::vector<uint32_t> &gkeys = run->gkeys;
    CcdfPlan &bands = run->bands;
    run->half = 0;
    for (int log2w = info.band_log2; log2w >= std::max(info.band_log2 - 3, 8) && !run->half; log2w--) {
        gkeys.assign((size_t)nlevels, 0);
        bands.keys.assign(2 * (size_t)nlevels, 0);
        const int m = papr_sweep_bands(guess_levels, nlevels, log2w, gkeys.data(), bands.keys.data());
        if (m <= 0)
            continue;  // (a narrower band may still fit between crowded thresholds)
        gkeys.resize((size_t)m);
        bands.keys.resize(2 * (size_t)m);
        char keep[sizeof(ctx->err)];
        memcpy(keep, ctx->err, sizeof(keep));
        const bool fits = finish_plan(ctx, &bands, vblock, run->stash_lds) == PAPR_OK && bands.lut;
        memcpy(ctx->err, keep, sizeof(keep));  // not an error of this call: a narrower band or the plain pass follows
        if (fits) {
            run->half = 1u << log2w;
            info.band_log2 = log2w;
        }
    }
    if (!run->half)
        return PAPR_OK;
    // the sweep kernel's LUT has a sentinel cell at either end and its histogram one more (NaN) bin
    bands.P.table_words = 2 * (bands.P.ncells + 2);
    run->nbins = bands.P.nkeys + 2;
    bands.lds_bytes = (size_t)bands.P.table_words * 4 + (size_t)bands.P.copies * run->nbins * 4;
    if (bands.lds_bytes + run->stash_lds > (size_t)papr_ccdf_max_dynamic_lds())
        return PAPR_OK;

    run->blocks = pick_blocks(ctx, SWEEP, n_launch / run->tile);
    // buffers: band histogram with the stash-segment lengths right behind it; stash = 1/4 of the shard's samples
    // (as floats: 1/8 of its bytes), one equal segment per workgroup
    constexpr size_t kMaxSweepBlocks = 65536;
    if (!ctx->d_sweep_hist) {
        const size_t bytes = (2 * (size_t)PAPR_HIP_MAX_LEVELS + 2 + kMaxSweepBlocks) * sizeof(unsigned long long);
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_sweep_hist, bytes));
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_sweep_hist, bytes, hipHostMallocDefault));
    }
    run->seg_cap = std::max<uint64_t>((n_shard / 4 / (uint64_t)run->blocks + 3) & ~3ull, 4096);
    const uint64_t want_stash = run->seg_cap * (uint64_t)run->blocks;
    if (ctx->stash_cap < want_stash) {
        if (ctx->d_stash) HIPCHK(ctx, hipFree(ctx->d_stash));
        ctx->d_stash = nullptr;
        ctx->stash_cap = 0;
        if (hipMalloc((void **)&ctx->d_stash, want_stash * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            ctx->d_stash = nullptr;
            *reason = PAPR_SWEEP_STASH_FULL;
            return PAPR_OK;
        }
        ctx->stash_cap = want_stash;
    }
    int rc = ensure_table(ctx, bands.P.table_words);
    if (rc)
        return rc;
    {
        // lut[0] = below everything, lut[1 + c] = {edges below cell c, the edge inside it or never},
        // lut[ncells + 1] = above every edge; a NaN pattern compares >= 0x7F800001 and lands in the trash bin
        const papr_ccdf_params &P = bands.P;
        uint32_t *tab = ctx->h_table;
        tab[0] = 0;
        tab[1] = kNever;
        uint32_t k = 0;
        for (uint32_t c = 0; c < P.ncells; c++) {
            uint32_t in_cell = kNever;
            const uint32_t below = k;
            if (k < P.nkeys && (bands.keys[k] >> P.shift) == P.cell_lo + c)
                in_cell = bands.keys[k++];
            tab[2 * (c + 1)] = below;
            tab[2 * (c + 1) + 1] = in_cell;
        }
        tab[2 * (P.ncells + 1)] = P.nkeys;
        tab[2 * (P.ncells + 1) + 1] = 0x7F800001u;
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_table, ctx->h_table, (size_t)P.table_words * 4, hipMemcpyHostToDevice,
                                   ctx->stream));
    }
    HIPCHK(ctx, hipMemsetAsync(ctx->d_sweep_hist, 0, ((size_t)run->nbins + (size_t)run->blocks) * sizeof(unsigned long long),
                               ctx->stream));
    *reason = PAPR_SWEEP_OK;
    return PAPR_OK;
}

// One launch of the sweep kernel over [data, data + n): full tiles by the grid, the sub-tile remainder binned by the
// last workgroup (its pass-1 half belongs to papr_stats_finalize).  Histogram and stash segments accumulate over launches.
int sweep_launch(papr_hip_ctx *ctx, const SweepRun &run, const float *data, uint64_t n, uint64_t base_index, size_t slot,
                 int *nrecords)
{
    const uint64_t ntiles = n / run.tile;
    const uint32_t tail = (uint32_t)(n - ntiles * run.tile);
    const int blocks = (int)std::min<uint64_t>((uint64_t)run.blocks, std::max<uint64_t>(ntiles, 1));
    const int map = effective_map(ctx, SWEEP, blocks);
    int rc = ensure_partials(ctx, slot + (size_t)blocks + 1);
    if (rc)
        return rc;
    time_begin(ctx, 3, n * 8);
    papr_launch_sweep(ctx->stream, run.variant, blocks, run.bands.lds_bytes + run.stash_lds, data, ntiles, base_index, map,
                      ctx->d_partials + slot, data + 2 * (n - tail), tail, ctx->d_table, run.bands.P, ctx->d_sweep_hist,
                      ctx->d_stash, ctx->d_sweep_hist + run.nbins, run.seg_cap);
    time_end(ctx);
    HIPCHK(ctx, hipGetLastError());
    *nrecords = blocks;
    return PAPR_OK;
}

// queue the copy of the band histogram + segment lengths to the host (valid after the next stream synchronisation)
int sweep_fetch(papr_hip_ctx *ctx, const SweepRun &run)
{
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_sweep_hist, ctx->d_sweep_hist,
                               ((size_t)run.nbins + (size_t)run.blocks) * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                               ctx->stream));
    return PAPR_OK;
}

// what the sweep decided: samples in even bins above each band; odd bins are exactly the stash
int sweep_collect(papr_hip_ctx *ctx, const SweepRun &run)
{
    papr_hip_sweep_info &info = ctx->sweep_info;
    const unsigned long long *H = ctx->h_sweep_hist;
    uint64_t stash_count = 0, in_bands = 0;
    bool overflow = false;
    for (int b = 0; b < run.blocks; b++) {
        stash_count += H[run.nbins + b];
        overflow = overflow || H[run.nbins + b] > run.seg_cap;
    }
    for (uint32_t b = 1; b < run.nbins; b += 2)
        in_bands += H[b];
    if (in_bands != stash_count)
        return fail(ctx, PAPR_E_INTERNAL, "one-sweep invariant broken: %llu samples binned inside bands, %llu stashed",
                    (unsigned long long)in_bands, (unsigned long long)stash_count);
    const size_t m = run.gkeys.size();
    ctx->sweep_even_above.assign(m, 0);
    uint64_t above = 0;
    for (size_t j = m; j-- > 0;) {
        above += H[2 * j + 2];
        ctx->sweep_even_above[j] = above;
    }
    ctx->sweep_keys = run.gkeys;
    ctx->sweep_half = run.half;
    ctx->sweep_stash_count = stash_count;
    ctx->sweep_seg_cap = run.seg_cap;
    ctx->sweep_nsegs = (uint32_t)run.blocks;
    ctx->sweep_nbins = run.nbins;
    ctx->sweep_overflow = overflow;
    ctx->sweep_valid = true;
    info.swept = 1;
    info.reason = PAPR_SWEEP_OK;
    info.stash_samples = stash_count;
    info.stash_capacity = run.seg_cap * (uint64_t)run.blocks;  // what this sweep could use (one segment per workgroup)
    return PAPR_OK;
}

}  // namespace papr_rt

extern "C" {

// ---- one-sweep mode (papr_sweep.hip) -----------------------------------------------

int papr_hip_estimate(papr_hip_ctx *ctx, papr_stats *est)
{
    if (!ctx || !est)
        return PAPR_E_ARG;
    if (!ctx->loaded)
        return fail(ctx, PAPR_E_STATE, "papr_hip_estimate called before a shard was loaded");
    papr_stats_init(est);
    ctx->sweep_info.estimate_samples = 0;
    if (ctx->have_file_stats) {  // pass 1 already ran while the file streamed in: the "estimate" is the real thing
        est->sum = ctx->file_stats.sum;
        est->n = ctx->file_stats.n;
        return PAPR_OK;
    }
    if (!ctx->resident)
        return fail(ctx, PAPR_E_STATE, "the shard is not resident and has no pass-1 result: reload it");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t ntiles = ctx->n / PAPR_ESTIMATE_TILE_SAMPLES;
    if (ntiles == 0)
        return PAPR_OK;  // nothing to sample: n = 0 tells the caller there is no estimate
    uint64_t ratio = ctx->tune.estimate_ratio > 0 ? (uint64_t)ctx->tune.estimate_ratio : (uint64_t)kEstimateRatio;
    ratio = std::max<uint64_t>(1, std::min<uint64_t>(ratio, ntiles / kEstimateMinTiles));
    const uint64_t ngroups = ntiles / ratio;
    const int blocks = (int)std::min<uint64_t>(ngroups, (uint64_t)ctx->num_cus * 8);
    int rc = ensure_partials(ctx, (size_t)blocks + 1);
    if (rc)
        return rc;
    time_begin(ctx, 4, ngroups * PAPR_ESTIMATE_TILE_SAMPLES * 8);
    papr_launch_estimate(ctx->stream, blocks, ctx->d_iq, ngroups, (uint32_t)ratio, ctx->d_partials);
    time_end(ctx);
    HIPCHK(ctx, hipGetLastError());
    papr_launch_stats_finalize(ctx->stream, nullptr, 0, 0, ctx->d_partials, (uint32_t)blocks, ctx->h_result_dev);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    est->sum = ctx->h_result->sum;
    est->n = ngroups * PAPR_ESTIMATE_TILE_SAMPLES;
    ctx->sweep_info.estimate_samples = est->n;
    return PAPR_OK;
}

int papr_hip_stats_sweep(papr_hip_ctx *ctx, const float *guess_levels, int nlevels, papr_stats *out)
{
    if (!ctx || !out || nlevels < 0 || (nlevels && !guess_levels))
        return PAPR_E_ARG;
    if (!ctx->loaded)
        return fail(ctx, PAPR_E_STATE, "papr_hip_stats_sweep called before a shard was loaded");
    papr_hip_sweep_info &info = ctx->sweep_info;
    info.swept = info.resolved = 0;
    info.stash_samples = 0;
    ctx->sweep_valid = false;
    auto plain = [&](int reason) {
        info.reason = reason;
        return papr_hip_stats(ctx, out);
    };
    if (ctx->have_file_stats || !ctx->resident || ctx->exact)
        return plain(PAPR_SWEEP_MODE);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    SweepRun run;
    int reason = PAPR_SWEEP_OK;
    int rc = sweep_prepare(ctx, guess_levels, nlevels, ctx->n, ctx->n, &run, &reason);
    if (rc)
        return rc;
    if (reason != PAPR_SWEEP_OK)
        return plain(reason);
    int nrec = 0;
    rc = sweep_launch(ctx, run, ctx->d_iq, ctx->n, ctx->base, 0, &nrec);
    if (rc)
        return rc;
    rc = sweep_fetch(ctx, run);
    if (rc)
        return rc;
    const uint32_t tail = (uint32_t)(ctx->n % run.tile);
    rc = finish_stats(ctx, (size_t)nrec, ctx->d_iq + 2 * (ctx->n - tail), tail, ctx->base + ctx->n - tail, out);  // synchronises
    if (rc)
        return rc;
    if (std::isnan(out->sum))  // NaN in the data: the sweep's integer-max trackers do not apply (papr_sweep.hip)
        return plain(PAPR_SWEEP_NO_BANDS);
    return sweep_collect(ctx, run);
}

int papr_hip_get_sweep_info(const papr_hip_ctx *ctx, papr_hip_sweep_info *out)
{
    if (!ctx || !out)
        return PAPR_E_ARG;
    *out = ctx->sweep_info;
    return PAPR_OK;
}

}  // extern "C"

namespace papr_rt {

// Answer papr_hip_ccdf from the last one-sweep pass if every true threshold lies inside the band of its guess:
// samples outside the bands were decided by the sweep, the stash holds the rest.
int resolve_from_sweep(papr_hip_ctx *ctx, const CcdfPlan &plan, const float *levels, int nlevels, uint64_t *counts_above,
                       bool *done)
{
    papr_hip_sweep_info &info = ctx->sweep_info;
    *done = false;
    info.resolved = 0;
    // every true threshold must lie inside one of the bands (papr_sweep_resolve, first without stash counts: a dry run)
    const int band_log2 = __builtin_ctz(ctx->sweep_half);  // of the sweep that left this state behind
    std::vector<uint64_t> stash_above((size_t)nlevels, 0);
    if (!papr_sweep_resolve(ctx->sweep_keys.data(), (int)ctx->sweep_keys.size(), band_log2, ctx->sweep_even_above.data(),
                            levels, nlevels, stash_above.data(), counts_above)) {
        info.reason = PAPR_SWEEP_OUT_OF_BAND;
        return PAPR_OK;
    }
    if (ctx->sweep_overflow) {
        info.reason = PAPR_SWEEP_STASH_FULL;
        return PAPR_OK;
    }
    const uint32_t m = plan.P.nkeys;
    if (ctx->sweep_stash_count) {
        int rc = upload_ccdf_table(ctx, plan);
        if (rc)
            return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_hist, 0, (size_t)(m + 1) * sizeof(unsigned long long), ctx->stream));
        time_begin(ctx, 4, ctx->sweep_stash_count * 4);
        // enough workgroups to fill the chip: every segment is split over `split` of them
        const uint32_t split = std::max<uint32_t>(1, (uint32_t)(ctx->num_cus * 8) / ctx->sweep_nsegs);
        papr_launch_ccdf_power(ctx->stream, (int)(ctx->sweep_nsegs * split), plan.lut, plan.lds_bytes, ctx->d_stash,
                               ctx->d_sweep_hist + ctx->sweep_nbins, ctx->sweep_seg_cap, ctx->sweep_nsegs, split,
                               ctx->d_table, plan.P, ctx->d_hist);
        time_end(ctx);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_hist, ctx->d_hist, (size_t)(m + 1) * sizeof(unsigned long long),
                                   hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    } else {
        memset(ctx->h_hist, 0, (size_t)(m + 1) * sizeof(unsigned long long));
    }
    counts_from_histogram(ctx, plan, nlevels, stash_above.data());  // stash powers above each level ...
    (void)papr_sweep_resolve(ctx->sweep_keys.data(), (int)ctx->sweep_keys.size(), band_log2, ctx->sweep_even_above.data(),
                             levels, nlevels, stash_above.data(), counts_above);  // ... + everything above its band
    info.resolved = 1;
    info.reason = PAPR_SWEEP_OK;
    *done = true;
    return PAPR_OK;
}

}  // namespace papr_rt
