// papr_sweep_rt.cpp — host side of the one-sweep mode (papr_sweep.hip): band planning, buffers, launches, the
// bookkeeping a sweep leaves behind and how papr_hip_ccdf is answered from it.

#include "papr_runtime_internal.h"

using namespace papr_rt;

namespace papr_rt {

// Floats per workgroup segment of the HBM stash (capacity and stride): 1/4 of the workgroup's samples, in whole 1 KiB
// spills, plus a skew — equal shares of a power-of-two shard are a multiple of the memory channels' interleave apart,
// so that all workgroups, filling their segments at the same pace, would write to the same channels at the same time.
static uint64_t stash_segment_floats(uint64_t n_shard, int blocks)
{
    const uint64_t share = std::max<uint64_t>((n_shard / 4 / (uint64_t)blocks + 255) & ~255ull, 4096);
    const uint64_t skew = (uint64_t)std::max(0, env_int("PAPR_STASH_SKEW", kStashSkewFloats)) & ~255ull;
    return share + skew;
}

// Compact LUT (papr_kernels.h) for ascending, distinct band edges: the coarsest cell size that leaves at most two
// edges in any cell.  False: no such form (more than PAPR_LUT2_MAX_EDGES edges, or the table would not fit).
static bool plan_compact_lut(const std::vector<uint32_t> &edges, papr_ccdf_params *P)
{
    const size_t n = edges.size();
    if (n == 0 || n > PAPR_LUT2_MAX_EDGES || edges.front() < 0x00800000u)
        return false;
    for (int s = PAPR_LUT2_MAX_SHIFT; s >= 8; s--) {
        bool ok = true;
        for (size_t i = 0; i + 2 < n && ok; i++)
            ok = (edges[i + 2] >> s) != (edges[i] >> s);
        if (!ok)
            continue;
        const uint32_t c0 = edges.front() >> s, c1 = edges.back() >> s;
        const uint64_t ncells = (uint64_t)c1 - c0 + 1;
        if ((ncells + 2) * 8 > 48 * 1024)
            return false;  // finer cells only get more
        memset(P, 0, sizeof(*P));
        P->shift = (uint32_t)s;
        P->cell_lo = c0;
        P->ncells = (uint32_t)ncells;
        P->nkeys = (uint32_t)n;
        P->table_words = (uint32_t)((2 * (ncells + 2) + 3) & ~3ull);
        return true;
    }
    return false;
}

static void fill_compact_lut(const std::vector<uint32_t> &edges, const papr_ccdf_params &P, uint32_t *tab)
{
    const uint32_t mask = (1u << P.shift) - 1u;
    memset(tab, 0, (size_t)P.table_words * 4);
    tab[0] = PAPR_LUT2_NEVER;  // below everything: 0 edges below, none inside
    tab[1] = kNever;
    uint32_t k = 0;
    for (uint32_t c = 0; c < P.ncells; c++) {
        const uint32_t below = k;
        uint32_t off1 = PAPR_LUT2_NEVER, off2 = kNever;
        if (k < P.nkeys && (edges[k] >> P.shift) == P.cell_lo + c)
            off1 = edges[k++] & mask;
        if (k < P.nkeys && (edges[k] >> P.shift) == P.cell_lo + c)
            off2 = edges[k++] & mask;
        tab[2 * (c + 1)] = (below << PAPR_LUT2_OFF_BITS) | off1;
        tab[2 * (c + 1) + 1] = off2;
    }
    tab[2 * (P.ncells + 1)] = (P.nkeys << PAPR_LUT2_OFF_BITS) | PAPR_LUT2_NEVER;  // above every edge (+Inf, NaN too:
    tab[2 * (P.ncells + 1) + 1] = kNever;                                         // a NaN voids the sweep anyway)
}

// Plan the bands for `guess_levels`, size and clear the buffers, upload the LUT.  *reason != PAPR_SWEEP_OK: the guess
// has no band form (or memory is short) and the caller runs the plain pass instead.  `n_shard` sizes the stash,
// `n_launch` (a whole resident shard, or one ingest chunk) the grid.
int sweep_prepare(papr_hip_ctx *ctx, const float *guess_levels, int nlevels, uint64_t n_shard, uint64_t n_launch,
                  SweepRun *run, int *reason)
{
    papr_hip_sweep_info &info = ctx->sweep_info;
    info.band_log2 = ctx->tune.sweep_band_log2 > 0 ? ctx->tune.sweep_band_log2 : ctx->band_hint > 0 ? ctx->band_hint : kSweepBandLog2;
    *reason = PAPR_SWEEP_NO_BANDS;
    if (nlevels <= 0 || nlevels > PAPR_HIP_MAX_LEVELS)
        return PAPR_OK;
    int vblock = 512, v2_exact = 0;
    run->variant = variant_of(ctx, SWEEP);
    run->v2 = papr_sweep2_geometry(run->variant, &vblock, &run->tile, &run->stash_lds, &v2_exact) == 0;
    run->v3 = !run->v2 && papr_sweep3_geometry(run->variant, &vblock, &run->stash_lds, &v2_exact) == 0;
    if (run->v3)
        run->tile = PAPR_EXACT_SEG_SAMPLES;  // (a wave's segment; exact mode: whole 2048-sample tiles, below)
    // papr_sweep_kernel's LUT has one band edge per cell: its cells are as narrow as the bands, and a 20-octave table of
    // them fits the LDS budget from 2^14 on (a narrower hint is simply not taken up; papr_sweep2_kernel's cells hold two)
    run->lut2 = run->v2 || PAPR_SWEEP_VARIANT_IS_LUT2(run->variant);
    if (!run->lut2 && ctx->tune.sweep_band_log2 <= 0 && info.band_log2 < kSweepBandLog2)
        info.band_log2 = kSweepBandLog2;
    run->exact = v2_exact != 0;
    run->threads = vblock;
    if (!run->v2 && !run->v3)
        (void)papr_sweep_geometry(run->variant, &vblock, &run->tile, &run->stash_lds);
    const size_t lds_cap = (size_t)papr_ccdf_max_dynamic_lds();
    std::vector<uint32_t> &gkeys = run->gkeys;
    CcdfPlan &bands = run->bands;
    run->half = 0;
    // widest band (<= the configured width) that has a band form (papr_sweep_bands) and whose edges have a LUT form
    for (int log2w = info.band_log2; log2w >= std::max(info.band_log2 - 3, 8) && !run->half; log2w--) {
        gkeys.assign((size_t)nlevels, 0);
        bands.keys.assign(2 * (size_t)nlevels, 0);
        const int m = papr_sweep_bands(guess_levels, nlevels, log2w, gkeys.data(), bands.keys.data());
        if (m <= 0)
            continue;  // (a narrower band may still fit between crowded thresholds)
        gkeys.resize((size_t)m);
        bands.keys.resize(2 * (size_t)m);
        bool fits;
        if (run->lut2) {
            fits = plan_compact_lut(bands.keys, &bands.P);
            bands.lut = fits;
        } else {
            char keep[sizeof(ctx->err)];
            memcpy(keep, ctx->err, sizeof(keep));
            fits = finish_plan(ctx, &bands, vblock, run->stash_lds) == PAPR_OK && bands.lut;
            memcpy(ctx->err, keep, sizeof(keep));  // not an error of this call: a narrower band or the plain pass follows
        }
        if (fits) {
            run->half = 1u << log2w;
            info.band_log2 = log2w;
        }
    }
    if (!run->half)
        return PAPR_OK;
    // a persistent workgroup has the CU's LDS to itself: as many histogram copies (up to 4) as fit beside table and stash
    auto copies_that_fit = [&](uint32_t nbins, int waves) {
        int copies = ctx->tune.hist_copies > 0 ? std::min(ctx->tune.hist_copies, waves) : std::min(waves, 4);
        auto lds_of = [&](int c) {
            return (size_t)bands.P.table_words * 4 + (((size_t)c * nbins + 3) & ~(size_t)3) * 4 + run->stash_lds;
        };
        while (copies > 1 && lds_of(copies) > lds_cap - 2048)  // (2 KiB: the kernel's static LDS)
            copies--;
        return lds_of(copies) > lds_cap - 2048 ? 0 : copies;
    };
    if (run->v2 || run->v3) {
        if (run->v3)  // papr_sweep_kernel's table: a sentinel cell at either end, one more (NaN) bin
            bands.P.table_words = 2 * (bands.P.ncells + 2);
        run->nbins = bands.P.nkeys + (run->v3 ? 2 : 1);
        const int waves = vblock / 64;
        const int copies = copies_that_fit(run->nbins, waves);
        if (!copies)
            return PAPR_OK;
        bands.P.copies = (uint32_t)copies;
        bands.lds_bytes = (size_t)bands.P.table_words * 4 + (((size_t)copies * run->nbins + 3) & ~(size_t)3) * 4;
        if (run->exact) {
            if (!ctx->exact || (!ctx->est_groups_valid && !env_int("PAPR_SWEEP2_FAKE_E", 0))) {
                *reason = PAPR_SWEEP_MODE;  // no per-group estimate to speculate the binades from
                return PAPR_OK;
            }
            run->tile = PAPR_EXACT_TILE_SAMPLES;  // the launch covers whole 2048-sample tiles (two segments each)
        }
        const uint64_t nsegs = n_launch / (run->exact ? (uint64_t)PAPR_EXACT_SEG_SAMPLES : run->tile);
        const uint64_t want = ctx->tune.sweep_blocks > 0 ? (uint64_t)ctx->tune.sweep_blocks : (uint64_t)ctx->num_cus;
        run->blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>(want, (nsegs + waves - 1) / waves));
    } else {
        // the sweep kernel's LUT has a sentinel cell at either end and its histogram one more (NaN) bin
        if (!run->lut2)
            bands.P.table_words = 2 * (bands.P.ncells + 2);
        run->nbins = bands.P.nkeys + 2;
        if (run->lut2) {
            const int waves = vblock / 64;
            bands.P.copies = (uint32_t)(ctx->tune.hist_copies > 0 ? std::min(ctx->tune.hist_copies, waves) : std::min(waves, 4));
        }
        if (PAPR_SWEEP_VARIANT_HAS_HIST_SETS(run->variant)) {
            // [bin][copy] histogram: as many copies (a power of two, at most 32) as 16 KiB hold
            uint32_t c = 32;
            while (c > 1 && (size_t)c * run->nbins * 4 > 16 * 1024)
                c >>= 1;
            bands.P.copies = c;
        } else if (PAPR_SWEEP_VARIANT_IS_PERSISTENT(run->variant)) {
            // (finish_plan sized the copies for a CU shared between workgroups: one, next to a 96 KiB stash slice)
            const int copies = copies_that_fit(run->nbins, vblock / 64);
            if (copies)
                bands.P.copies = (uint32_t)copies;
        }
        bands.lds_bytes = (size_t)bands.P.table_words * 4 + (((size_t)bands.P.copies * run->nbins + 3) & ~(size_t)3) * 4;
        if (bands.lds_bytes + run->stash_lds > lds_cap)
            return PAPR_OK;
        run->blocks = pick_blocks(ctx, SWEEP, n_launch / run->tile);
    }

    // buffers: band histogram with the stash-segment lengths right behind it; stash = 1/4 of the shard's samples
    // (as floats: 1/8 of its bytes), one equal segment per workgroup
    constexpr size_t kMaxSweepBlocks = 65536;
    if (!ctx->d_sweep_hist) {
        const size_t bytes = (2 * (size_t)PAPR_HIP_MAX_LEVELS + 2 + 2 * kMaxSweepBlocks + 1) * sizeof(unsigned long long);
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_sweep_hist, bytes));
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_sweep_hist, bytes, hipHostMallocDefault));
    }
    run->seg_cap = stash_segment_floats(n_shard, run->blocks);
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
    if (run->lut2) {
        fill_compact_lut(bands.keys, bands.P, ctx->h_table);
    } else {
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
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_table, ctx->h_table, (size_t)bands.P.table_words * 4, hipMemcpyHostToDevice,
                               ctx->stream));
    // (+ 1: the give-up counter behind the segment arrays)
    HIPCHK(ctx, hipMemsetAsync(ctx->d_sweep_hist, 0,
                               ((size_t)run->nbins + 2 * (size_t)run->blocks + 1) * sizeof(unsigned long long),
                               ctx->stream));
    *reason = PAPR_SWEEP_OK;
    return PAPR_OK;
}

// One launch of the sweep kernel over [data, data + n): full tiles by the grid, the sub-tile remainder binned by the
// last workgroup (its pass-1 half belongs to papr_stats_finalize).  Histogram and stash segments accumulate over launches.
static bool xcd_even_slow(papr_hip_ctx *ctx);

int sweep_launch(papr_hip_ctx *ctx, const SweepRun &run, const float *data, uint64_t n, uint64_t base_index, size_t slot,
                 int *nrecords)
{
    const uint64_t ntiles = n / run.tile;
    const uint32_t tail = (uint32_t)(n - ntiles * run.tile);
    if (run.v2 || run.v3) {
        const int waves = run.threads / 64;
        const uint64_t nsegs = run.exact ? 2 * ntiles : ntiles;
        const int blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)run.blocks, (nsegs + waves - 1) / waves));
        int rc = ensure_partials(ctx, slot + (size_t)blocks + 1);
        if (rc)
            return rc;
        papr_sweep2_params p{};
        p.data = data;
        p.nsegs = nsegs;
        p.base_index = base_index;
        p.out = ctx->d_partials + slot;
        p.tail = data + 2 * (n - tail);
        p.tail_samples = tail;
        p.table = ctx->d_table;
        p.P = run.bands.P;
        p.ghist = ctx->d_sweep_hist;
        p.stash = ctx->d_stash;
        p.seg_slots = ctx->d_sweep_hist + run.nbins;
        p.seg_real = ctx->d_sweep_hist + run.nbins + run.blocks;
        p.seg_cap = run.seg_cap;
        p.gave_up = ctx->d_sweep_hist + run.nbins + 2 * run.blocks;
        p.tile_E_spec = ctx->d_tile_E_spec;
        p.seg_D = ctx->d_seg_D;
        p.seg_offset = (base_index - ctx->base) / PAPR_EXACT_SEG_SAMPLES;
        p.fine_table = run.bands.P.nkeys > 128u ? 1u : 0u;
        p.xcd_skew = xcd_even_slow(ctx) ? 0x80000000u : 0u;  // (which workgroups sit on the odd XCDs: the walk's skew is theirs)
        time_begin_kernel(ctx, 3, n * 8);
        if (run.v3)  // (a persistent workgroup: all of the CU's LDS — what the table leaves goes to the stash slices)
            papr_launch_sweep3(ctx->stream, run.variant, blocks,
                               std::max<size_t>(run.bands.lds_bytes + run.stash_lds, (size_t)papr_ccdf_max_dynamic_lds() - 2048), p);
        else
            papr_launch_sweep2(ctx->stream, run.variant, blocks, run.bands.lds_bytes + run.stash_lds, p);
        time_end_kernel(ctx);
        HIPCHK(ctx, hipGetLastError());
        *nrecords = blocks;
        return PAPR_OK;
    }
    const int blocks = (int)std::min<uint64_t>((uint64_t)run.blocks, std::max<uint64_t>(ntiles, 1));
    const int map = effective_map(ctx, SWEEP, blocks) | (xcd_even_slow(ctx) ? PAPR_MAP_EVEN_SLOW : 0);
    int rc = ensure_partials(ctx, slot + (size_t)blocks + 1);
    if (rc)
        return rc;
    time_begin_kernel(ctx, 3, n * 8);
    papr_launch_sweep(ctx->stream, run.variant, blocks, run.bands.lds_bytes + run.stash_lds, data, ntiles, base_index, map,
                      ctx->d_partials + slot, data + 2 * (n - tail), tail, ctx->d_table, run.bands.P, ctx->d_sweep_hist,
                      ctx->d_stash, ctx->d_sweep_hist + run.nbins, run.seg_cap, ctx->d_sweep_hist + run.nbins + 2 * run.blocks,
                      ctx->d_sweep_hist + run.nbins + run.blocks, nullptr);
    time_end_kernel(ctx);
    HIPCHK(ctx, hipGetLastError());
    *nrecords = blocks;
    return PAPR_OK;
}

// queue the copy of the band histogram + segment lengths to the host (valid after the next stream synchronisation)
int sweep_fetch(papr_hip_ctx *ctx, const SweepRun &run)
{
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_sweep_hist, ctx->d_sweep_hist,
                               ((size_t)(run.seg_off ? run.seg_off : run.nbins) + 2 * (size_t)run.blocks + 1) *
                                   sizeof(unsigned long long),
                               hipMemcpyDeviceToHost, ctx->stream));
    return PAPR_OK;
}

// what the sweep decided: samples in even bins above each band; odd bins are exactly the stash
int sweep_collect(papr_hip_ctx *ctx, const SweepRun &run)
{
    papr_hip_sweep_info &info = ctx->sweep_info;
    const unsigned long long *H = ctx->h_sweep_hist;
    uint64_t stash_count = 0, in_bands = 0;
    bool overflow = false;
    const uint32_t seg_off = run.seg_off ? run.seg_off : run.nbins;
    for (int b = 0; b < run.blocks; b++) {
        stash_count += H[seg_off + run.blocks + b];  // the powers stashed, without padding
        overflow = overflow || H[seg_off + b] > run.seg_cap;
    }
    for (uint32_t b = 1; b < run.nbins; b += 2)
        in_bands += H[b];
    // (a workgroup that gave up — papr_sweep.hip sweep_give_up — leaves both numbers meaningless: `overflow` says so)
    if (in_bands != stash_count && !overflow && !((run.variant >= 60 && run.variant <= 69) || (run.variant >= 90 && run.variant <= 99) || (run.variant >= 120 && run.variant <= 129)))  // (ablation launches: timing only)
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
    ctx->sweep_seg_off = seg_off;
    ctx->sweep_overflow = overflow;
    ctx->sweep_valid = true;
    ctx->spec_recount_valid = false;  // (stats_sweep_fused sets it behind this call)
    info.swept = 1;
    info.kernel_variant = run.variant;
    info.reason = PAPR_SWEEP_OK;
    info.stash_samples = stash_count;
    info.stash_capacity = run.seg_cap * (uint64_t)run.blocks;  // what this sweep could use (one segment per workgroup)
    info.gave_up = (uint32_t)std::min<unsigned long long>(H[seg_off + 2 * run.blocks], 0xFFFFFFFFull);
    return PAPR_OK;
}

// A whole step's GPU work as ONE uninterrupted sequence of launches and one wait, for a shard without peers (nothing
// crosses an exchange), resident, with the default kernel choice — instead of estimate / wait / host guess + LUT +
// upload / sweep / wait / host table + LUT + upload / recount / wait:
//   estimate kernel -> papr_guess_bands_kernel (the host half of the speculation, on the device) -> [exact-sum mode:
//   binade speculation] -> sweep kernel (its table geometry read from device memory) -> finalize -> [exact-sum mode:
//   classification, redo, groups, program] -> papr_true_table_kernel (the reference's level table with the device's
//   libm: a speculation the host checks bit for bit) -> stash recount on that table.
// Anything else: *done stays false and the caller takes the host path.  The TRUE level table is the host's
// (papr_levels) as before; what moved to the device are guesses.
// With peers over an in-stream exchange (RCCL) the same sequence carries the step's three exchanges as collectives ON the
// stream, between the kernels that produce and consume them — still one wait:
//   estimate -> record of it -> ALL-GATHER -> guess from every shard's record -> sweep -> finalize -> ALL-GATHER of the
//   pass-1 records -> ordered merge -> the FILE's level table -> recount -> [sweep bins | recount bins | flags] ->
//   ALL-REDUCE -> host.
// Every rank then holds the same global numbers and takes the same decisions from them.  (Exact-sum mode with peers
// keeps the host path: its programs are exchanged and chained on the host.)
// Whether the EVEN workgroups of this context's stream sit on the odd — slower-reading — XCDs.  Workgroup b of a launch runs on
// XCD (b + first) mod 8, and `first` is the queue's, not the device's (6 in a plain process, 5 once RCCL has queues of its
// own: profiles/r05_xcd_skew.txt): asked of an 8-workgroup launch the first time, and from then on read out of every
// sweep's record.  PAPR_XCD_PARITY=0|1 pins the answer (measurements).
// REPRODUCIBILITY: the parity is probed from the hardware (it differs between a plain process and one with RCCL's
// queues beside the step's) and decides which workgroups sit a round out — coverage and every integer result are the same
// either way, and so is the exact-sum mode's sum (the reference's accumulator, whatever the walk); but the TREE sum
// (PAPR_EXACT_SUM=0 / papr_hip_set_exact(0)) groups the workgroups' double sums by the walk, so its last bits can differ
// between two runs of the same input that probed different parities.  PAPR_XCD_PARITY=0|1 pins the parity — and with it
// the tree sum, bit for bit — at the cost of the skew sitting on the wrong XCDs half of the time (+2 % kernel time).
static bool xcd_even_slow(papr_hip_ctx *ctx)
{
    static const int pinned = env_int("PAPR_XCD_PARITY", -1);
    if (pinned >= 0)
        return pinned != 0;
    if (ctx->xcd_first < 0) {
        unsigned long long first = 0;
        papr_launch_xcd_probe(ctx->stream, ctx->d_nan_key);
        if (hipMemcpyAsync(&first, ctx->d_nan_key, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        ctx->xcd_first = (int)(first & 0xF);
    }
    return (ctx->xcd_first & 1) != 0;
}

int stats_sweep_fused(papr_hip_ctx *ctx, papr_exchange *x, int graph, double max_db, float spoil, papr_stats *out, bool *done,
                      PeerStep *peer)
{
    *done = false;
    ctx->peer_global = false;
    const bool exact = ctx->exact;
    const bool peers = x && !papr_exchange_is_identity(x);
    if (peers && (!xch_in_stream(x, ctx) || !peer))
        return PAPR_OK;
    // (a chosen kernel form goes through the host path, except the forms that share the default's launch shape: one
    // persistent workgroup per CU)
    const int chosen = ctx->tune.sweep_variant > 0 ? variant_of(ctx, SWEEP) : -1;
    int dummy_threads = 0;
    size_t dummy_lds = 0;
    int chosen_exact3 = 0;
    const bool chosen_v3 = chosen >= 0 && papr_sweep3_geometry(chosen, &dummy_threads, &dummy_lds, &chosen_exact3) == 0;
    const bool chosen_ok = chosen < 0 || (exact ? (chosen_v3 && chosen_exact3) || chosen == 56
                                                : (chosen_v3 && !chosen_exact3) || PAPR_SWEEP_VARIANT_IS_PERSISTENT(chosen));
    const uint64_t ntiles_est = ctx->n / PAPR_ESTIMATE_TILE_SAMPLES;
    const uint32_t nl = graph ? (uint32_t)(max_db * 10.0) + 1u : (uint32_t)max_db + 1u;
    const bool eligible = ctx->loaded && ctx->resident && !ctx->have_file_stats && chosen_ok && ctx->tune.sweep_map <= 0 &&
                          ctx->tune.sweep_blocks <= 0 && env_int(exact ? "PAPR_FUSED_EXACT" : "PAPR_FUSED_GUESS", 1) &&
                          ntiles_est != 0 && nl <= PAPR_GUESS_MAX_BANDS && max_db >= 0 &&
                          ctx->n / (exact ? (uint64_t)PAPR_EXACT_TILE_SAMPLES : 8192ull) != 0;
    if (!peers && !eligible)
        return PAPR_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // With peers the rest of this function queues collectives every rank must enter, so the ranks AGREE first that all of
    // them take this path (one host all-reduce) — and everything that can fail for a local reason (the buffers, the
    // stash) is allocated in front of that agreement and folded into it: behind it a rank leaves only with an error, and
    // then cancels the exchange so that nobody waits for it (papr_exchange_abort).  The agreement is repeated whenever
    // the exchange, the mode or the shard state may have changed: the key holds the exchange and what the step depends
    // on, and every call that changes shard state (load / upload / adopt / generate / set_tuning / set_exact) clears it —
    // on every rank of an SPMD caller alike, whether or not the values changed on that rank.
    const uint64_t agree_key = ((uint64_t)(uintptr_t)x * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(peers ? xch_world(x) : 1) << 48) ^
                               ((uint64_t)graph << 62) ^ ((uint64_t)exact << 61) ^ ctx->peer_epoch ^ 1ull;
    const bool agreed = peers && ctx->peer_agreed_key == agree_key;
    if (agreed && !ctx->peer_agreed_ok)
        return PAPR_OK;
    bool local_ok = eligible;  // (folded into the agreement; a rank that is not eligible still takes part in it)
    auto leave = [&](int rc) {  // behind the agreement: no rank-local way out but with an error, and the peers are released
        if (peers && rc != PAPR_OK)
            papr_exchange_abort(x);
        return rc;
    };
    papr_hip_sweep_info &info = ctx->sweep_info;
    // ---- geometry: the default kernel of the mode; its table is what the device builds ----
    SweepRun run;
    int vblock = 0;
    uint64_t ntiles = 0, nsegs_launch = 0, ratio = 1, ngroups = 0;
    int waves = 0, est_blocks = 0;
    bool by_segments = false;
    constexpr uint32_t kBinsMax = 2 * PAPR_GUESS_MAX_BANDS + 2;
    constexpr uint32_t kCopies = 4;
    constexpr uint32_t kXvecWords = kBinsMax + (PAPR_TRUE_MAX_LEVELS + 1) + PAPR_XVEC_FLAGS;
    size_t table_lds = 0;
    uint32_t table_cap_words = 0, soft_lds = 0;
    papr_partial *est_partials = nullptr;
    const uint32_t world = peers ? (uint32_t)xch_world(x) : 1u, my_rank = peers ? (uint32_t)xch_rank(x) : 0u;
    papr_est_record *d_est_mine = nullptr, *d_est_all = nullptr;
    papr_partial *d_recs_all = nullptr, *d_total = nullptr;
    unsigned long long *d_n_total = nullptr, *d_xvec = nullptr, *d_xvec_sum = nullptr;
    double *group_sums = nullptr;
    // everything that can say "not this way" or fail for a reason of this rank's own; `local_ok` = this rank can go on
    auto prepare = [&]() -> int {
        if (!local_ok)
            return PAPR_OK;
        local_ok = false;
        if (exact) {
            int v2_exact = 0;
            run.variant = chosen >= 0 ? chosen : kSweepExactVariant;
            if (papr_sweep3_geometry(run.variant, &vblock, &run.stash_lds) == 0) {
                run.v3 = run.exact = true;  // papr_sweep_kernel's one-edge table, papr_sweep2_kernel's segments and launch
            } else {
                if (papr_sweep2_geometry(run.variant, &vblock, &run.tile, &run.stash_lds, &v2_exact) != 0 || !v2_exact)
                    return PAPR_OK;
                run.v2 = run.lut2 = run.exact = true;
            }
            run.tile = PAPR_EXACT_TILE_SAMPLES;  // the launch covers whole 2048-sample tiles (two segments each)
        } else {
            run.variant = chosen >= 0 ? chosen : kSweepVariant;
            run.lut2 = PAPR_SWEEP_VARIANT_IS_LUT2(run.variant);
            if (papr_sweep3_geometry(run.variant, &vblock, &run.stash_lds) == 0) {
                run.v3 = true;  // the plain form of papr_sweep3_kernel: wave-private 1024-sample segments
                run.tile = PAPR_EXACT_SEG_SAMPLES;
            } else if (papr_sweep_geometry(run.variant, &vblock, &run.tile, &run.stash_lds) != 0) {
                return PAPR_OK;
            }
        }
        run.threads = vblock;
        ntiles = ctx->n / run.tile;
        if (ntiles == 0)
            return PAPR_OK;
        waves = vblock / 64;
        by_segments = exact || run.v3;  // launched through papr_sweep2_params: one persistent workgroup per CU
        nsegs_launch = exact ? 2 * ntiles : ntiles;
        if (by_segments)
            run.blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)ctx->num_cus, (nsegs_launch + waves - 1) / waves));
        else
            run.blocks = pick_blocks(ctx, SWEEP, ntiles);
        const size_t lds_cap = (size_t)papr_ccdf_max_dynamic_lds();
        // LDS for table + histogram copies: what the launch is given, and what the device-side plan has to fit into
        table_lds = by_segments ? (lds_cap - 2048 > run.stash_lds ? lds_cap - 2048 - run.stash_lds : 0)
                                : (size_t)((PAPR_SWEEP_VARIANT_IS_LUT2(run.variant) ? 48 : 40) * 1024 + 32) +
                                      (((size_t)kCopies * kBinsMax + 3) & ~(size_t)3) * 4;  // (one-edge table: <= 40 KiB)
        if (table_lds < 16 * 1024 || table_lds + run.stash_lds > lds_cap)
            return PAPR_OK;
        table_cap_words = by_segments ? (uint32_t)(table_lds / 4) : 48 * 1024 / 4 + 8;
        // (the workgroup's share of the CU's LDS: 80 bytes per thread where several are resident, everything for a persistent one)
        soft_lds = by_segments ? (uint32_t)table_lds
                   : PAPR_SWEEP_VARIANT_IS_PERSISTENT(run.variant)
                       ? (uint32_t)(lds_cap - 2048 - run.stash_lds)
                       : (uint32_t)std::max<long long>(0, (long long)vblock * 80 - (long long)run.stash_lds);
        run.seg_off = kBinsMax;
        // ---- estimate geometry (as papr_hip_estimate) ----
        ratio = ctx->tune.estimate_ratio > 0 ? (uint64_t)ctx->tune.estimate_ratio : (uint64_t)kEstimateRatio;
        ratio = std::max<uint64_t>(1, std::min<uint64_t>(ratio, ntiles_est / kEstimateMinTiles));
        ngroups = ntiles_est / ratio;
        est_blocks = (int)std::min<uint64_t>(ngroups, (uint64_t)ctx->num_cus * 8);
        // ---- buffers ----
        int rc = ensure_partials(ctx, (size_t)run.blocks + 1 + (size_t)est_blocks + 1);
        if (rc)
            return rc;
        est_partials = ctx->d_partials + run.blocks + 1;
        if (!ctx->d_est_sq) {
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_est_sq, (size_t)ctx->num_cus * 8 * sizeof(double)));
            HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_est_sq, (size_t)ctx->num_cus * 8 * sizeof(double), hipHostMallocDefault));
        }
        if (!ctx->d_result_copy) {
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_result_copy, sizeof(papr_partial)));
        }
        if (!ctx->d_true) {
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_true, sizeof(papr_true_out)));
            HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_true, sizeof(papr_true_out), hipHostMallocMapped));
            HIPCHK(ctx, hipHostGetDevicePointer((void **)&ctx->h_true_dev, ctx->h_true, 0));
        }
        if (!ctx->d_pow_tab) {  // the host libm's pow(10, x_j) of both level tables, for the two table-building kernels
            int cnt = 0;
            const double *t0 = papr_level_pow_table(0, &cnt), *t1 = papr_level_pow_table(1, &cnt);
            static_assert(PAPR_POW_TABLE == 2048, "papr_host.c: PAPR_POW_CACHE");
            if (cnt == PAPR_POW_TABLE) {
                HIPCHK(ctx, hipMalloc((void **)&ctx->d_pow_tab, 2 * (size_t)PAPR_POW_TABLE * sizeof(double)));
                HIPCHK(ctx, hipMemcpy(ctx->d_pow_tab, t0, (size_t)PAPR_POW_TABLE * sizeof(double), hipMemcpyHostToDevice));
                HIPCHK(ctx, hipMemcpy(ctx->d_pow_tab + PAPR_POW_TABLE, t1, (size_t)PAPR_POW_TABLE * sizeof(double), hipMemcpyHostToDevice));
            }
        }
        if (!ctx->d_guess) {
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_guess, sizeof(papr_guess_out)));
            HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_guess, sizeof(papr_guess_out), hipHostMallocMapped));
            HIPCHK(ctx, hipHostGetDevicePointer((void **)&ctx->h_guess_dev, ctx->h_guess, 0));
        }
        // peers: device scratch [my estimate record | all of them | all pass-1 records | the merged one | n, before | vector | reduced vector]
        if (peers) {
            const size_t a_est = 64, a_all = (size_t)world * sizeof(papr_est_record), a_recs = (size_t)world * sizeof(papr_partial);
            auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
            const size_t need = up(a_est) + up(a_all) + up(a_recs) + up(sizeof(papr_partial)) + 256 + 2 * up((size_t)kXvecWords * 8);
            if (ctx->peer_cap < need) {
                if (ctx->d_peer) HIPCHK(ctx, hipFree(ctx->d_peer));
                ctx->d_peer = nullptr;
                ctx->peer_cap = 0;
                HIPCHK(ctx, hipMalloc((void **)&ctx->d_peer, need));
                ctx->peer_cap = need;
            }
            if (!ctx->h_peer) {
                HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_peer, sizeof(papr_peer_out), hipHostMallocMapped));
                HIPCHK(ctx, hipHostGetDevicePointer((void **)&ctx->h_peer_dev, ctx->h_peer, 0));
                HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_xvec, (size_t)kXvecWords * 8, hipHostMallocDefault));
            }
            unsigned char *q = ctx->d_peer;
            d_est_mine = (papr_est_record *)q;          q += up(a_est);
            d_est_all = (papr_est_record *)q;           q += up(a_all);
            d_recs_all = (papr_partial *)q;             q += up(a_recs);
            d_total = (papr_partial *)q;                q += up(sizeof(papr_partial));
            d_n_total = (unsigned long long *)q;        q += 256;   // [0] the file's length, [1] (a double) the sum in front of this shard
            d_xvec = (unsigned long long *)q;           q += up((size_t)kXvecWords * 8);
            d_xvec_sum = (unsigned long long *)q;
            if (exact) {
                // the in-stream exchange of the sum programs: one slot per rank, sized from the LARGEST shard the ranks
                // could hold under this geometry — every rank computes the same number from ctx->xprog_slot_samples,
                // which the agreement settles (the maximum of the ranks' shard sizes)
                if ((int)ctx->xprog_sizes.size() != (int)world || ctx->xprog_slot == 0) {
                    ctx->xprog_sizes.assign(world, ctx->xprog_slot);
                    ctx->xprog_world = 0;
                }
                ctx->xprog_offs.assign(world + 1, 0);
                for (uint32_t r = 0; r < world; r++)
                    ctx->xprog_offs[r + 1] = ctx->xprog_offs[r] + ctx->xprog_sizes[r];
                const size_t mine = ctx->xprog_sizes[my_rank], all = ctx->xprog_offs[world];
                if (mine == 0)
                    return PAPR_OK;  // (no slot size agreed yet: not this way)
                if (ctx->xprog_cap_mine < mine) {
                    if (ctx->d_xprog) HIPCHK(ctx, hipFree(ctx->d_xprog));
                    ctx->d_xprog = nullptr;
                    ctx->xprog_cap_mine = 0;
                    HIPCHK(ctx, hipMalloc((void **)&ctx->d_xprog, mine));
                    ctx->xprog_cap_mine = mine;
                }
                if (ctx->xprog_cap_all < all) {
                    if (ctx->d_xprog_all) HIPCHK(ctx, hipFree(ctx->d_xprog_all));
                    if (ctx->h_xprog_all) HIPCHK(ctx, hipHostFree(ctx->h_xprog_all));
                    ctx->d_xprog_all = ctx->h_xprog_all = ctx->h_xprog_all_dev = nullptr;
                    ctx->xprog_cap_all = 0;
                    HIPCHK(ctx, hipMalloc((void **)&ctx->d_xprog_all, all));
                    HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_xprog_all, all, hipHostMallocMapped));
                    HIPCHK(ctx, hipHostGetDevicePointer((void **)&ctx->h_xprog_all_dev, ctx->h_xprog_all, 0));
                    ctx->xprog_cap_all = all;
                }
                ctx->xprog_world = (int)world;
            }
        }
        constexpr size_t kMaxSweepBlocks = 65536;
        if (!ctx->d_sweep_hist) {
            const size_t bytes = (2 * (size_t)PAPR_HIP_MAX_LEVELS + 2 + 2 * kMaxSweepBlocks + 1) * sizeof(unsigned long long);
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_sweep_hist, bytes));
            HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_sweep_hist, bytes, hipHostMallocDefault));
        }
        if (!ctx->h_sweep_hist_dev)
            HIPCHK(ctx, hipHostGetDevicePointer((void **)&ctx->h_sweep_hist_dev, ctx->h_sweep_hist, 0));
        run.seg_cap = stash_segment_floats(ctx->n, run.blocks);
        const uint64_t want_stash = run.seg_cap * (uint64_t)run.blocks;
        if (ctx->stash_cap < want_stash) {
            if (ctx->d_stash) HIPCHK(ctx, hipFree(ctx->d_stash));
            ctx->d_stash = nullptr;
            ctx->stash_cap = 0;
            if (hipMalloc((void **)&ctx->d_stash, want_stash * sizeof(float)) != hipSuccess) {
                (void)hipGetLastError();
                ctx->d_stash = nullptr;
                return PAPR_OK;  // (no room for the stash: the host path reports it)
            }
            ctx->stash_cap = want_stash;
        }
        rc = ensure_table(ctx, std::max<uint32_t>(table_cap_words, 48 * 1024 / 4 + 8));
        if (rc)
            return rc;
        if (exact) {
            rc = ensure_exact_buffers(ctx);
            if (rc)
                return rc;
            if (ctx->est_groups_cap < ngroups) {
                if (ctx->d_est_groups) HIPCHK(ctx, hipFree(ctx->d_est_groups));
                ctx->d_est_groups = nullptr;
                ctx->est_groups_cap = 0;
                const uint64_t cap = std::max<uint64_t>(ngroups, 4096);
                HIPCHK(ctx, hipMalloc((void **)&ctx->d_est_groups, cap * 5 * sizeof(double)));  // 4 wave sums + 1 prefix per group
                ctx->est_groups_cap = cap;
            }
            group_sums = ctx->d_est_groups;
            if (peers) {
                rc = reserve_exact_lists(ctx);
                if (rc)
                    return rc;
            }
        }
        local_ok = true;
        return PAPR_OK;
    };
    if (peers && !agreed) {
        // the agreement (and, exact-sum mode, the slot size of the program exchange: from the LARGEST shard, so that every
        // rank allocates the same): [ranks that can, largest shard]
        uint64_t nmax = ctx->n;
        if (exact) {
            std::vector<uint64_t> all((size_t)world);
            const uint64_t mine = ctx->n;
            const int xrc = xch_allgather_host(x, &mine, all.data(), sizeof(uint64_t));
            if (xrc)
                return fail(ctx, xrc, "exchange: %s", papr_exchange_last_error(x));
            nmax = *std::max_element(all.begin(), all.end());
        }
        const size_t new_slot = exact ? exact_program_slot_bytes(nmax) : 0;
        if (new_slot != ctx->xprog_slot)
            ctx->xprog_sizes.clear();  // (slots grown under another agreement belong to it: every rank starts again alike)
        ctx->xprog_slot = new_slot;
        int prc = prepare();
        uint64_t ok = (prc == PAPR_OK && local_ok) ? 1u : 0u;
        const int xrc = papr_exchange_counts(x, &ok, 1);
        if (xrc)
            return fail(ctx, xrc, "exchange: %s", papr_exchange_last_error(x));
        ctx->peer_agreed_key = agree_key;
        ctx->peer_agreed_ok = ok == (uint64_t)world;
        if (prc)
            return prc;  // (an allocation failed here; the peers heard "no" and take the host path without this rank)
        if (!ctx->peer_agreed_ok)
            return PAPR_OK;
    } else {
        const int prc = prepare();
        if (prc)
            return leave(prc);
        if (!local_ok) {
            if (!peers)
                return PAPR_OK;
            // (agreed earlier, and now this rank cannot: its state changed behind the agreement's back)
            return leave(fail(ctx, PAPR_E_STATE, "the shard's state changed since the ranks agreed on the single-wait step"));
        }
    }
    info.swept = info.resolved = 0;
    info.stash_samples = 0;
    ctx->sweep_valid = false;
    ctx->exact_swept = false;
    ctx->exact_program_launched = false;
    ctx->est_groups_valid = false;
    ctx->exact_program_launched = false;
    // ---- the launches ----
    // (from here on a failure of this rank's own must release the peers: leave())
#define XCHK(call)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return leave(fail(ctx, PAPR_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)));   \
    } while (0)
    int rc = PAPR_OK;
    ctx->trace.mark("prepared");
    time_begin_kernel(ctx, 4, ngroups * PAPR_ESTIMATE_TILE_SAMPLES * 8);
    papr_launch_estimate(ctx->stream, est_blocks, ctx->d_iq, ngroups, (uint32_t)ratio, est_partials, group_sums, ctx->d_est_sq);
    time_end_kernel(ctx);
    XCHK(hipGetLastError());
    const int band_override = ctx->tune.sweep_band_log2 > 0 ? ctx->tune.sweep_band_log2 : 0;
    if (peers) {  // exchange 1a, in the stream: every shard's estimate record
        papr_launch_est_record(ctx->stream, est_partials, ctx->d_est_sq, (uint32_t)est_blocks, ngroups,
                               ngroups * PAPR_ESTIMATE_TILE_SAMPLES, ctx->n, (uint32_t)ratio, ctx->shard_flags, d_est_mine);
        XCHK(hipGetLastError());
        rc = xch_allgather_dev(x, ctx, d_est_mine, d_est_all, sizeof(papr_est_record));
        if (rc)
            return leave(fail(ctx, rc, "exchange: %s", papr_exchange_last_error(x)));
    }
    papr_launch_guess_bands(ctx->stream, est_partials, ctx->d_est_sq, (uint32_t)est_blocks, ngroups,
                            ngroups * PAPR_ESTIMATE_TILE_SAMPLES, ctx->n, (uint32_t)ratio, graph, (float)max_db, spoil,
                            band_override, kCopies, run.lut2 ? 1 : 0, soft_lds, ctx->d_table, table_cap_words, ctx->d_guess,
                            ctx->h_guess_dev, ctx->d_sweep_hist,
                            kBinsMax + 2u * (uint32_t)run.blocks + 1u,  // (also clears the sweep's bins and segment counters)
                            d_est_all, peers ? world : 0u, my_rank,
                            // (exact-sum mode, no peers: the scan half of the binade speculation runs beside the guess)
                            exact && !peers ? ctx->d_est_groups : nullptr, exact && !peers ? ngroups : 0, (double)ratio,
                            exact && !peers ? ctx->d_est_groups + 4 * ctx->est_groups_cap : nullptr, ctx->d_pow_tab);
    XCHK(hipGetLastError());
    const uint32_t tail = (uint32_t)(ctx->n - ntiles * run.tile);
    const bool even_slow = xcd_even_slow(ctx);
    papr_ccdf_params none{};
    if (by_segments) {
        if (exact) {
            // every tile's running-sum binade, speculated from the estimate's per-group sums (nothing in front of this shard)
            ctx->est_ngroups = ngroups;
            ctx->est_ratio = ratio;
            ctx->est_groups_valid = true;
            // (peers: ... what papr_guess_bands_kernel made of the shards' estimate records in front of this one)
            papr_launch_exact_spec(ctx->stream, ctx->d_est_groups, ngroups, (uint32_t)ratio, (double)ratio, 0.0,
                                   ctx->d_est_groups + 4 * ctx->est_groups_cap, ctx->n / PAPR_EXACT_TILE_SAMPLES, ctx->d_tile_E_spec,
                                   peers ? &ctx->d_guess->est_before : nullptr, /*scan_done=*/!peers && ngroups != 0);
            XCHK(hipGetLastError());
        }
        papr_sweep2_params p{};
        p.data = ctx->d_iq;
        p.nsegs = nsegs_launch;
        p.base_index = ctx->base;
        p.out = ctx->d_partials;
        p.tail = ctx->d_iq + 2 * (ctx->n - tail);
        p.tail_samples = tail;
        p.table = ctx->d_table;
        p.P = none;
        p.Pdev = &ctx->d_guess->P;
        p.ghist = ctx->d_sweep_hist;
        p.stash = ctx->d_stash;
        p.seg_slots = ctx->d_sweep_hist + kBinsMax;
        p.seg_real = ctx->d_sweep_hist + kBinsMax + run.blocks;
        p.gave_up = ctx->d_sweep_hist + kBinsMax + 2 * run.blocks;
        p.seg_cap = run.seg_cap;
        p.tile_E_spec = ctx->d_tile_E_spec;
        p.seg_D = ctx->d_seg_D;
        p.seg_offset = 0;
        p.fine_table = graph ? 1u : 0u;  // (the table is planned on the device: 301 bands for -g, 31 otherwise)
        p.xcd_skew = even_slow ? 0x80000000u : 0u;
        time_begin_kernel(ctx, 3, ctx->n * 8);
        ctx->sweep_blocks_last = (uint32_t)run.blocks;
        if (run.v3)
            papr_launch_sweep3(ctx->stream, run.variant, run.blocks, table_lds + run.stash_lds, p);
        else
            papr_launch_sweep2(ctx->stream, run.variant, run.blocks, table_lds + run.stash_lds, p);
        time_end_kernel(ctx);
    } else {
        const int map = effective_map(ctx, SWEEP, run.blocks) | (graph ? 0x80 : 0) | (even_slow ? PAPR_MAP_EVEN_SLOW : 0);  // (0x80: the 0.1 dB table — its own XCD skew)
        time_begin_kernel(ctx, 3, ctx->n * 8);
        ctx->sweep_blocks_last = (uint32_t)run.blocks;
        papr_launch_sweep(ctx->stream, run.variant, run.blocks, table_lds + run.stash_lds, ctx->d_iq, ntiles, ctx->base, map,
                          ctx->d_partials, ctx->d_iq + 2 * (ctx->n - tail), tail, ctx->d_table, none, ctx->d_sweep_hist,
                          ctx->d_stash, ctx->d_sweep_hist + kBinsMax, run.seg_cap, ctx->d_sweep_hist + kBinsMax + 2 * run.blocks,
                          ctx->d_sweep_hist + kBinsMax + run.blocks, &ctx->d_guess->P);
        time_end_kernel(ctx);
    }
    XCHK(hipGetLastError());
    // Exact-sum mode, alone: the sum program FIRST — the pairs the sweep built, the true prefix, the refuted tiles rebuilt, groups,
    // gather: nothing of it needs pass 1's record — so that the host's replay of it (0.07 ms) has the finalize kernel, the
    // table, the recount and the copy to run beside (round 6: the chain used to sit behind the finalize kernel, and 40 us of
    // the replay stood in the open)
    const bool exact_chain_first = exact && !peers && env_int("PAPR_EXACT_CHAIN_FIRST", 1) != 0;
    auto queue_exact_chain = [&]() -> int {
        const int xrc = run_exact_swept(ctx, 0.0, ctx->n);
        if (xrc)
            return xrc;
        ctx->exact_program_launched = true;
        ctx->exact_program_before = 0.0;
        return PAPR_OK;
    };
    if (exact_chain_first) {
        rc = queue_exact_chain();
        if (rc)
            return leave(rc);
    }
    // pass 1's record (tail + merge of the workgroups' records) — and, on the way, the sweep's bins and segment counters
    // into mapped host memory (sweep_fetch without a copy in the stream; PAPR_FUSED_COPIES=1 keeps the copies) ...
    const bool by_kernel = env_int("PAPR_FUSED_COPIES", 0) == 0;
    if (!by_kernel) {
        rc = sweep_fetch(ctx, run);
        if (rc)
            return leave(rc);
    }
    papr_launch_stats_finalize(ctx->stream, ctx->d_iq + 2 * (ctx->n - tail), tail, ctx->base + ctx->n - tail, ctx->d_partials,
                               (uint32_t)run.blocks, ctx->h_result_dev, ctx->d_result_copy, ctx->d_sweep_hist,
                               ctx->h_sweep_hist_dev, by_kernel ? kBinsMax + 2u * (uint32_t)run.blocks + 1u : 0u);
    XCHK(hipGetLastError());
    if (exact && !peers && !exact_chain_first) {
        // ... the sum program from the pairs the sweep built (true prefix, the refuted tiles rebuilt, groups, gather) ...
        rc = queue_exact_chain();
        if (rc)
            return leave(rc);
    }
    // ... and, speculatively, what follows from the record: the reference's level table with the device's libm, the
    // recount LUT for it, and the recount of the stash — so that the step's second half needs no launch + wait round
    // trip either (resolve_from_sweep takes the histogram if the host's own table turns out to be this one, bit for bit)
    constexpr uint32_t kTrueCopies = 4;
    constexpr uint32_t true_soft = 20 * 1024;  // LDS the recount is launched with (as the host path: table + histogram copies)
    ctx->h_true->ok = 0;
    if (peers) {  // exchange 1b, in the stream: the shards' pass-1 records, folded in rank (= file) order on every rank
        rc = xch_allgather_dev(x, ctx, ctx->d_result_copy, d_recs_all, sizeof(papr_partial));
        if (rc)
            return leave(fail(ctx, rc, "exchange: %s", papr_exchange_last_error(x)));
        papr_launch_record_merge(ctx->stream, d_recs_all, d_est_all, world, my_rank, d_total, d_n_total, ctx->h_peer_dev);
        XCHK(hipGetLastError());
        if (exact) {
            // exact-sum mode: this shard's sum program from the pairs the sweep built, classified against the TRUE prefix —
            // which starts at the merged records' sum in front of this shard, known on the device only (d_n_total[1]) —
            // into this rank's slot; exchange 1c, in the stream: every rank's slot; then the used bytes of all of them into
            // mapped host memory, where the host replays them in rank (= file) order while the stream goes on
            // (papr_hip_analyze: overlap_work).  A program that outgrew its slot, or is not final, is marked: every rank
            // sees that and all of them exchange on the host afterwards.
            rc = run_exact_swept(ctx, 0.0, 0, reinterpret_cast<const double *>(d_n_total) + 1, d_n_total, ctx->d_xprog,
                                 ctx->xprog_sizes[my_rank]);
            if (rc)
                return leave(rc);
            rc = xch_allgatherv_dev(x, ctx, ctx->d_xprog, ctx->d_xprog_all, ctx->xprog_sizes.data(), ctx->xprog_offs.data());
            if (rc)
                return leave(fail(ctx, rc, "exchange: %s", papr_exchange_last_error(x)));
            papr_xprog_layout lay{};
            lay.world = world;
            for (uint32_t r = 0; r <= world; r++)
                lay.offs[r] = ctx->xprog_offs[r];
            papr_launch_exact_programs_to_host(ctx->stream, ctx->d_xprog_all, lay, ctx->h_xprog_all_dev);
            XCHK(hipGetLastError());
            rc = mark_program_ready(ctx);
            if (rc)
                return leave(rc);
            ctx->xprog_ready = true;
            ctx->program_view = ctx->h_xprog_all + ctx->xprog_offs[my_rank];
            ctx->exact_program_launched = true;  // (exact_program_before: known after the wait)
        }
    }
    papr_launch_true_table(ctx->stream, peers ? d_total : ctx->d_result_copy, ctx->n, graph, kTrueCopies, true_soft, ctx->d_table,
                           std::max<uint32_t>(table_cap_words, 48 * 1024 / 4 + 8), ctx->d_true, ctx->h_true_dev, ctx->d_hist,
                           PAPR_TRUE_MAX_LEVELS + 1,  // (also clears the recount's bins)
                           ctx->d_sweep_hist + kBinsMax + 2 * run.blocks, peers ? d_n_total : nullptr, ctx->d_pow_tab);
    XCHK(hipGetLastError());
    {
        time_begin_kernel(ctx, 4, 0);
        papr_launch_ccdf_power(ctx->stream, ctx->num_cus, true, true_soft, ctx->d_stash, ctx->d_sweep_hist + kBinsMax,
                               run.seg_cap, (uint32_t)run.blocks, ctx->d_table, none, ctx->d_hist, &ctx->d_true->P);
        time_end_kernel(ctx);
        XCHK(hipGetLastError());
    }
    XCHK(hipMemcpyAsync(ctx->h_hist, ctx->d_hist, (size_t)(PAPR_TRUE_MAX_LEVELS + 1) * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost, ctx->stream));
    if (peers) {  // exchange 2, in the stream: what the sweep decided, what the recount found, and whether either may be used
        papr_launch_xpack(ctx->stream, ctx->d_sweep_hist, kBinsMax, ctx->d_sweep_hist + kBinsMax, (uint32_t)run.blocks, run.seg_cap,
                          ctx->d_sweep_hist + kBinsMax + 2 * run.blocks, ctx->d_hist, PAPR_TRUE_MAX_LEVELS + 1, ctx->d_guess,
                          ctx->d_true, d_xvec);
        XCHK(hipGetLastError());
        rc = xch_allreduce_u64_dev(x, ctx, d_xvec, d_xvec_sum, kXvecWords);
        if (rc)
            return leave(fail(ctx, rc, "exchange: %s", papr_exchange_last_error(x)));
        XCHK(hipMemcpyAsync(ctx->h_xvec, d_xvec_sum, (size_t)kXvecWords * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    ctx->trace.mark("queued");
    if (peers)
        xch_wait_begin(x, "the step's one wait behind its in-stream collectives");
    run_overlap_work(ctx);  // (exact-sum mode: the program's replay, while the recount runs)
    ctx->trace.mark("overlap_done");
    const hipError_t sync_e = hipStreamSynchronize(ctx->stream);
    if (peers) {
        xch_wait_end(x);
        if (xch_cancelled(x))  // (a peer gave up, or PAPR_XCH_TIMEOUT_S ran out: whatever the stream holds is no result)
            return leave(fail(ctx, PAPR_E_STATE, "exchange: %s", papr_exchange_last_error(x)));
    }
    XCHK(sync_e);
    ctx->trace.mark("synced");
#undef XCHK
    struct LeaveMark {  // (where the host's microseconds behind the wait go: PAPR_HOST_TRACE=1)
        papr_hip_ctx *c;
        ~LeaveMark() { c->trace.mark("fused_leave"); }
    } leave_mark{ctx};
    ctx->program_pending = false;
    if (peers && exact)
        ctx->exact_program_before = ctx->h_peer->before;
    if (ctx->sweep_blocks_last >= 8)
        ctx->xcd_first = (int)(ctx->h_result->pad >> 28);  // (where this stream's workgroup 0 ran: the next step's skew follows it)
    partial_to_stats(*ctx->h_result, ctx->n, out);
    out->flags |= ctx->shard_flags;
    if (peers) {
        // decisions from GLOBAL numbers only: every rank takes the same way from here.  A NaN anywhere, or a guess without
        // a band form: the host path, from the start, on every rank.
        const unsigned long long *flags = ctx->h_xvec + kBinsMax + (PAPR_TRUE_MAX_LEVELS + 1);
        if (ctx->h_peer->nan_ranks != 0 || flags[1] != 0) {
            ctx->sweep_info.reason = PAPR_SWEEP_NO_BANDS;
            ctx->xprog_ready = ctx->exact_program_launched = false;  // (whatever was gathered belongs to no usable step)
            ctx->program_view = nullptr;
            ctx->overlap_work = nullptr;
            return PAPR_OK;  // (*done stays false)
        }
    }
    *done = true;
    // ---- what the device decided ----
    const papr_guess_out &g = *ctx->h_guess;
    info.estimate_samples = ngroups * PAPR_ESTIMATE_TILE_SAMPLES;
    info.band_log2 = (int)g.band_log2;
    const bool usable = !std::isnan(out->sum) && g.ok && g.nbands != 0 && g.nbands <= PAPR_GUESS_MAX_BANDS;
    if (!usable) {
        // NaN in the data (the sweep's integer-max trackers do not apply), or a guess without a band form (the sweep
        // was a plain pass 1): the plain pass of the mode takes over (exact-sum mode: with its per-tile sums)
        info.reason = PAPR_SWEEP_NO_BANDS;
        ctx->exact_program_launched = false;
        ctx->xprog_ready = false;
        ctx->program_view = nullptr;
        ctx->est_groups_valid = false;
        if (exact || std::isnan(out->sum))
            return papr_hip_stats(ctx, out);
        return PAPR_OK;  // (tree-sum mode: the launch's pass-1 record stands)
    }
    run.gkeys.assign(g.gkeys, g.gkeys + g.nbands);
    run.half = 1u << g.band_log2;
    run.bands.P = g.P;
    run.nbins = g.P.nkeys + (run.v2 ? 1 : 2);
    rc = sweep_collect(ctx, run);
    if (rc == PAPR_OK && peers) {
        // the file's numbers beside the shard's: bins above each band, the recount against the (common) speculated table
        const unsigned long long *G = ctx->h_xvec, *flags = G + kBinsMax + (PAPR_TRUE_MAX_LEVELS + 1);
        const size_t m = run.gkeys.size();
        ctx->sweep_even_above_global.assign(m, 0);
        uint64_t above = 0;
        for (size_t j = m; j-- > 0;) {
            above += G[2 * j + 2];
            ctx->sweep_even_above_global[j] = above;
        }
        ctx->recount_global.assign(G + kBinsMax, G + kBinsMax + PAPR_TRUE_MAX_LEVELS + 1);
        ctx->sweep_overflow = flags[0] != 0;  // (on ANY rank: then every rank reads its shard again)
        ctx->peer_global = true;
        partial_to_stats(ctx->h_peer->total, ctx->h_peer->n_total, &peer->total);
        peer->total.flags |= (uint32_t)ctx->h_peer->flags;
        peer->before = ctx->h_peer->before;
        peer->global = true;
        ctx->spec_recount_valid = ctx->h_true->ok != 0 && flags[2] == 0;
        if (exact) {
            ctx->exact_swept = true;  // d_seg_D holds every segment's sum and its pair (speculated, or rebuilt)
            ctx->exact_valid = true;
        }
    } else if (rc == PAPR_OK) {
        ctx->spec_recount_valid = ctx->h_true->ok != 0;  // (whether it is the RIGHT table is for resolve_from_sweep to say)
        if (exact) {
            ctx->exact_swept = true;  // d_seg_D holds every segment's sum and its pair (speculated, or rebuilt)
            ctx->exact_valid = true;
        }
    } else {
        ctx->exact_program_launched = false;
    }
    return rc;
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
    // exact-sum mode: keep the per-group sampled sums on the device — the one-read sweep speculates every tile's
    // running-sum binade from them (papr_exact.hip: papr_exact_spec_*)
    ctx->est_groups_valid = false;
    double *group_sums = nullptr;
    if (ctx->exact) {
        if (ctx->est_groups_cap < ngroups) {
            if (ctx->d_est_groups) HIPCHK(ctx, hipFree(ctx->d_est_groups));
            ctx->d_est_groups = nullptr;
            ctx->est_groups_cap = 0;
            const uint64_t cap = std::max<uint64_t>(ngroups, 4096);
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_est_groups, cap * 5 * sizeof(double)));  // 4 wave sums + 1 prefix per group
            ctx->est_groups_cap = cap;
        }
        group_sums = ctx->d_est_groups;
    }
    if (!ctx->d_est_sq) {
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_est_sq, (size_t)ctx->num_cus * 8 * sizeof(double)));
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_est_sq, (size_t)ctx->num_cus * 8 * sizeof(double), hipHostMallocDefault));
    }
    time_begin_kernel(ctx, 4, ngroups * PAPR_ESTIMATE_TILE_SAMPLES * 8);
    papr_launch_estimate(ctx->stream, blocks, ctx->d_iq, ngroups, (uint32_t)ratio, ctx->d_partials, group_sums, ctx->d_est_sq);
    time_end_kernel(ctx);
    HIPCHK(ctx, hipGetLastError());
    papr_launch_stats_finalize(ctx->stream, nullptr, 0, 0, ctx->d_partials, (uint32_t)blocks, ctx->h_result_dev);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_est_sq, ctx->d_est_sq, (size_t)blocks * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    {
        // relative standard error of the estimated mean from the scatter of its P = 4 x ngroups pieces
        double sq = 0.0;
        for (int b = 0; b < blocks; b++)
            sq += ctx->h_est_sq[b];
        const double S = ctx->h_result->sum, P = 4.0 * (double)ngroups;
        const double var_total = P > 1.0 ? P / (P - 1.0) * std::max(0.0, sq - S * S / P) : 0.0;
        est->peak = S > 0.0 ? (float)(std::sqrt(var_total) / S) : 0.0f;
        if (ratio == 1)
            est->peak = 0.0f;  // everything was read: the "estimate" is the mean itself (up to summation order)
    }
    // the record describes the SHARD: the sampled sum scaled to all of its samples, so that shards of different
    // size (or sampling ratio) merge with the right weights and a shard's record is also the estimate of what it
    // adds to the running sum of the shards behind it
    const uint64_t sampled = ngroups * PAPR_ESTIMATE_TILE_SAMPLES;
    est->sum = ctx->h_result->sum * ((double)ctx->n / (double)sampled);
    est->n = ctx->n;
    ctx->sweep_info.estimate_samples = sampled;
    if (group_sums) {
        ctx->est_ngroups = ngroups;
        ctx->est_ratio = ratio;
        ctx->est_groups_valid = true;
    }
    return PAPR_OK;
}

int papr_hip_set_band(papr_hip_ctx *ctx, int band_log2)
{
    if (!ctx || (band_log2 != 0 && (band_log2 < 8 || band_log2 > 20)))
        return PAPR_E_ARG;
    ctx->band_hint = band_log2;
    return PAPR_OK;
}

int papr_hip_set_exact_hint(papr_hip_ctx *ctx, double before_estimate)
{
    if (!ctx)
        return PAPR_E_ARG;
    if (!(before_estimate >= 0.0) || !std::isfinite(before_estimate))
        return fail(ctx, PAPR_E_ARG, "the estimated sum before the shard must be finite and non-negative");
    ctx->exact_before_hint = before_estimate;
    return PAPR_OK;
}

static int papr_hip_stats_sweep_impl(papr_hip_ctx *ctx, const float *guess_levels, int nlevels, papr_stats *out)
{
    if (!ctx || !out || nlevels < 0 || (nlevels && !guess_levels))
        return PAPR_E_ARG;
    if (!ctx->loaded)
        return fail(ctx, PAPR_E_STATE, "papr_hip_stats_sweep called before a shard was loaded");
    if (ctx->have_file_stats) {  // pass 1 (or a whole one-sweep ingest, which stays valid) ran while the file streamed in
        *out = ctx->file_stats;
        return PAPR_OK;
    }
    papr_hip_sweep_info &info = ctx->sweep_info;
    info.swept = info.resolved = 0;
    info.stash_samples = 0;
    ctx->sweep_valid = false;
    auto plain = [&](int reason) {
        info.reason = reason;
        return papr_hip_stats(ctx, out);
    };
    if (!ctx->resident)
        return plain(PAPR_SWEEP_MODE);
    ctx->exact_swept = false;
    ctx->exact_program_launched = false;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    SweepRun run;
    int reason = PAPR_SWEEP_OK;
    int rc = sweep_prepare(ctx, guess_levels, nlevels, ctx->n, ctx->n, &run, &reason);
    if (rc)
        return rc;
    if (reason != PAPR_SWEEP_OK)
        return plain(reason);
    if (run.exact) {
        // speculate every tile's running-sum binade from the estimate's per-group sums (device side, no round trip)
        rc = ensure_exact_buffers(ctx);
        if (rc)
            return rc;
        const uint64_t ntiles = ctx->n / PAPR_EXACT_TILE_SAMPLES;
        if (ctx->est_groups_valid)
            papr_launch_exact_spec(ctx->stream, ctx->d_est_groups, ctx->est_ngroups, (uint32_t)ctx->est_ratio,
                                   (double)ctx->est_ratio, ctx->exact_before_hint, ctx->d_est_groups + 4 * ctx->est_groups_cap,
                                   ntiles, ctx->d_tile_E_spec);
        else  // PAPR_SWEEP2_FAKE_E: one binade for every tile (kernel timing without an estimate; the redo pass repairs it)
            papr_launch_exact_fill_spec(ctx->stream, ctx->d_tile_E_spec, ntiles, env_int("PAPR_SWEEP2_FAKE_E", 0));
        HIPCHK(ctx, hipGetLastError());
    }
    int nrec = 0;
    rc = sweep_launch(ctx, run, ctx->d_iq, ctx->n, ctx->base, 0, &nrec);
    if (rc)
        return rc;
    // (the sweep's bins and segment counters reach the host through the finalize kernel, as in stats_sweep_fused)
    const size_t fetch_words = (size_t)(run.seg_off ? run.seg_off : run.nbins) + 2 * (size_t)run.blocks + 1;
    const bool by_kernel = env_int("PAPR_FUSED_COPIES", 0) == 0 && fetch_words <= 16384;  // (one workgroup copies them)
    if (!ctx->h_sweep_hist_dev)
        HIPCHK(ctx, hipHostGetDevicePointer((void **)&ctx->h_sweep_hist_dev, ctx->h_sweep_hist, 0));
    if (!by_kernel) {
        rc = sweep_fetch(ctx, run);
        if (rc)
            return rc;
    }
    const uint32_t tail = (uint32_t)(ctx->n % run.tile);
    rc = finish_stats(ctx, (size_t)nrec, ctx->d_iq + 2 * (ctx->n - tail), tail, ctx->base + ctx->n - tail, out,
                      ctx->d_sweep_hist, ctx->h_sweep_hist_dev,
                      by_kernel ? (uint32_t)fetch_words : 0u);  // synchronises
    if (rc)
        return rc;
    if (std::isnan(out->sum))  // NaN in the data: the sweep's integer-max trackers do not apply (papr_sweep.hip)
        return plain(PAPR_SWEEP_NO_BANDS);
    rc = sweep_collect(ctx, run);
    if (rc == PAPR_OK && run.exact) {
        ctx->exact_swept = true;   // d_seg_D holds every segment's sum and its pair for the speculated binade
        ctx->exact_valid = true;
    }
    return rc;
}

int papr_hip_get_wg_finish(papr_hip_ctx *ctx, uint32_t *ticks, int cap)
{
    if (!ctx || cap < 0 || (cap && !ticks))
        return PAPR_E_ARG;
    const uint32_t n = ctx->sweep_blocks_last;
    if (!n || !ctx->d_partials || ctx->partials_cap < n)
        return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<papr_partial> recs(n);
    HIPCHK(ctx, hipMemcpy(recs.data(), ctx->d_partials, (size_t)n * sizeof(papr_partial), hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < n && (int)k < cap; k++)
        ticks[k] = recs[k].pad;
    return (int)n;
}

int papr_hip_get_sweep_info(const papr_hip_ctx *ctx, papr_hip_sweep_info *out)
{
    if (!ctx || !out)
        return PAPR_E_ARG;
    *out = ctx->sweep_info;
    out->xcd_first = ctx->xcd_first;
    return PAPR_OK;
}

int papr_hip_stats_sweep(papr_hip_ctx *ctx, const float *guess_levels, int nlevels, papr_stats *out)
{
    return guarded(ctx, [&] { return papr_hip_stats_sweep_impl(ctx, guess_levels, nlevels, out); });
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
    // (peers, single-wait step: the FILE's bins and recount came through the stream — usable as long as the recount ran
    // against the right table; every input of that decision is the same on every rank)
    const bool global = ctx->peer_global;
    ctx->peer_global = false;
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
    // the recount may already be there: stats_sweep_fused ran it against the table the device expected (papr_true_table_kernel)
    const papr_true_out *sp = ctx->spec_recount_valid ? ctx->h_true : nullptr;
    ctx->spec_recount_valid = false;
    // (PAPR_SPEC_RECOUNT=0: never take it — the tests compare the two ways)
    const bool speculated = sp && sp->ok && env_int("PAPR_SPEC_RECOUNT", 1) && (int)sp->nlevels == nlevels && plan.lut && sp->P.nkeys == m &&
                            sp->P.shift == plan.P.shift && sp->P.cell_lo == plan.P.cell_lo &&
                            memcmp(sp->levels, levels, (size_t)nlevels * sizeof(float)) == 0;
    if (speculated) {
        // (h_hist holds it — the shard's; with peers the file's is taken instead)
        if (global)
            memcpy(ctx->h_hist, ctx->recount_global.data(), (size_t)(m + 1) * sizeof(unsigned long long));
    } else if (ctx->sweep_stash_count) {
        int rc = upload_ccdf_table(ctx, plan);
        if (rc)
            return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_hist, 0, (size_t)(m + 1) * sizeof(unsigned long long), ctx->stream));
        time_begin_kernel(ctx, 4, ctx->sweep_stash_count * 4);
        papr_launch_ccdf_power(ctx->stream, ctx->num_cus, plan.lut, plan.lds_bytes, ctx->d_stash,
                               ctx->d_sweep_hist + ctx->sweep_seg_off, ctx->sweep_seg_cap, ctx->sweep_nsegs, ctx->d_table,
                               plan.P, ctx->d_hist, nullptr);
        time_end_kernel(ctx);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_hist, ctx->d_hist, (size_t)(m + 1) * sizeof(unsigned long long),
                                   hipMemcpyDeviceToHost, ctx->stream));
        run_overlap_work(ctx);  // (exact-sum step: the program's replay, while the recount runs)
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    } else {
        memset(ctx->h_hist, 0, (size_t)(m + 1) * sizeof(unsigned long long));
    }
    counts_from_histogram(ctx, plan, nlevels, stash_above.data());  // stash powers above each level ...
    const bool file_wide = global && speculated;
    (void)papr_sweep_resolve(ctx->sweep_keys.data(), (int)ctx->sweep_keys.size(), band_log2,
                             file_wide ? ctx->sweep_even_above_global.data() : ctx->sweep_even_above.data(), levels, nlevels,
                             stash_above.data(), counts_above);  // ... + everything above its band
    ctx->counts_global = file_wide;  // (the caller then needs no exchange of the counters)
    info.resolved = 1;
    info.reason = PAPR_SWEEP_OK;
    *done = true;
    return PAPR_OK;
}

}  // namespace papr_rt
