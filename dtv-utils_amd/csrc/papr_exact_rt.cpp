// papr_exact_rt.cpp — host side of the bit-exact sequential sum (papr_exact.hip): buffers, the classify / segment /
// group / pack launch sequence, the sum program and its host-assembled fallback.

#include "papr_runtime_internal.h"

using namespace papr_rt;

namespace papr_rt {

// How far the prefix sums the tiles are classified with may be from the reference's running sum, relative: the reference's
// own recursive sum is within n * 2^-53 of the real sum of non-negative terms, the prefixes here — tile sums out of pairs
// built for a binade that is at most one too high, i.e. additions rounded to at most twice the true ulp, then a tree — within
// 2 n * 2^-53; 4 n * 2^-53 covers both with a third to spare.  (It was 8 n: a tile counts as undecided within that
// distance of a power of two and travels raw, and the distance grows with n * the sum, i.e. with n squared — a single
// shard of 192 GiB had more undecided tiles around 2^34 and 2^35 than the program's raw list holds, and lost its
// sequential sum to the tree sum.)
static inline double exact_delta(uint64_t n)
{
    return std::max(1.0e-6, 4.0 * (double)n * 1.1102230246251565e-16);
}

// exact-sum mode: device buffers sized for the current shard
int ensure_exact_buffers(papr_hip_ctx *ctx)
{
    const uint64_t ntiles = ctx->n / PAPR_EXACT_TILE_SAMPLES;
    if (ntiles <= ctx->exact_tiles_cap && ctx->d_tile_sums)
        return PAPR_OK;
    if (ctx->d_tile_sums) (void)hipFree(ctx->d_tile_sums);
    if (ctx->d_block_sums) (void)hipFree(ctx->d_block_sums);
    if (ctx->d_tile_E) (void)hipFree(ctx->d_tile_E);
    if (ctx->d_seg_D) (void)hipFree(ctx->d_seg_D);
    if (ctx->d_groups) (void)hipFree(ctx->d_groups);
    if (ctx->d_tile_E_spec) (void)hipFree(ctx->d_tile_E_spec);
    ctx->d_tile_E_spec = nullptr;
    ctx->d_tile_sums = ctx->d_block_sums = ctx->d_seg_D = nullptr;
    ctx->d_tile_E = nullptr;
    ctx->d_groups = nullptr;
    ctx->exact_tiles_cap = 0;
    const uint64_t cap = std::max<uint64_t>(ntiles, 1024);
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_tile_sums, cap * PAPR_EXACT_TILE_WAVES * sizeof(double)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_block_sums, (cap / 1024 + 2) * sizeof(double)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_tile_E, cap * sizeof(int32_t)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_tile_E_spec, cap * sizeof(int32_t)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_seg_D, cap * 2 * 2 * sizeof(double)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_groups, (cap / PAPR_EXACT_GROUP_TILES + 2) * sizeof(papr_exact_group)));
    ctx->exact_tiles_cap = cap;
    return PAPR_OK;
}

// exact-sum mode on a re-streamed shard: the fused sweep (rounding functions + pass 2) over one staged chunk,
// and the unprovable tiles of that chunk kept for the sum program
int launch_fused_chunk(papr_hip_ctx *ctx, const CcdfPlan &plan, const float *chunk, uint64_t s0, uint64_t cnt, bool last)
{
    const uint64_t tile0 = s0 / PAPR_EXACT_TILE_SAMPLES, ntiles = cnt / PAPR_EXACT_TILE_SAMPLES;
    const uint32_t tail = (uint32_t)(cnt - ntiles * PAPR_EXACT_TILE_SAMPLES);  // only the last chunk has one
    const uint64_t nsegs = 2 * ntiles;
    const uint64_t wg_waves = (uint64_t)papr_exact_fused_waves();
    const int per_cu = std::max(1, env_int("PAPR_EXACT_WG_PER_CU", 2));
    const int blocks =
        (int)std::max<uint64_t>(1, std::min<uint64_t>((nsegs + wg_waves - 1) / wg_waves, (uint64_t)ctx->num_cus * per_cu));
    time_begin(ctx, 2, cnt * 8);
    papr_launch_exact_segments_ccdf(ctx->stream, blocks, chunk, nsegs, ctx->d_tile_E + tile0, ctx->d_seg_D + 4 * tile0,
                                    chunk + 2 * ntiles * PAPR_EXACT_TILE_SAMPLES, tail, ctx->d_table, plan.P,
                                    plan.lds_bytes, ctx->d_hist);
    time_end(ctx);
    papr_launch_exact_capture(ctx->stream, chunk, tile0, ntiles, ctx->d_ambig + kCapRaw, ctx->d_ambig + 2 * kCapRaw,
                              kCapRaw, ctx->d_raw_store);
    HIPCHK(ctx, hipGetLastError());
    if (last && tail)
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_tail, chunk + 2 * ntiles * PAPR_EXACT_TILE_SAMPLES, (size_t)tail * 8,
                                   hipMemcpyDeviceToDevice, ctx->stream));
    return PAPR_OK;
}

}  // namespace papr_rt

extern "C" {

// ---- bit-exact mean ---------------------------------------------------------------

int papr_hip_set_exact(papr_hip_ctx *ctx, int enabled)
{
    if (!ctx)
        return PAPR_E_ARG;
    if ((enabled != 0) != ctx->exact) {
        ctx->exact = enabled != 0;
        ctx->exact_valid = false;
        ctx->peer_epoch += 0x9E3779B97F4A7C15ull;  // (the ranks agree again on the single-wait step: stats_sweep_fused)
        ctx->exact_swept = ctx->est_groups_valid = ctx->exact_program_launched = false;
        if (ctx->resident)
            ctx->have_file_stats = false;  // pass 1 is re-run over the resident shard in the other mode
    }
    return PAPR_OK;
}

}  // extern "C"

namespace papr_rt {

int exact_preconditions(papr_hip_ctx *ctx, double before, bool allow_restreamed)
{
    if (!ctx->exact || !ctx->exact_valid || !ctx->loaded)
        return fail(ctx, PAPR_E_STATE, "exact sum needs papr_hip_set_exact(1) and papr_hip_stats on the current shard first");
    if (!ctx->resident && !allow_restreamed)
        return fail(ctx, PAPR_E_STATE, "papr_hip_exact_program needs a shard that is resident in HBM "
                                       "(re-streamed shards: papr_hip_ccdf_exact)");
    if (!(before >= 0.0) || !std::isfinite(before))
        return fail(ctx, PAPR_E_ARG, "`before` must be a finite, non-negative sum");
    return PAPR_OK;
}

int reserve_program(papr_hip_ctx *ctx, size_t want)  // grow the pinned program buffer, keeping its contents
{
    if (want <= ctx->h_program_cap)
        return PAPR_OK;
    unsigned char *fresh = nullptr;
    const size_t cap = std::max<size_t>(want + want / 4, (size_t)1 << 20);
    HIPCHK(ctx, hipHostMalloc((void **)&fresh, cap, hipHostMallocMapped));
    if (ctx->h_program) {
        memcpy(fresh, ctx->h_program, ctx->h_program_cap);
        (void)hipHostFree(ctx->h_program);
    }
    ctx->h_program = fresh;
    ctx->h_program_cap = cap;
    return PAPR_OK;
}

// Device side of the exact sum: classify tiles, one sweep over the samples for the per-segment
// rounding functions (with pass 2 fused in when `fused` is given), pre-compose the groups and gather
// the sum program into mapped host memory — one stream synchronisation in total.  *bytes = 0 means
// the device-side gather overflowed its lists and the caller has to assemble the program itself.
int run_exact_device(papr_hip_ctx *ctx, double before, uint64_t n_total, const CcdfPlan *fused, size_t *bytes)
{
    const uint64_t ntiles = ctx->n / PAPR_EXACT_TILE_SAMPLES;
    const uint64_t ngroups = (ntiles + PAPR_EXACT_GROUP_TILES - 1) / PAPR_EXACT_GROUP_TILES;
    const uint32_t tail = (uint32_t)(ctx->n - ntiles * PAPR_EXACT_TILE_SAMPLES);
    // margin >= the worst-case relative drift of a sequential double sum of n_total non-negative terms
    const double delta = exact_delta(std::max<uint64_t>(n_total, ctx->n));
    *bytes = 0;
    int rc = ensure_exact_buffers(ctx);
    if (rc)
        return rc;
    if (!ctx->d_plan) {
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_mixed_list, kCapMixed * sizeof(uint32_t)));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_raw_list, kCapRaw * sizeof(uint32_t)));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_plan, sizeof(papr_exact_plan)));
    }
    rc = reserve_program(ctx, sizeof(papr_exact_header) + ngroups * sizeof(papr_exact_group_rec) +
                                  (size_t)kCapMixed * sizeof(papr_exact_mixed_rec) +
                                  (size_t)kCapRaw * sizeof(papr_exact_raw_rec) + (size_t)tail * 8);
    if (rc)
        return rc;
    unsigned char *program_dev = nullptr;
    HIPCHK(ctx, hipHostGetDevicePointer((void **)&program_dev, ctx->h_program, 0));
    const uint64_t nsegs = 2 * ntiles;
    if (ctx->resident) {
        // plain sweep: 4-wave workgroups, 2 per CU; fused sweep: 8-wave workgroups, 2 per CU (16 waves share the LDS tables)
        const int per_cu = std::max(1, env_int("PAPR_EXACT_WG_PER_CU", 2));
        const uint64_t wg_waves = fused ? (uint64_t)papr_exact_fused_waves() : 4;
        const int blocks = (int)std::max<uint64_t>(
            1, std::min<uint64_t>((nsegs + wg_waves - 1) / wg_waves, (uint64_t)ctx->num_cus * per_cu));
        time_begin(ctx, 2, ctx->n * 8);
        papr_launch_exact_classify(ctx->stream, ctx->d_tile_sums, ntiles, ctx->d_block_sums, before, delta, ctx->d_tile_E,
                                   nullptr, 0, nullptr, nullptr);
        if (fused)
            papr_launch_exact_segments_ccdf(ctx->stream, blocks, ctx->d_iq, nsegs, ctx->d_tile_E, ctx->d_seg_D,
                                            ctx->d_iq + 2 * ntiles * PAPR_EXACT_TILE_SAMPLES, tail, ctx->d_table,
                                            fused->P, fused->lds_bytes, ctx->d_hist);
        else
            papr_launch_exact_segments(ctx->stream, blocks, ctx->d_iq, nsegs, ctx->d_tile_E, ctx->d_seg_D);
        papr_launch_exact_groups(ctx->stream, ctx->d_tile_E, ntiles, ctx->d_seg_D, ngroups, ctx->d_groups);
        time_end(ctx);
        papr_launch_exact_pack(ctx->stream, ctx->d_groups, ngroups, ctx->d_tile_E, ntiles, ctx->d_seg_D, ctx->d_iq, nullptr,
                               ctx->d_iq + 2 * ntiles * PAPR_EXACT_TILE_SAMPLES, ctx->n, tail, ctx->d_mixed_list,
                               kCapMixed, ctx->d_raw_list, kCapRaw, ctx->d_plan, program_dev);
    } else {
        // re-streamed shard: classify (also listing the unprovable tiles), then the file goes through the
        // staging buffers once more with the fused sweep on every chunk
        if (!fused)
            return fail(ctx, PAPR_E_STATE, "a re-streamed shard builds its sum program in the pass-2 sweep only");
        if (!ctx->d_ambig) {
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_ambig, (2 * kCapRaw + 1) * sizeof(uint32_t)));
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_raw_store, (size_t)kCapRaw * PAPR_EXACT_TILE_SAMPLES * 8));
        }
        HIPCHK(ctx, hipMemsetAsync(ctx->d_ambig + 2 * kCapRaw, 0, sizeof(uint32_t), ctx->stream));
        papr_launch_exact_classify(ctx->stream, ctx->d_tile_sums, ntiles, ctx->d_block_sums, before, delta, ctx->d_tile_E,
                                   ctx->d_ambig, kCapRaw, ctx->d_ambig + 2 * kCapRaw, ctx->d_ambig + kCapRaw);
        HIPCHK(ctx, hipGetLastError());
        rc = stream_file(ctx, PASS_STREAM_CCDF_EXACT, fused, nullptr);
        if (rc)
            return rc;
        papr_launch_exact_groups(ctx->stream, ctx->d_tile_E, ntiles, ctx->d_seg_D, ngroups, ctx->d_groups);
        papr_launch_exact_pack(ctx->stream, ctx->d_groups, ngroups, ctx->d_tile_E, ntiles, ctx->d_seg_D, nullptr,
                               ctx->d_raw_store, ctx->d_tail, ctx->n, tail, ctx->d_mixed_list, kCapMixed, ctx->d_raw_list,
                               kCapRaw, ctx->d_plan, program_dev);
    }
    HIPCHK(ctx, hipGetLastError());
    if (fused)
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_hist, ctx->d_hist, (size_t)(fused->P.nkeys + 1) * sizeof(unsigned long long),
                                   hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    papr_exact_header h;
    memcpy(&h, ctx->h_program, sizeof(h));
    if (h.magic == PAPR_EXACT_MAGIC && h.reserved == 0 && h.ngroups == ngroups && h.nsamples == ctx->n &&
        !env_int("PAPR_EXACT_HOST_ASSEMBLY", 0))  // (the env switch lets the tests exercise the fallback)
        *bytes = sizeof(h) + ngroups * sizeof(papr_exact_group_rec) + (size_t)h.nmixed * sizeof(papr_exact_mixed_rec) +
                 (size_t)h.nraw * sizeof(papr_exact_raw_rec) + (size_t)tail * 8;
    return PAPR_OK;
}

int reserve_exact_lists(papr_hip_ctx *ctx)
{
    const uint64_t ntiles = ctx->n / PAPR_EXACT_TILE_SAMPLES;
    const uint64_t ngroups = (ntiles + PAPR_EXACT_GROUP_TILES - 1) / PAPR_EXACT_GROUP_TILES;
    const uint32_t tail = (uint32_t)(ctx->n - ntiles * PAPR_EXACT_TILE_SAMPLES);
    if (!ctx->d_plan) {
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_mixed_list, kCapMixed * sizeof(uint32_t)));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_raw_list, kCapRaw * sizeof(uint32_t)));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_plan, sizeof(papr_exact_plan)));
    }
    if (!ctx->d_redo) {
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_redo, (kCapRedo + 1) * sizeof(uint32_t)));
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_redo_count, sizeof(uint32_t), hipHostMallocDefault));
        HIPCHK(ctx, hipHostGetDevicePointer((void **)&ctx->h_redo_count_dev, ctx->h_redo_count, 0));
    }
    return reserve_program(ctx, sizeof(papr_exact_header) + ngroups * sizeof(papr_exact_group_rec) +
                                    (size_t)kCapMixed * sizeof(papr_exact_mixed_rec) +
                                    (size_t)kCapRaw * sizeof(papr_exact_raw_rec) + (size_t)tail * 8);
}

// Slot of the in-stream program exchange, to begin with: the group table of the largest shard, room for 4 mixed groups and
// 8 raw tiles (shards inside a file: 1-2 and 2-4; the shard that STARTS the file crosses the binades of the small sums: 13
// and 23 for the 10 GiB bench shard, more when the file is longer) and a whole tail, in units of 64 KiB.  A program that
// outgrows its slot is marked, that step's programs cross on the host (nothing is lost), and the slot of THAT rank grows
// to what it needed (papr_hip_analyze: every rank sees the same headers).
size_t exact_program_slot_bytes(uint64_t nsamples)
{
    const uint64_t ntiles = nsamples / PAPR_EXACT_TILE_SAMPLES;
    const uint64_t ngroups = (ntiles + PAPR_EXACT_GROUP_TILES - 1) / PAPR_EXACT_GROUP_TILES;
    const size_t want = sizeof(papr_exact_header) + ngroups * sizeof(papr_exact_group_rec) + 4 * sizeof(papr_exact_mixed_rec) +
                        8 * sizeof(papr_exact_raw_rec) + (size_t)PAPR_EXACT_TILE_SAMPLES * 8;
    const size_t forced = (size_t)std::max(0, env_int("PAPR_XPROG_SLOT_KB", 0)) * 1024;  // (tests: a slot too small on purpose)
    return forced ? forced : (want + 65535) & ~(size_t)65535;
}

// from here on the program(s) in mapped host memory and the redo count are complete: whoever waits for this event may use
// them while the stream goes on with the recount (run_overlap_work)
int mark_program_ready(papr_hip_ctx *ctx)
{
    if (!ctx->ev_program)
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_program, hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(ctx->ev_program, ctx->stream));
    ctx->program_pending = true;
    return PAPR_OK;
}

// The one-read form: papr_hip_stats_sweep already left every segment's sum and its pair for a SPECULATED binade in
// d_seg_D.  What is left to do needs no sweep over the samples: true prefix sums from the segment sums, the true
// classification, the pairs of the tiles whose speculated binade was wrong rebuilt from the resident samples
// (normally a fraction of a per cent of them), group composition and the program gather.  Everything is queued on
// the stream; the caller synchronises.  *redo_overflow (valid after that synchronisation) != 0: more tiles to redo
// than the list holds — the caller then runs the full rounding-function sweep (run_exact_full_redo).
int run_exact_swept_streamed(papr_hip_ctx *ctx, double before, double delta, uint64_t ntiles, uint64_t ngroups, uint32_t tail,
                             unsigned char *program_dev);

// before_dev / n_total_dev (peers, single-wait step): the same two numbers where only the device knows them yet;
// slot_dev / slot_cap: the program goes into that device buffer (a slot of the in-stream exchange) instead of h_program.
int run_exact_swept(papr_hip_ctx *ctx, double before, uint64_t n_total, const double *before_dev,
                    const unsigned long long *n_total_dev, unsigned char *slot_dev, uint64_t slot_cap)
{
    const uint64_t ntiles = ctx->n / PAPR_EXACT_TILE_SAMPLES;
    const uint64_t ngroups = (ntiles + PAPR_EXACT_GROUP_TILES - 1) / PAPR_EXACT_GROUP_TILES;
    const uint32_t tail = (uint32_t)(ctx->n - ntiles * PAPR_EXACT_TILE_SAMPLES);
    const double delta = exact_delta(std::max<uint64_t>(n_total, ctx->n));
    int rc = reserve_exact_lists(ctx);
    if (rc)
        return rc;
    unsigned char *program_dev = nullptr;
    HIPCHK(ctx, hipHostGetDevicePointer((void **)&program_dev, ctx->h_program, 0));
    // (the redo counter is zeroed by papr_launch_exact_classify_swept)
    if (!ctx->resident)
        return run_exact_swept_streamed(ctx, before, delta, ntiles, ngroups, tail, program_dev);
    if (slot_dev)
        program_dev = slot_dev;
    ctx->program_view = nullptr;
    ctx->xprog_ready = false;
    time_begin(ctx, 5, 0);  // (helper kernels: timed like the estimate / recount kernels, reported with kind 2)
    papr_launch_exact_classify_swept(ctx->stream, ctx->d_seg_D, ntiles, ctx->d_block_sums, before, delta, ctx->d_tile_E,
                                     ctx->d_tile_E_spec, ctx->d_redo, kCapRedo, ctx->d_redo + kCapRedo, nullptr, 0, nullptr,
                                     nullptr, before_dev, n_total_dev, ctx->n);
    papr_launch_exact_redo(ctx->stream, ctx->num_cus, ctx->d_iq, ctx->d_tile_E, ctx->d_seg_D, ctx->d_redo,
                           ctx->d_redo + kCapRedo, kCapRedo, 0);
    // (the group table goes straight into the program — unless the program's destination is a bounded slot it might not fit)
    const bool table_fits = !slot_dev || 48 + ngroups * sizeof(papr_exact_group_rec) <= slot_cap;
    papr_launch_exact_groups(ctx->stream, ctx->d_tile_E, ntiles, ctx->d_seg_D, ngroups, ctx->d_groups,
                             table_fits ? program_dev : nullptr);
    time_end(ctx);
    papr_launch_exact_pack(ctx->stream, ctx->d_groups, ngroups, ctx->d_tile_E, ntiles, ctx->d_seg_D, ctx->d_iq, nullptr,
                           ctx->d_iq + 2 * ntiles * PAPR_EXACT_TILE_SAMPLES, ctx->n, tail, ctx->d_mixed_list, kCapMixed,
                           ctx->d_raw_list, kCapRaw, ctx->d_plan, program_dev, ctx->d_redo + kCapRedo, ctx->h_redo_count_dev,
                           slot_dev ? slot_cap : 0, kCapRedo,
                           papr_exact_prefix_src{(const double *)ctx->d_seg_D, ctx->d_block_sums, before_dev, before, 1,
                                                 table_fits ? 1 : 0});
    HIPCHK(ctx, hipGetLastError());
    if (slot_dev)
        return PAPR_OK;  // (the caller gathers the ranks' slots and records the event behind that)
    return mark_program_ready(ctx);
}

// The same for a shard that STREAMED through the sweep (papr_hip_load_file_sweep in exact-sum mode) and is gone: the
// few tiles whose samples are needed once more — those whose speculated binade was wrong (their pairs are rebuilt) and
// the unprovable ones (they travel raw in the sum program) — are read back from the file, a few MB instead of a
// second pass over it.  More of them than the lists hold: *h_redo_count says so and the caller reads the file again.
int run_exact_swept_streamed(papr_hip_ctx *ctx, double before, double delta, uint64_t ntiles, uint64_t ngroups, uint32_t tail,
                             unsigned char *program_dev)
{
    if (!ctx->d_ambig) {
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_ambig, (2 * kCapRaw + 1) * sizeof(uint32_t)));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_raw_store, (size_t)kCapRaw * PAPR_EXACT_TILE_SAMPLES * 8));
    }
    HIPCHK(ctx, hipMemsetAsync(ctx->d_ambig + 2 * kCapRaw, 0, sizeof(uint32_t), ctx->stream));
    time_begin(ctx, 5, 0);  // (helper kernels: timed like the estimate / recount kernels, reported with kind 2)
    papr_launch_exact_classify_swept(ctx->stream, ctx->d_seg_D, ntiles, ctx->d_block_sums, before, delta, ctx->d_tile_E,
                                     ctx->d_tile_E_spec, ctx->d_redo, kCapRedo, ctx->d_redo + kCapRedo, ctx->d_ambig, kCapRaw,
                                     ctx->d_ambig + 2 * kCapRaw, ctx->d_ambig + kCapRaw);
    time_end(ctx);
    HIPCHK(ctx, hipGetLastError());
    // which tiles?  (one round trip; the lists are a few KB)
    // tiles (of 16 KiB: 512 MiB in all) read back before a second pass over the file is the better deal
    // (PAPR_STREAM_REDO_CAP lowers it: the tests force the fall-back with 0)
    const uint32_t kStreamRedoCap = (uint32_t)std::max(0, std::min(32768, env_int("PAPR_STREAM_REDO_CAP", 32768)));
    std::vector<uint32_t> redo((size_t)kStreamRedoCap + 1), ambig(kCapRaw);
    uint32_t nambig = 0;
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_redo_count, ctx->d_redo + kCapRedo, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&nambig, ctx->d_ambig + 2 * kCapRaw, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t nredo = *ctx->h_redo_count;
    if (nredo > kStreamRedoCap || nambig > kCapRaw) {
        *ctx->h_redo_count = kCapRedo + 1;  // "too many": the caller falls back to the second read
        return PAPR_OK;
    }
    if (nredo)
        HIPCHK(ctx, hipMemcpyAsync(redo.data(), ctx->d_redo, nredo * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (nambig)
        HIPCHK(ctx, hipMemcpyAsync(ambig.data(), ctx->d_ambig + kCapRaw, nambig * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                   ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->redo_store_tiles < nredo) {
        if (ctx->d_redo_store) HIPCHK(ctx, hipFree(ctx->d_redo_store));
        ctx->d_redo_store = nullptr;
        ctx->redo_store_tiles = 0;
        const size_t want = std::max<size_t>(nredo, 256);
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_redo_store, want * PAPR_EXACT_TILE_SAMPLES * 8));
        ctx->redo_store_tiles = want;
    }
    // read the tiles (through read_samples: the phantom sample of an odd-float file is patched there)
    FileSrc fs;
    int rc = open_file_src(ctx, ctx->path.c_str(), &fs);
    if (rc)
        return rc;
    constexpr size_t kTileBytes = (size_t)PAPR_EXACT_TILE_SAMPLES * 8;
    std::vector<unsigned char> host;
    try {
        host.resize(((size_t)nredo + nambig) * kTileBytes);
    } catch (...) {
        close_file_src(&fs);
        return fail(ctx, PAPR_E_NOMEM, "out of host memory");
    }
    for (uint32_t k = 0; k < nredo + nambig && rc == PAPR_OK; k++) {
        const uint64_t tile = k < nredo ? redo[k] : ambig[k - nredo];
        if (read_samples(fs, ctx->file_first + tile * PAPR_EXACT_TILE_SAMPLES, PAPR_EXACT_TILE_SAMPLES,
                         host.data() + (size_t)k * kTileBytes))
            rc = fail(ctx, PAPR_E_IO, "read error in %s", ctx->path.c_str());
    }
    close_file_src(&fs);
    if (rc)
        return rc;
    if (nredo)
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_redo_store, host.data(), (size_t)nredo * kTileBytes, hipMemcpyHostToDevice, ctx->stream));
    if (nambig)
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_raw_store, host.data() + (size_t)nredo * kTileBytes, (size_t)nambig * kTileBytes,
                                   hipMemcpyHostToDevice, ctx->stream));
    time_begin(ctx, 5, 0);  // (helper kernels: timed like the estimate / recount kernels, reported with kind 2)
    if (nredo)
        papr_launch_exact_redo(ctx->stream, (int)std::min<uint32_t>((uint32_t)ctx->num_cus, (nredo + 1) / 2), ctx->d_redo_store,
                               ctx->d_tile_E, ctx->d_seg_D, ctx->d_redo, ctx->d_redo + kCapRedo, kCapRedo, 1);
    papr_launch_exact_groups(ctx->stream, ctx->d_tile_E, ntiles, ctx->d_seg_D, ngroups, ctx->d_groups);
    time_end(ctx);
    papr_launch_exact_pack(ctx->stream, ctx->d_groups, ngroups, ctx->d_tile_E, ntiles, ctx->d_seg_D, nullptr, ctx->d_raw_store,
                           ctx->d_tail, ctx->n, tail, ctx->d_mixed_list, kCapMixed, ctx->d_raw_list, kCapRaw, ctx->d_plan,
                           program_dev);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (`host` must outlive the copies)
    return PAPR_OK;
}

// speculation missed on more tiles than the redo list holds: every tile's pairs from its true binade (a second read)
int run_exact_full_redo(papr_hip_ctx *ctx)
{
    const uint64_t ntiles = ctx->n / PAPR_EXACT_TILE_SAMPLES;
    const uint64_t ngroups = (ntiles + PAPR_EXACT_GROUP_TILES - 1) / PAPR_EXACT_GROUP_TILES;
    const uint32_t tail = (uint32_t)(ctx->n - ntiles * PAPR_EXACT_TILE_SAMPLES);
    unsigned char *program_dev = nullptr;
    HIPCHK(ctx, hipHostGetDevicePointer((void **)&program_dev, ctx->h_program, 0));
    ctx->program_view = nullptr;  // (the program is rebuilt into h_program)
    const uint64_t nsegs = 2 * ntiles;
    const int blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((nsegs + 3) / 4, (uint64_t)ctx->num_cus * 2));
    time_begin(ctx, 2, ctx->n * 8);
    papr_launch_exact_segments(ctx->stream, blocks, ctx->d_iq, nsegs, ctx->d_tile_E, ctx->d_seg_D);
    papr_launch_exact_groups(ctx->stream, ctx->d_tile_E, ntiles, ctx->d_seg_D, ngroups, ctx->d_groups);
    time_end(ctx);
    papr_launch_exact_pack(ctx->stream, ctx->d_groups, ngroups, ctx->d_tile_E, ntiles, ctx->d_seg_D, ctx->d_iq, nullptr,
                           ctx->d_iq + 2 * ntiles * PAPR_EXACT_TILE_SAMPLES, ctx->n, tail, ctx->d_mixed_list, kCapMixed,
                           ctx->d_raw_list, kCapRaw, ctx->d_plan, program_dev);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PAPR_OK;
}

// the program the device-side gather left in h_program: its size, or 0 if the lists overflowed
size_t swept_program_bytes(papr_hip_ctx *ctx)
{
    const uint64_t ntiles = ctx->n / PAPR_EXACT_TILE_SAMPLES;
    const uint64_t ngroups = (ntiles + PAPR_EXACT_GROUP_TILES - 1) / PAPR_EXACT_GROUP_TILES;
    const uint32_t tail = (uint32_t)(ctx->n - ntiles * PAPR_EXACT_TILE_SAMPLES);
    papr_exact_header h;
    memcpy(&h, current_program(ctx), sizeof(h));
    if (h.magic == PAPR_EXACT_MAGIC && h.reserved == 0 && h.ngroups == ngroups && h.nsamples == ctx->n &&
        !env_int("PAPR_EXACT_HOST_ASSEMBLY", 0))
        return sizeof(h) + ngroups * sizeof(papr_exact_group_rec) + (size_t)h.nmixed * sizeof(papr_exact_mixed_rec) +
               (size_t)h.nraw * sizeof(papr_exact_raw_rec) + (size_t)tail * 8;
    return 0;
}

// where the current step's program was gathered: h_program, or this rank's slot of the in-stream exchange
const unsigned char *current_program(const papr_hip_ctx *ctx)
{
    return ctx->program_view ? ctx->program_view : ctx->h_program;
}

// Host-driven assembly, for the (never yet seen) case that a shard has more mixed groups / raw tiles
// than the device-side lists hold.
int assemble_program_on_host(papr_hip_ctx *ctx, const void **program, size_t *bytes);

int assemble_program_on_host(papr_hip_ctx *ctx, const void **program, size_t *bytes)
{
    const uint64_t ntiles = ctx->n / PAPR_EXACT_TILE_SAMPLES;
    const uint64_t ngroups = (ntiles + PAPR_EXACT_GROUP_TILES - 1) / PAPR_EXACT_GROUP_TILES;
    const uint32_t tail = (uint32_t)(ctx->n - ntiles * PAPR_EXACT_TILE_SAMPLES);
    std::vector<papr_exact_group> groups(ngroups);
    if (ngroups) {
        HIPCHK(ctx, hipMemcpyAsync(groups.data(), ctx->d_groups, ngroups * sizeof(papr_exact_group),
                                   hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    // what has to travel in detail: mixed groups (per-tile classes + per-segment pairs) ...
    std::vector<uint64_t> mixed;
    for (uint64_t g = 0; g < ngroups; g++)
        if (groups[g].E == PAPR_EXACT_AMBIG)
            mixed.push_back(g);
    auto reserve = [&](size_t want) -> int { return reserve_program(ctx, want); };
    static_assert(sizeof(papr_exact_group) == sizeof(papr_exact_group_rec), "group record layout");
    const size_t off_groups = sizeof(papr_exact_header);
    const size_t off_mixed = off_groups + ngroups * sizeof(papr_exact_group_rec);
    const size_t off_raw = off_mixed + mixed.size() * sizeof(papr_exact_mixed_rec);
    int rc = reserve(off_raw + (size_t)tail * 8);
    if (rc)
        return rc;
    if (ngroups)
        memcpy(ctx->h_program + off_groups, groups.data(), ngroups * sizeof(papr_exact_group_rec));
    for (size_t k = 0; k < mixed.size(); k++) {
        const uint64_t g = mixed[k];
        papr_exact_mixed_rec *m = (papr_exact_mixed_rec *)(ctx->h_program + off_mixed + k * sizeof(papr_exact_mixed_rec));
        const uint64_t t0 = g * PAPR_EXACT_GROUP_TILES, t1 = std::min<uint64_t>(t0 + PAPR_EXACT_GROUP_TILES, ntiles);
        m->group = g;
        for (uint64_t j = t1 - t0; j < PAPR_EXACT_GROUP_TILES; j++)
            m->tile_E[j] = PAPR_EXACT_ZERO;
        memset(m->seg_D, 0, sizeof(m->seg_D));
        HIPCHK(ctx, hipMemcpyAsync(m->tile_E, ctx->d_tile_E + t0, (t1 - t0) * sizeof(int32_t), hipMemcpyDeviceToHost,
                                   ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(m->seg_D, ctx->d_seg_D + 4 * t0, (t1 - t0) * 2 * 2 * sizeof(double),
                                   hipMemcpyDeviceToHost, ctx->stream));
    }
    if (!mixed.empty())
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // ... and their tiles that are not provably inside one binade (raw samples)
    std::vector<uint64_t> raw;
    for (size_t k = 0; k < mixed.size(); k++) {
        const papr_exact_mixed_rec *m =
            (const papr_exact_mixed_rec *)(ctx->h_program + off_mixed + k * sizeof(papr_exact_mixed_rec));
        for (uint64_t j = 0; j < PAPR_EXACT_GROUP_TILES; j++)
            if (m->tile_E[j] == PAPR_EXACT_AMBIG)
                raw.push_back(mixed[k] * PAPR_EXACT_GROUP_TILES + j);
    }
    const size_t off_tail = off_raw + raw.size() * sizeof(papr_exact_raw_rec);
    const size_t total = off_tail + (size_t)tail * 8;
    rc = reserve(total);
    if (rc)
        return rc;
    for (size_t k = 0; k < raw.size(); k++) {
        papr_exact_raw_rec *r = (papr_exact_raw_rec *)(ctx->h_program + off_raw + k * sizeof(papr_exact_raw_rec));
        r->tile = raw[k];
        for (int q = 0; q < PAPR_XF_TILE_RUNS; q++)  // (no pairs for the runs: the host adds the whole tile sample by sample)
            r->run_E[q] = PAPR_XF_AMBIG;
        memset(r->run_D, 0, sizeof(r->run_D));
    }
    // (the samples of those tiles, then their powers as the device forms them: separate roundings, no FMA — this file is
    // compiled with -ffp-contract=off)
    std::vector<float> iq;
    try {
        iq.resize(raw.size() * 2 * (size_t)PAPR_EXACT_TILE_SAMPLES);
    } catch (...) {
        return fail(ctx, PAPR_E_NOMEM, "out of host memory");
    }
    for (size_t k = 0; k < raw.size(); k++)
        HIPCHK(ctx, hipMemcpyAsync(iq.data() + k * 2 * (size_t)PAPR_EXACT_TILE_SAMPLES, ctx->d_iq + 2 * raw[k] * PAPR_EXACT_TILE_SAMPLES,
                                   PAPR_EXACT_TILE_SAMPLES * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (tail)
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_program + off_tail, ctx->d_iq + 2 * ntiles * PAPR_EXACT_TILE_SAMPLES,
                                   (size_t)tail * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (!raw.empty() || tail)
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t k = 0; k < raw.size(); k++) {
        papr_exact_raw_rec *r = (papr_exact_raw_rec *)(ctx->h_program + off_raw + k * sizeof(papr_exact_raw_rec));
        const float *q = iq.data() + k * 2 * (size_t)PAPR_EXACT_TILE_SAMPLES;
        for (int j = 0; j < PAPR_EXACT_TILE_SAMPLES; j++) {
            const volatile float re2 = q[2 * j] * q[2 * j], im2 = q[2 * j + 1] * q[2 * j + 1];
            r->pw[j] = re2 + im2;
        }
    }
    papr_exact_header h;
    memset(&h, 0, sizeof(h));
    h.magic = PAPR_EXACT_MAGIC;
    h.version = PAPR_EXACT_VERSION;
    h.nsamples = ctx->n;
    h.ntiles = ntiles;
    h.ngroups = ngroups;
    h.tail_samples = tail;
    h.nmixed = (uint32_t)mixed.size();
    h.nraw = (uint32_t)raw.size();
    memcpy(ctx->h_program, &h, sizeof(h));
    *program = ctx->h_program;
    *bytes = total;
    return PAPR_OK;
}

}  // namespace papr_rt

extern "C" {

static int papr_hip_exact_program_impl(papr_hip_ctx *ctx, double before, uint64_t n_total, const void **program, size_t *bytes)
{
    if (!ctx || !program || !bytes)
        return PAPR_E_ARG;
    int rc = exact_preconditions(ctx, before, ctx->exact_swept);
    if (rc)
        return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->exact_swept) {
        const bool launched = ctx->exact_program_launched && before == ctx->exact_program_before;  // (stats_sweep_fused did it)
        ctx->exact_program_launched = false;
        rc = launched ? PAPR_OK : run_exact_swept(ctx, before, n_total);
        if (rc)
            return rc;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->sweep_info.exact_redo_tiles = *ctx->h_redo_count;
        if (*ctx->h_redo_count > kCapRedo) {
            if (!ctx->resident)
                return fail(ctx, PAPR_E_LIMIT, "the streamed shard's binade speculation missed on too many tiles: use "
                                               "papr_hip_ccdf_exact (it reads the file once more)");
            rc = run_exact_full_redo(ctx);
            if (rc)
                return rc;
        }
        *bytes = swept_program_bytes(ctx);
        *program = current_program(ctx);
        if (!*bytes)
            ctx->program_view = nullptr;  // (assembled into h_program)
        return *bytes ? PAPR_OK : assemble_program_on_host(ctx, program, bytes);
    }
    rc = run_exact_device(ctx, before, n_total, nullptr, bytes);
    if (rc)
        return rc;
    *program = ctx->h_program;
    return *bytes ? PAPR_OK : assemble_program_on_host(ctx, program, bytes);
}

static int papr_hip_ccdf_exact_impl(papr_hip_ctx *ctx, const float *levels, int nlevels, uint64_t *counts_above, double before,
                        uint64_t n_total, const void **program, size_t *bytes)
{
    if (!ctx || !program || !bytes || nlevels < 0 || (nlevels && (!levels || !counts_above)))
        return PAPR_E_ARG;
    if (nlevels > PAPR_HIP_MAX_LEVELS)
        return fail(ctx, PAPR_E_LIMIT, "%d levels exceeds PAPR_HIP_MAX_LEVELS (%d)", nlevels, PAPR_HIP_MAX_LEVELS);
    int rc = exact_preconditions(ctx, before, true);
    if (rc)
        return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    bool swept_form = ctx->exact_swept;
    if (swept_form) {
        // one-read form: the sweep already holds what both results need (papr_hip_stats_sweep, or a one-sweep ingest,
        // in exact-sum mode)
        const bool launched = ctx->exact_program_launched && before == ctx->exact_program_before;  // (stats_sweep_fused did it)
        ctx->exact_program_launched = false;
        rc = launched ? PAPR_OK : run_exact_swept(ctx, before, n_total);
        if (rc)
            return rc;
        if (!ctx->resident && *ctx->h_redo_count > kCapRedo) {
            // a streamed shard whose speculation missed on more tiles than are worth reading back: the file goes
            // through the fused sweep once more (below), classified from the segment sums the first pass left
            ctx->sweep_info.exact_redo_tiles = *ctx->h_redo_count;
            rc = ensure_exact_buffers(ctx);
            if (rc)
                return rc;
            papr_launch_exact_segsums_to_tilesums(ctx->stream, ctx->d_seg_D, ctx->n / PAPR_EXACT_TILE_SAMPLES, ctx->d_tile_sums);
            HIPCHK(ctx, hipGetLastError());
            ctx->exact_swept = false;
            ctx->exact_program_launched = false;
            swept_form = false;
        }
    }
    if (swept_form) {
        rc = papr_hip_ccdf(ctx, levels, nlevels, counts_above);  // the stash recount (or, speculation missed, pass 2)
        if (rc)
            return rc;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->program_pending = false;
        ctx->sweep_info.exact_redo_tiles = *ctx->h_redo_count;
        if (*ctx->h_redo_count > kCapRedo) {
            rc = run_exact_full_redo(ctx);
            if (rc)
                return rc;
        }
        *bytes = swept_program_bytes(ctx);
        *program = current_program(ctx);
        if (!*bytes)
            ctx->program_view = nullptr;  // (assembled into h_program)
        return *bytes ? PAPR_OK : assemble_program_on_host(ctx, program, bytes);
    }
    CcdfPlan plan;
    if (nlevels) {
        rc = plan_ccdf(ctx, levels, nlevels, &plan);
        if (rc)
            return rc;
    }
    // the fused sweep holds the LUT form of the table plus the 36 KiB transpose buffer in LDS
    size_t fused_lds = 0;
    if (nlevels && plan.lut && plan.P.nkeys) {
        plan.P.copies = std::min<uint32_t>(plan.P.copies, 4);
        fused_lds = (size_t)plan.P.table_words * 4 + (size_t)plan.P.copies * (plan.P.nkeys + 1) * 4;
        while (plan.P.copies > 1 && fused_lds > 12 * 1024) {  // two workgroups per CU: 2 x (64 KiB + this) <= 160 KiB
            plan.P.copies--;
            fused_lds = (size_t)plan.P.table_words * 4 + (size_t)plan.P.copies * (plan.P.nkeys + 1) * 4;
        }
        plan.lds_bytes = fused_lds;
    }
    const bool fuse = fused_lds != 0 && fused_lds + papr_exact_transpose_lds_bytes() <= 150 * 1024 && ctx->n > 0;
    if (!fuse) {  // unusual level table: the two sweeps run one after the other
        rc = papr_hip_ccdf(ctx, levels, nlevels, counts_above);
        if (rc)
            return rc;
        return papr_hip_exact_program(ctx, before, n_total, program, bytes);
    }
    rc = upload_ccdf_table(ctx, plan);
    if (rc)
        return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_hist, 0, (size_t)(plan.P.nkeys + 1) * sizeof(unsigned long long), ctx->stream));
    rc = run_exact_device(ctx, before, n_total, &plan, bytes);
    if (rc)
        return rc;
    counts_from_histogram(ctx, plan, nlevels, counts_above);
    *program = ctx->h_program;
    if (*bytes)
        return PAPR_OK;
    if (!ctx->resident)  // the raw tiles of a re-streamed shard are gone once their chunk has left the device
        return fail(ctx, PAPR_E_LIMIT, "too many binade crossings for the device-side program lists");
    return assemble_program_on_host(ctx, program, bytes);
}

int papr_hip_exact_program(papr_hip_ctx *ctx, double before, uint64_t n_total, const void **program, size_t *bytes)
{
    return guarded(ctx, [&] { return papr_hip_exact_program_impl(ctx, before, n_total, program, bytes); });
}

int papr_hip_ccdf_exact(papr_hip_ctx *ctx, const float *levels, int nlevels, uint64_t *counts_above, double before,
                        uint64_t n_total, const void **program, size_t *bytes)
{
    return guarded(ctx, [&] { return papr_hip_ccdf_exact_impl(ctx, levels, nlevels, counts_above, before, n_total, program, bytes); });
}

}  // extern "C"
