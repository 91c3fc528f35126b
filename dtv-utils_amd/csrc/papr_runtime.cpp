// papr_runtime.cpp — core of the host runtime behind the C ABI of include/papr_hip.h: context and shard
// management, launch geometry, launch sequencing for the two passes (papr_hip_stats, papr_hip_ccdf) and the exact
// threshold-table construction for pass 2.  The file ingest engine lives in papr_ingest.cpp, the one-sweep mode in
// papr_sweep_rt.cpp, the bit-exact sequential sum in papr_exact_rt.cpp.
//
// No CPU compute path: every sample is reduced on the GPU; the host only reads file bytes into pinned buffers,
// builds the <= 16 K-entry level tables and folds a handful of scalars.

#include "papr_runtime_internal.h"
#include "papr_uring.h"

using namespace papr_rt;

namespace papr_rt {

char g_open_error[256] = "";

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int fail(papr_hip_ctx *ctx, int code, const char *fmt, ...)
{
    char *dst = ctx ? ctx->err : g_open_error;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 256, fmt, ap);
    va_end(ap);
    return code;
}

int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

void parse_tune_env(papr_hip_tuning *t)
{
    const char *v = getenv("PAPR_HIP_TUNE");
    if (!v)
        return;
    std::string s(v);
    size_t pos = 0;
    while (pos < s.size()) {
        size_t end = s.find(',', pos);
        if (end == std::string::npos)
            end = s.size();
        std::string kv = s.substr(pos, end - pos);
        size_t eq = kv.find('=');
        if (eq != std::string::npos) {
            std::string k = kv.substr(0, eq);
            int val = atoi(kv.c_str() + eq + 1);
            if (k == "blocks") t->stats_blocks = t->ccdf_blocks = val;
            else if (k == "variant") t->stats_variant = t->ccdf_variant = val + 1;
            else if (k == "sblocks") t->stats_blocks = val;
            else if (k == "svariant") t->stats_variant = val + 1;
            else if (k == "cblocks") t->ccdf_blocks = val;
            else if (k == "cvariant") t->ccdf_variant = val + 1;
            else if (k == "map") t->stats_map = t->ccdf_map = val + 1;
            else if (k == "smap") t->stats_map = val + 1;
            else if (k == "cmap") t->ccdf_map = val + 1;
            else if (k == "nt") t->nontemporal = val ? 1 : 2;
            else if (k == "copies") t->hist_copies = val;
            else if (k == "search") t->flags = val ? (t->flags | 1) : (t->flags & ~1);
            else if (k == "wblocks") t->sweep_blocks = val;
            else if (k == "wvariant") t->sweep_variant = val + 1;
            else if (k == "wmap") t->sweep_map = val + 1;
            else if (k == "band") t->sweep_band_log2 = val;
            else if (k == "ratio") t->estimate_ratio = val;
        }
        pos = end + 1;
    }
}

// Host work that only needs the exact-sum program, done while the GPU is still busy with what was queued behind it
void run_overlap_work(papr_hip_ctx *ctx)
{
    if (!ctx->overlap_work)
        return;
    std::function<void()> work;
    work.swap(ctx->overlap_work);  // (once)
    if (!ctx->program_pending || !ctx->ev_program)
        return;
    ctx->program_pending = false;
    if (hipEventSynchronize(ctx->ev_program) != hipSuccess)
        return;
    work();
}

int variant_of(const papr_hip_ctx *ctx, Pass p)
{
    if (p == PASS1 && ctx->exact)
        return 1;  // 256 x 4 pipelined: the geometry papr_launch_stats_tilesums is built for
    if (p == SWEEP) {
        const int want = ctx->tune.sweep_variant - 1;
        int threads, exact = 0;
        uint64_t seg;
        size_t lds;
        const bool is_v2 = want >= 0 && papr_sweep2_geometry(want, &threads, &seg, &lds, &exact) == 0;
        int exact3 = 0;
        const bool is_v3 = want >= 0 && papr_sweep3_geometry(want, &threads, &lds, &exact3) == 0;
        if (ctx->exact)  // exact-sum mode: only the variants that also build the rounding-function pairs
            return (is_v2 && exact) || (is_v3 && exact3) ? want : kSweepExactVariant;
        if ((is_v2 && !exact) || (is_v3 && !exact3))
            return want;
        const int v = papr_sweep_variant(want);
        return v >= 0 ? v : kSweepVariant;
    }
    const int v = (p == PASS1 ? ctx->tune.stats_variant : ctx->tune.ccdf_variant) - 1;
    int b, u;
    if (v >= 0 && papr_variant_geometry(v, &b, &u) == 0)
        return v;
    return p == PASS1 ? kStatsVariant : kCcdfVariant;
}

int blocks_of(const papr_hip_ctx *ctx, Pass p)
{
    const int b = p == PASS1 ? ctx->tune.stats_blocks : p == PASS2 ? ctx->tune.ccdf_blocks : ctx->tune.sweep_blocks;
    if (b > 0)
        return b;
    if (p == SWEEP && PAPR_SWEEP_VARIANT_IS_PERSISTENT(variant_of(ctx, SWEEP)))
        return ctx->num_cus;
    return ctx->num_cus * (p == PASS1 ? kStatsPerCU : p == PASS2 ? kCcdfPerCU : kSweepPerCU);
}

// samples one workgroup consumes per loop iteration under the pass's kernel variant
uint64_t tile_samples(const papr_hip_ctx *ctx, Pass p)
{
    int block = 256, unroll = 8;
    (void)papr_variant_geometry(variant_of(ctx, p), &block, &unroll);
    return 2ull * (uint64_t)block * (uint64_t)unroll;
}

int map_of(const papr_hip_ctx *ctx, Pass p)
{
    const int m = (p == PASS1 ? ctx->tune.stats_map : p == PASS2 ? ctx->tune.ccdf_map : ctx->tune.sweep_map) - 1;
    if (m >= 0 && m <= 2)
        return m;
    if (p == SWEEP)
        return kSweepMap;
    // exact-sum mode: the per-tile sums of neighbouring tiles are then written by workgroups of the same
    // XCD (1.70 ms vs 1.83 ms per 10 GiB with grid-stride; its slow mode costs no more than that)
    if (p == PASS1 && ctx->exact)
        return PAPR_MAP_XCD_SPAN;
    return p == PASS1 ? kStatsMap : kCcdfMap;
}

int pick_blocks(const papr_hip_ctx *ctx, Pass p, uint64_t ntiles)
{
    int blocks = blocks_of(ctx, p);
    if ((uint64_t)blocks > ntiles)
        blocks = (int)std::max<uint64_t>(ntiles, 1);
    if (map_of(ctx, p) == PAPR_MAP_XCD_SPAN && blocks >= 8)
        blocks &= ~7;
    return blocks;
}

int effective_map(const papr_hip_ctx *ctx, Pass p, int blocks)
{
    int map = map_of(ctx, p);
    if (map == PAPR_MAP_XCD_SPAN && (blocks % 8) != 0)
        map = PAPR_MAP_GRID_STRIDE;
    return map;
}

bool use_nt(const papr_hip_ctx *ctx)
{
    return ctx->tune.nontemporal != 2;
}

int ensure_partials(papr_hip_ctx *ctx, size_t count)
{
    if (count <= ctx->partials_cap)
        return PAPR_OK;
    if (ctx->d_partials)
        HIPCHK(ctx, hipFree(ctx->d_partials));
    ctx->d_partials = nullptr;
    ctx->partials_cap = 0;
    size_t cap = std::max<size_t>(count, 4096);
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_partials, cap * sizeof(papr_partial)));
    ctx->partials_cap = cap;
    return PAPR_OK;
}

int ensure_table(papr_hip_ctx *ctx, size_t words)
{
    if (words <= ctx->table_cap_words)
        return PAPR_OK;
    if (ctx->d_table) HIPCHK(ctx, hipFree(ctx->d_table));
    if (ctx->h_table) HIPCHK(ctx, hipHostFree(ctx->h_table));
    ctx->d_table = nullptr;
    ctx->h_table = nullptr;
    ctx->table_cap_words = 0;
    size_t cap = std::max<size_t>(words, 16384);
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_table, cap * sizeof(uint32_t)));
    HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_table, cap * sizeof(uint32_t), hipHostMallocDefault));
    ctx->table_cap_words = cap;
    return PAPR_OK;
}

void release_shard(papr_hip_ctx *ctx)
{
    if (ctx->owns_iq && ctx->d_iq)
        (void)hipFree(ctx->d_iq);
    ctx->d_iq = nullptr;
    ctx->owns_iq = false;
    ctx->cap = ctx->n = ctx->base = 0;
    ctx->loaded = ctx->resident = false;
    ctx->have_file_stats = false;
    ctx->exact_valid = false;
    ctx->peer_epoch += 0x9E3779B97F4A7C15ull;  // (the ranks agree again on the single-wait step: stats_sweep_fused)
    ctx->sweep_valid = false;
    ctx->est_groups_valid = ctx->exact_swept = ctx->exact_program_launched = false;
    ctx->shard_flags = 0;
    ctx->path.clear();
}

int ensure_owned_capacity(papr_hip_ctx *ctx, uint64_t nsamples)
{
    if (ctx->d_iq && ctx->cap >= nsamples)
        return PAPR_OK;
    release_shard(ctx);
    // one extra tile of slack keeps every 16-byte lane load in bounds
    size_t bytes = (size_t)(nsamples + PAPR_TILE_SAMPLES_MAX) * 8;
    hipError_t e = hipMalloc((void **)&ctx->d_iq, bytes);
    if (e != hipSuccess) {
        ctx->d_iq = nullptr;
        return fail(ctx, PAPR_E_NOMEM, "hipMalloc(%zu bytes) for the shard failed: %s", bytes, hipGetErrorString(e));
    }
    ctx->owns_iq = true;
    ctx->cap = nsamples;
    return PAPR_OK;
}

// PAPR_TIME_KINDS: bit mask of the launch kinds that are timed at all (experiment knob; default: all)
static bool kind_timed(int kind)
{
    static const int mask = [] {
        const char *e = getenv("PAPR_TIME_KINDS");
        return e && *e ? atoi(e) : 0xFF;
    }();
    return (mask >> kind) & 1;
}

void time_begin(papr_hip_ctx *ctx, int kind, uint64_t bytes)
{
    ctx->time_skipped = !kind_timed(kind) || (kind >= 4 && !ctx->timing_aux);
    if (ctx->time_skipped)
        return;
    if (!ctx->timing || ctx->timed_used >= (size_t)kMaxTimed)
        return;
    if (ctx->timed_used == ctx->timed.size()) {
        TimedLaunch t{};
        if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess)
            return;
        ctx->timed.push_back(t);
    }
    TimedLaunch &t = ctx->timed[ctx->timed_used];
    t.kind = kind;
    t.bytes = bytes;
    (void)hipEventRecord(t.a, ctx->stream);
}

void time_end(papr_hip_ctx *ctx)
{
    if (ctx->time_skipped)
        return;
    if (!ctx->timing || ctx->timed_used >= ctx->timed.size() || ctx->timed_used >= (size_t)kMaxTimed)
        return;
    (void)hipEventRecord(ctx->timed[ctx->timed_used].b, ctx->stream);
    ctx->timed_used++;
}

// The same bracket around exactly ONE kernel launch (estimate / sweep / recount wrappers of papr_sweep.hip): the two
// events are bound to the dispatch itself, so no marker packet goes into the stream on either side of the kernel and
// the elapsed time is the kernel's own.  PAPR_EXT_TIMING=0 keeps the event records.
static bool ext_timing()
{
    static const bool on = [] {
        const char *e = getenv("PAPR_EXT_TIMING");
        return !(e && e[0] == '0');
    }();
    return on;
}

void time_begin_kernel(papr_hip_ctx *ctx, int kind, uint64_t bytes)
{
    if (!ext_timing()) {
        time_begin(ctx, kind, bytes);
        return;
    }
    ctx->time_skipped = !kind_timed(kind) || (kind >= 4 && !ctx->timing_aux);
    if (ctx->time_skipped)
        return;
    if (!ctx->timing || ctx->timed_used >= (size_t)kMaxTimed)
        return;
    if (ctx->timed_used == ctx->timed.size()) {
        TimedLaunch t{};
        if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess)
            return;
        ctx->timed.push_back(t);
    }
    TimedLaunch &t = ctx->timed[ctx->timed_used];
    t.kind = kind;
    t.bytes = bytes;
    const papr_launch_timer timer{t.a, t.b};
    papr_time_next_launch(&timer);
}

void time_end_kernel(papr_hip_ctx *ctx)
{
    if (!ext_timing()) {
        time_end(ctx);
        return;
    }
    papr_time_next_launch(nullptr);  // (nothing was launched in between: disarm)
    if (ctx->time_skipped)
        return;
    if (!ctx->timing || ctx->timed_used >= ctx->timed.size() || ctx->timed_used >= (size_t)kMaxTimed)
        return;
    ctx->timed_used++;
}

// ---- pass 1 over device-resident samples -------------------------------------
// Launches the streaming kernel over the full tiles of [data, data + n) and
// returns how many partial records it appended at d_partials + slot.
int launch_stats_range(papr_hip_ctx *ctx, const float *data, uint64_t n, uint64_t base_index, size_t slot,
                       int *nrecords)
{
    const uint64_t tile = tile_samples(ctx, PASS1);
    const uint64_t ntiles = n / tile;
    *nrecords = 0;
    if (ntiles == 0)
        return PAPR_OK;
    int blocks = pick_blocks(ctx, PASS1, ntiles);
    const int map = effective_map(ctx, PASS1, blocks);
    if (ctx->exact) {
        int rc = ensure_exact_buffers(ctx);
        if (rc)
            return rc;
    }
    time_begin(ctx, 0, ntiles * tile * 8);
    if (ctx->exact) {
        papr_launch_stats_tilesums(ctx->stream, blocks, data, ntiles, base_index, map, ctx->d_partials + slot,
                                   ctx->d_tile_sums, (base_index - ctx->base) / PAPR_EXACT_TILE_SAMPLES);
    } else {
        papr_launch_stats(ctx->stream, variant_of(ctx, PASS1), blocks, use_nt(ctx), data, ntiles, base_index, map,
                          ctx->d_partials + slot);
    }
    time_end(ctx);
    HIPCHK(ctx, hipGetLastError());
    *nrecords = blocks;
    return PAPR_OK;
}

void partial_to_stats(const papr_partial &r, uint64_t n, papr_stats *out)
{
    papr_stats_init(out);
    out->sum = r.sum;
    out->n = n;
    out->peak = r.val[0];    out->peak_idx = r.idx[0];
    out->re_pos = r.val[1];  out->re_pos_idx = r.idx[1];
    out->re_neg = r.val[2];  out->re_neg_idx = r.idx[2];
    out->im_pos = r.val[3];  out->im_pos_idx = r.idx[3];
    out->im_neg = r.val[4];  out->im_neg_idx = r.idx[4];
}

void apply_nan_key(papr_stats *out, unsigned long long key)
{
    if (key == ~0ull)
        return;
    out->flags |= PAPR_FLAG_NAN;
    out->nan_first_idx = key >> 1;
    out->nan_first_neg = (uint32_t)(key & 1u);
    // papr.c:104 — the running double sum takes the first NaN power (with the
    // sign x86 propagates) and keeps it
    out->sum = out->nan_first_neg ? -(double)NAN : (double)NAN;
}

// LUT / search form and LDS layout for plan->keys (unique, ascending); `vblock` = threads of the workgroup that
// will use it, `extra_lds` = what else that workgroup keeps in LDS
int finish_plan(papr_hip_ctx *ctx, CcdfPlan *plan, int vblock, size_t extra_lds)
{
    const uint32_t m = (uint32_t)plan->keys.size();
    papr_ccdf_params &P = plan->P;
    memset(&P, 0, sizeof(P));
    P.nkeys = m;
    plan->lut = false;
    plan->lds_bytes = 0;
    if (m == 0)
        return PAPR_OK;
    const uint32_t nbins = m + 1;
    const size_t lds_cap = (size_t)papr_ccdf_max_dynamic_lds();
    const int waves = vblock / 64;
    const int want_copies = ctx->tune.hist_copies > 0 ? std::min(ctx->tune.hist_copies, waves) : std::min(waves, 4);

    // LUT: the coarsest cell size that still isolates every key in its own cell
    // keys in the normal-float range; a key of 0x7F800000 (a level of FLT_MAX: only +Inf is above it) shares its cell
    // with NaN patterns whatever the cell size, and a LUT cell compares patterns as numbers: the search form, which
    // screens NaNs, serves such tables
    if (plan->keys.front() >= 0x00800000u && plan->keys.back() < 0x7F800000u && !(ctx->tune.flags & 1)) {
        for (int shift = 23; shift >= 8; shift--) {
            const uint32_t c0 = plan->keys.front() >> shift, c1 = plan->keys.back() >> shift;
            const uint64_t ncells = (uint64_t)c1 - c0 + 1;
            if (ncells * 8 > 40 * 1024)
                break;  // finer cells only get bigger
            bool unique_cells = true;
            for (uint32_t k = 1; k < m && unique_cells; k++)
                unique_cells = (plan->keys[k] >> shift) != (plan->keys[k - 1] >> shift);
            if (!unique_cells)
                continue;
            P.shift = (uint32_t)shift;
            P.cell_lo = c0;
            P.ncells = (uint32_t)ncells;
            // patterns past the table that are still numbers (<= +Inf) lie above every level; when the top cell
            // reaches past +Inf (a level equal to FLT_MAX) there are none
            const uint64_t above = ((uint64_t)c1 + 1) << shift;
            P.above_lo = above <= 0x7F800000u ? (uint32_t)above : 0x7F800001u;
            P.above_count = above <= 0x7F800000u ? 0x7F800001u - P.above_lo : 0u;
            P.table_words = 2 * P.ncells;
            plan->lut = true;
            break;
        }
    }
    if (!plan->lut) {
        P.table_words = m;
        uint32_t step = 1;
        while (step * 2 <= m)
            step *= 2;
        P.search_step = step;
    }
    // histogram copies: one per wave when it is cheap, fewer for huge tables
    int copies = want_copies;
    const size_t soft_cap = (size_t)vblock * 80;  // the workgroup's share of 160 KiB when the CU is full of threads
    while (copies > 1 && (size_t)P.table_words * 4 + (size_t)copies * nbins * 4 + extra_lds > soft_cap)
        copies--;
    P.copies = (uint32_t)copies;
    plan->lds_bytes = (size_t)P.table_words * 4 + (size_t)copies * nbins * 4;
    if (plan->lds_bytes + extra_lds > lds_cap)
        return fail(ctx, PAPR_E_LIMIT, "level table needs %zu bytes of LDS (limit %zu)", plan->lds_bytes + extra_lds,
                    lds_cap);
    return PAPR_OK;
}

int plan_ccdf(papr_hip_ctx *ctx, const float *levels, int nlevels, CcdfPlan *plan)
{
    plan->keys.clear();
    plan->pos.assign(nlevels, -1);
    std::vector<uint32_t> all(nlevels);
    for (int j = 0; j < nlevels; j++) {
        all[j] = level_key(levels[j]);
        if (all[j] != kNever)
            plan->keys.push_back(all[j]);
    }
    std::sort(plan->keys.begin(), plan->keys.end());
    plan->keys.erase(std::unique(plan->keys.begin(), plan->keys.end()), plan->keys.end());
    for (int j = 0; j < nlevels; j++)
        if (all[j] != kNever)
            plan->pos[j] = (int)(std::lower_bound(plan->keys.begin(), plan->keys.end(), all[j]) - plan->keys.begin());
    int vblock = 256, vunroll = 8;
    (void)papr_variant_geometry(variant_of(ctx, PASS2), &vblock, &vunroll);
    return finish_plan(ctx, plan, vblock, 0);
}

int upload_ccdf_table(papr_hip_ctx *ctx, const CcdfPlan &plan)
{
    const papr_ccdf_params &P = plan.P;
    int rc = ensure_table(ctx, P.table_words);
    if (rc)
        return rc;
    if (plan.lut) {
        // lut[cell] = {keys strictly below this cell, key inside this cell or never}
        uint32_t k = 0;
        for (uint32_t c = 0; c < P.ncells; c++) {
            uint32_t in_cell = kNever;
            const uint32_t below = k;
            if (k < P.nkeys && (plan.keys[k] >> P.shift) == P.cell_lo + c)
                in_cell = plan.keys[k++];
            ctx->h_table[2 * c] = below;
            ctx->h_table[2 * c + 1] = in_cell;
        }
    } else {
        memcpy(ctx->h_table, plan.keys.data(), (size_t)P.nkeys * 4);
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_table, ctx->h_table, (size_t)P.table_words * 4, hipMemcpyHostToDevice, ctx->stream));
    return PAPR_OK;
}

int launch_ccdf_range(papr_hip_ctx *ctx, const CcdfPlan &plan, const float *data, uint64_t n)
{
    const uint64_t tile = tile_samples(ctx, PASS2);
    const uint64_t ntiles = n / tile;
    const uint32_t tail = (uint32_t)(n - ntiles * tile);
    int blocks = pick_blocks(ctx, PASS2, ntiles);
    const int map = effective_map(ctx, PASS2, blocks);
    time_begin(ctx, 1, n * 8);
    papr_launch_ccdf(ctx->stream, variant_of(ctx, PASS2), blocks, use_nt(ctx), plan.lut, plan.lds_bytes, data, ntiles, map,
                     data + 2 * ntiles * tile, tail, ctx->d_table, plan.P, ctx->d_hist);
    time_end(ctx);
    HIPCHK(ctx, hipGetLastError());
    return PAPR_OK;
}

// finalize pass 1: tail + merge of `records` partials, NaN bookkeeping
int finish_stats(papr_hip_ctx *ctx, size_t records, const float *tail_ptr, uint32_t tail_samples, uint64_t tail_base,
                 papr_stats *out, const unsigned long long *copy_src, unsigned long long *copy_dst, uint32_t copy_words)
{
    papr_launch_stats_finalize(ctx->stream, tail_ptr, tail_samples, tail_base, ctx->d_partials, (uint32_t)records,
                               ctx->h_result_dev, nullptr, copy_src, copy_dst, copy_words);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    partial_to_stats(*ctx->h_result, ctx->n, out);
    out->flags |= ctx->shard_flags;
    return PAPR_OK;
}

// the double sum of a resident shard came out NaN: find the first NaN power and the sign x86 gives it
int resolve_resident_nan(papr_hip_ctx *ctx, papr_stats *out)
{
    if (!std::isnan(out->sum))
        return PAPR_OK;
    unsigned long long key = ~0ull;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_nan_key, &key, 8, hipMemcpyHostToDevice, ctx->stream));
    papr_launch_first_nan(ctx->stream, 1024, ctx->d_iq, ctx->n, ctx->base, ctx->d_nan_key);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(&key, ctx->d_nan_key, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    apply_nan_key(out, key);
    return PAPR_OK;
}

}  // namespace papr_rt

extern "C" {

int papr_hip_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        fail(nullptr, PAPR_E_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
        return e == hipErrorNoDevice ? 0 : PAPR_E_NO_DEVICE;
    }
    return n;
}

const char *papr_hip_last_error(const papr_hip_ctx *ctx)
{
    return ctx ? ctx->err : g_open_error;
}

int papr_hip_open(papr_hip_ctx **out, int device)
{
    if (!out)
        return PAPR_E_ARG;
    *out = nullptr;
    int n = papr_hip_device_count();
    if (n <= 0)
        return fail(nullptr, PAPR_E_NO_DEVICE, "no HIP device available (libpaprhip has no CPU fallback)");
    if (device < 0 || device >= n)
        return fail(nullptr, PAPR_E_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    papr_hip_ctx *ctx = new (std::nothrow) papr_hip_ctx();
    if (!ctx)
        return PAPR_E_NOMEM;
    ctx->device = device;
    auto bail = [&](int code) {
        snprintf(g_open_error, sizeof(g_open_error), "%s", ctx->err);
        papr_hip_close(ctx);
        return code;
    };
#define OPENCHK(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fail(ctx, PAPR_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_));                  \
            return bail(PAPR_E_HIP);                                                               \
        }                                                                                          \
    } while (0)
    const bool trace = env_int("PAPR_OPEN_TRACE", 0) != 0;  // where the time of opening a context goes (stderr)
    double t_prev = now_s();
    auto lap = [&](const char *what) {
        if (trace) {
            const double t = now_s();
            fprintf(stderr, "papr_hip_open: %-28s %8.3f ms\n", what, (t - t_prev) * 1e3);
            t_prev = t;
        }
    };
    OPENCHK(hipSetDevice(device));
    lap("hipSetDevice");
    hipDeviceProp_t prop;
    OPENCHK(hipGetDeviceProperties(&prop, device));
    lap("hipGetDeviceProperties");
    snprintf(ctx->name, sizeof(ctx->name), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    OPENCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    OPENCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    OPENCHK(hipHostMalloc((void **)&ctx->h_result, sizeof(papr_partial), hipHostMallocMapped));
    OPENCHK(hipHostGetDevicePointer((void **)&ctx->h_result_dev, ctx->h_result, 0));
    OPENCHK(hipMalloc((void **)&ctx->d_hist, (PAPR_HIP_MAX_LEVELS + 1) * sizeof(unsigned long long)));
    OPENCHK(hipHostMalloc((void **)&ctx->h_hist, (PAPR_HIP_MAX_LEVELS + 1) * sizeof(unsigned long long),
                          hipHostMallocDefault));
    OPENCHK(hipMalloc((void **)&ctx->d_nan_key, sizeof(unsigned long long)));
    lap("streams + small buffers");
#undef OPENCHK
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
        free_b = (size_t)64 << 30;
    const int budget_mb = env_int("PAPR_HBM_BUDGET_MB", 0);
    ctx->hbm_budget = budget_mb > 0 ? (size_t)budget_mb << 20 : free_b / 10 * 9;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    parse_tune_env(&ctx->tune);

    lap("hipMemGetInfo");
    if (ensure_partials(ctx, 4096) != PAPR_OK)
        return bail(PAPR_E_HIP);
    papr_kernels_prepare_device();  // function attributes are per device
    lap("kernel attributes (module load)");
    *out = ctx;
    return PAPR_OK;
}

void papr_hip_close(papr_hip_ctx *ctx)
{
    if (!ctx)
        return;
    if (ctx->device >= 0)
        (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->copy_stream2) (void)hipStreamSynchronize(ctx->copy_stream2);
    delete ctx->pool;
    delete ctx->uring;
    release_shard(ctx);
    for (auto &t : ctx->timed) {
        (void)hipEventDestroy(t.a);
        (void)hipEventDestroy(t.b);
    }
    for (int b = 0; b < kMaxBuf; b++) {
        if (ctx->h_stage[b]) (void)hipHostFree(ctx->h_stage[b]);
        if (ctx->d_stage[b]) (void)hipFree(ctx->d_stage[b]);
        if (ctx->ev_copy[b]) (void)hipEventDestroy(ctx->ev_copy[b]);
        if (ctx->ev_kernel[b]) (void)hipEventDestroy(ctx->ev_kernel[b]);
    }
    if (ctx->d_tail) (void)hipFree(ctx->d_tail);
    if (ctx->d_sweep_hist) (void)hipFree(ctx->d_sweep_hist);
    if (ctx->h_sweep_hist) (void)hipHostFree(ctx->h_sweep_hist);
    if (ctx->d_stash) (void)hipFree(ctx->d_stash);
    if (ctx->d_tile_E_spec) (void)hipFree(ctx->d_tile_E_spec);
    if (ctx->d_est_groups) (void)hipFree(ctx->d_est_groups);
    if (ctx->d_est_sq) (void)hipFree(ctx->d_est_sq);
    if (ctx->h_est_sq) (void)hipHostFree(ctx->h_est_sq);
    if (ctx->d_redo) (void)hipFree(ctx->d_redo);
    if (ctx->h_redo_count) (void)hipHostFree(ctx->h_redo_count);
    if (ctx->d_tile_sums) (void)hipFree(ctx->d_tile_sums);
    if (ctx->d_block_sums) (void)hipFree(ctx->d_block_sums);
    if (ctx->d_tile_E) (void)hipFree(ctx->d_tile_E);
    if (ctx->d_seg_D) (void)hipFree(ctx->d_seg_D);
    if (ctx->d_groups) (void)hipFree(ctx->d_groups);
    if (ctx->h_program) (void)hipHostFree(ctx->h_program);
    if (ctx->d_mixed_list) (void)hipFree(ctx->d_mixed_list);
    if (ctx->d_raw_list) (void)hipFree(ctx->d_raw_list);
    if (ctx->d_plan) (void)hipFree(ctx->d_plan);
    if (ctx->d_ambig) (void)hipFree(ctx->d_ambig);
    if (ctx->d_raw_store) (void)hipFree(ctx->d_raw_store);
    if (ctx->d_redo_store) (void)hipFree(ctx->d_redo_store);
    if (ctx->d_pow_tab) (void)hipFree(ctx->d_pow_tab);
    if (ctx->d_true) (void)hipFree(ctx->d_true);
    if (ctx->d_result_copy) (void)hipFree(ctx->d_result_copy);
    if (ctx->h_true) (void)hipHostFree(ctx->h_true);
    if (ctx->d_guess) (void)hipFree(ctx->d_guess);
    if (ctx->h_guess) (void)hipHostFree(ctx->h_guess);
    if (ctx->d_peer) (void)hipFree(ctx->d_peer);
    if (ctx->h_peer) (void)hipHostFree(ctx->h_peer);
    if (ctx->h_xvec) (void)hipHostFree(ctx->h_xvec);
    if (ctx->d_xprog) (void)hipFree(ctx->d_xprog);
    if (ctx->d_xprog_all) (void)hipFree(ctx->d_xprog_all);
    if (ctx->h_xprog_all) (void)hipHostFree(ctx->h_xprog_all);
    if (ctx->ev_program) (void)hipEventDestroy(ctx->ev_program);
    if (ctx->d_partials) (void)hipFree(ctx->d_partials);
    if (ctx->h_result) (void)hipHostFree(ctx->h_result);
    if (ctx->d_hist) (void)hipFree(ctx->d_hist);
    if (ctx->h_hist) (void)hipHostFree(ctx->h_hist);
    if (ctx->d_table) (void)hipFree(ctx->d_table);
    if (ctx->h_table) (void)hipHostFree(ctx->h_table);
    if (ctx->d_nan_key) (void)hipFree(ctx->d_nan_key);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->copy_stream2) (void)hipStreamDestroy(ctx->copy_stream2);
    delete ctx;
}

int papr_hip_device_name(const papr_hip_ctx *ctx, char *buf, int buflen)
{
    if (!ctx || !buf || buflen <= 0)
        return PAPR_E_ARG;
    snprintf(buf, (size_t)buflen, "%s", ctx->name);
    return PAPR_OK;
}

int papr_hip_set_tuning(papr_hip_ctx *ctx, const papr_hip_tuning *t)
{
    if (!ctx || !t)
        return PAPR_E_ARG;
    int vb, vu;
    uint64_t v2seg;
    size_t v2lds;
    if (t->stats_blocks < 0 || t->stats_blocks > 65536 || t->ccdf_blocks < 0 || t->ccdf_blocks > 65536 ||
        t->stats_map < 0 || t->stats_map > 3 || t->ccdf_map < 0 || t->ccdf_map > 3 || t->hist_copies < 0 ||
        (t->stats_variant != 0 && papr_variant_geometry(t->stats_variant - 1, &vb, &vu) != 0) ||
        (t->ccdf_variant != 0 && papr_variant_geometry(t->ccdf_variant - 1, &vb, &vu) != 0) ||
        t->sweep_blocks < 0 || t->sweep_blocks > 65536 || t->sweep_map < 0 || t->sweep_map > 3 ||
        (t->sweep_variant != 0 && papr_sweep_variant(t->sweep_variant - 1) < 0 &&
         papr_sweep2_geometry(t->sweep_variant - 1, &vb, &v2seg, &v2lds, &vu) != 0 &&
         papr_sweep3_geometry(t->sweep_variant - 1, &vb, &v2lds) != 0) ||
        (t->sweep_band_log2 != 0 && (t->sweep_band_log2 < 8 || t->sweep_band_log2 > 20)) || t->estimate_ratio < 0 ||
        t->estimate_ratio > 65536)
        return fail(ctx, PAPR_E_ARG, "bad tuning values");
    ctx->tune = *t;
    ctx->peer_epoch += 0x9E3779B97F4A7C15ull;
    return PAPR_OK;
}

int papr_hip_sweep_variant_built(int variant)
{
    int vb, vu;
    uint64_t seg;
    size_t lds;
    return papr_sweep_variant(variant) >= 0 || papr_sweep2_geometry(variant, &vb, &seg, &lds, &vu) == 0 ||
                   papr_sweep3_geometry(variant, &vb, &lds) == 0
               ? 1
               : 0;
}

int papr_hip_set_timing(papr_hip_ctx *ctx, int enabled)
{
    if (!ctx)
        return PAPR_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->timing = enabled != 0;
    ctx->timing_aux = enabled == 1;
    ctx->timed_used = 0;
    // the event pairs of the first few hundred timed launches exist before the first of them is queued: creating
    // them on demand put the runtime's signal-pool growth inside a benchmark's timed region
    while (enabled && ctx->timed.size() < 256) {
        TimedLaunch t{};
        if (hipEventCreate(&t.a) != hipSuccess)
            break;
        if (hipEventCreate(&t.b) != hipSuccess) {
            (void)hipEventDestroy(t.a);
            break;
        }
        ctx->timed.push_back(t);
    }
    return PAPR_OK;
}

int papr_hip_get_timing(papr_hip_ctx *ctx, papr_hip_timing *out)
{
    if (!ctx || !out)
        return PAPR_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memset(out, 0, sizeof(*out));
    for (size_t k = 0; k < ctx->timed_used; k++) {
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->timed[k].a, ctx->timed[k].b));
        if (ctx->timed[k].kind == 3) {
            out->sweep_ms += ms;
            out->sweep_launches++;
            out->sweep_bytes += ctx->timed[k].bytes;
        } else if (ctx->timed[k].kind == 4) {
            out->aux_ms += ms;
            out->aux_launches++;
            out->aux_bytes += ctx->timed[k].bytes;
        } else if (ctx->timed[k].kind == 2 || ctx->timed[k].kind == 5) {
            out->exact_ms += ms;
            out->exact_launches++;
            out->exact_bytes += ctx->timed[k].bytes;
        } else if (ctx->timed[k].kind == 0) {
            out->stats_ms += ms;
            out->stats_launches++;
            out->stats_bytes += ctx->timed[k].bytes;
        } else {
            out->ccdf_ms += ms;
            out->ccdf_launches++;
            out->ccdf_bytes += ctx->timed[k].bytes;
        }
    }
    return PAPR_OK;
}

int papr_hip_get_timing_launches(papr_hip_ctx *ctx, int kind, float *ms, int cap)
{
    if (!ctx || cap < 0 || (cap && !ms))
        return PAPR_E_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
        return fail(ctx, PAPR_E_HIP, "stream synchronisation failed");
    int n = 0;
    for (size_t k = 0; k < ctx->timed_used; k++) {
        const int cls = ctx->timed[k].kind == 5 ? 2 : ctx->timed[k].kind;
        if (cls != kind)
            continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, ctx->timed[k].a, ctx->timed[k].b) != hipSuccess)
            return fail(ctx, PAPR_E_HIP, "hipEventElapsedTime failed");
        if (n < cap)
            ms[n] = t;
        n++;
    }
    return n;
}

// ---- shard residency ------------------------------------------------------------

int papr_hip_adopt(papr_hip_ctx *ctx, void *device_iq, uint64_t nsamples, uint64_t base_index)
{
    if (!ctx || (!device_iq && nsamples))
        return PAPR_E_ARG;
    if (((uintptr_t)device_iq & 15u) != 0)
        return fail(ctx, PAPR_E_ARG, "adopted device memory must be 16-byte aligned");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    release_shard(ctx);
    ctx->d_iq = (float *)device_iq;
    ctx->owns_iq = false;
    ctx->cap = ctx->n = nsamples;
    ctx->base = base_index;
    ctx->loaded = ctx->resident = true;
    return PAPR_OK;
}

int papr_hip_upload(papr_hip_ctx *ctx, const float *iq, uint64_t nsamples, uint64_t base_index)
{
    if (!ctx || (!iq && nsamples))
        return PAPR_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->owns_iq)
        release_shard(ctx);
    int rc = ensure_owned_capacity(ctx, nsamples);
    if (rc)
        return rc;
    if (nsamples)
        HIPCHK(ctx, hipMemcpy(ctx->d_iq, iq, nsamples * 8, hipMemcpyHostToDevice));
    ctx->n = nsamples;
    ctx->base = base_index;
    ctx->loaded = ctx->resident = true;
    ctx->have_file_stats = false;
    ctx->exact_valid = false;
    ctx->peer_epoch += 0x9E3779B97F4A7C15ull;  // (the ranks agree again on the single-wait step: stats_sweep_fused)
    ctx->sweep_valid = false;
    ctx->est_groups_valid = ctx->exact_swept = ctx->exact_program_launched = false;
    ctx->shard_flags = 0;
    ctx->path.clear();
    return PAPR_OK;
}

int papr_hip_generate(papr_hip_ctx *ctx, const papr_synth_spec *spec, uint64_t first_index, uint64_t nsamples)
{
    if (!ctx || !spec)
        return PAPR_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!(ctx->d_iq && ctx->cap >= nsamples)) {
        int rc = ensure_owned_capacity(ctx, nsamples);
        if (rc)
            return rc;
    }
    if (nsamples) {
        const int blocks = (int)std::min<uint64_t>((nsamples + PAPR_BLOCK - 1) / PAPR_BLOCK, 8192);
        papr_launch_generate(ctx->stream, blocks, ctx->d_iq, nsamples, first_index, *spec);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->n = nsamples;
    ctx->base = first_index;
    ctx->loaded = ctx->resident = true;
    ctx->have_file_stats = false;
    ctx->exact_valid = false;
    ctx->peer_epoch += 0x9E3779B97F4A7C15ull;  // (the ranks agree again on the single-wait step: stats_sweep_fused)
    ctx->sweep_valid = false;
    ctx->est_groups_valid = ctx->exact_swept = ctx->exact_program_launched = false;
    ctx->shard_flags = 0;
    ctx->path.clear();
    return PAPR_OK;
}

int papr_hip_download(papr_hip_ctx *ctx, float *iq, uint64_t first, uint64_t nsamples)
{
    if (!ctx || (!iq && nsamples))
        return PAPR_E_ARG;
    if (!ctx->loaded || !ctx->resident)
        return fail(ctx, PAPR_E_STATE, "no resident shard to download from");
    if (first + nsamples > ctx->n)
        return fail(ctx, PAPR_E_ARG, "download range past the end of the shard");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (nsamples)
        HIPCHK(ctx, hipMemcpy(iq, ctx->d_iq + 2 * first, nsamples * 8, hipMemcpyDeviceToHost));
    return PAPR_OK;
}

// ---- pass 1 ---------------------------------------------------------------------

int papr_hip_stats(papr_hip_ctx *ctx, papr_stats *out)
{
    if (!ctx || !out)
        return PAPR_E_ARG;
    if (!ctx->loaded)
        return fail(ctx, PAPR_E_STATE, "papr_hip_stats called before a shard was loaded");
    if (ctx->have_file_stats) {  // computed while the file streamed in (a sweep made during that load stays valid)
        *out = ctx->file_stats;
        return PAPR_OK;
    }
    ctx->sweep_valid = false;  // a sweep only serves the papr_hip_ccdf calls that directly follow it
    ctx->exact_swept = false;
    ctx->exact_program_launched = false;
    if (!ctx->resident)
        return fail(ctx, PAPR_E_STATE, "the shard is not resident and has no pass-1 result: reload it");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int nrec = 0;
    int rc = ensure_partials(ctx, (size_t)blocks_of(ctx, PASS1) + 1);
    if (rc)
        return rc;
    rc = launch_stats_range(ctx, ctx->d_iq, ctx->n, ctx->base, 0, &nrec);
    if (rc)
        return rc;
    const uint32_t tail = (uint32_t)(ctx->n % tile_samples(ctx, PASS1));
    rc = finish_stats(ctx, (size_t)nrec, ctx->d_iq + 2 * (ctx->n - tail), tail, ctx->base + ctx->n - tail, out);
    if (rc)
        return rc;
    rc = resolve_resident_nan(ctx, out);
    if (rc)
        return rc;
    ctx->exact_valid = ctx->exact;
    return PAPR_OK;
}

}  // extern "C"

namespace papr_rt {

void counts_from_histogram(const papr_hip_ctx *ctx, const CcdfPlan &plan, int nlevels, uint64_t *counts_above)
{
    // samples above unique key i = everything binned at i + 1 or higher
    const uint32_t m = plan.P.nkeys;
    std::vector<uint64_t> above(m);
    uint64_t run = 0;
    for (uint32_t i = m; i-- > 0;) {
        run += ctx->h_hist[i + 1];
        above[i] = run;
    }
    for (int j = 0; j < nlevels; j++)
        counts_above[j] = plan.pos[j] >= 0 ? above[plan.pos[j]] : 0;
}

}  // namespace papr_rt

extern "C" {

// ---- pass 2 ---------------------------------------------------------------------

static int papr_hip_ccdf_impl(papr_hip_ctx *ctx, const float *levels, int nlevels, uint64_t *counts_above)
{
    if (!ctx || nlevels < 0 || (nlevels && (!levels || !counts_above)))
        return PAPR_E_ARG;
    if (!ctx->loaded)
        return fail(ctx, PAPR_E_STATE, "papr_hip_ccdf called before a shard was loaded");
    if (nlevels > PAPR_HIP_MAX_LEVELS)
        return fail(ctx, PAPR_E_LIMIT, "%d levels exceeds PAPR_HIP_MAX_LEVELS (%d)", nlevels, PAPR_HIP_MAX_LEVELS);
    ctx->sweep_info.resolved = 0;
    ctx->counts_global = false;
    if (nlevels == 0)
        return PAPR_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CcdfPlan plan;
    ctx->trace.mark("ccdf_enter");
    int rc = plan_ccdf(ctx, levels, nlevels, &plan);
    if (rc)
        return rc;
    ctx->trace.mark("ccdf_planned");
    const uint32_t m = plan.P.nkeys;
    if (m == 0 || ctx->n == 0) {
        for (int j = 0; j < nlevels; j++)
            counts_above[j] = 0;
        return PAPR_OK;
    }
    if (ctx->sweep_valid) {  // the one-sweep pass already decided everything outside the bands
        bool done = false;
        rc = resolve_from_sweep(ctx, plan, levels, nlevels, counts_above, &done);
        ctx->trace.mark("ccdf_resolved");
        if (rc || done)
            return rc;
    }
    rc = upload_ccdf_table(ctx, plan);
    if (rc)
        return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_hist, 0, (size_t)(m + 1) * sizeof(unsigned long long), ctx->stream));
    if (ctx->resident)
        rc = launch_ccdf_range(ctx, plan, ctx->d_iq, ctx->n);
    else
        rc = stream_file(ctx, PASS_STREAM_CCDF, &plan, nullptr);
    if (rc)
        return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_hist, ctx->d_hist, (size_t)(m + 1) * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost, ctx->stream));
    run_overlap_work(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    counts_from_histogram(ctx, plan, nlevels, counts_above);
    return PAPR_OK;
}

int papr_hip_ccdf(papr_hip_ctx *ctx, const float *levels, int nlevels, uint64_t *counts_above)
{
    return guarded(ctx, [&] { return papr_hip_ccdf_impl(ctx, levels, nlevels, counts_above); });
}

}  // extern "C"
