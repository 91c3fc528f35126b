// papr_analyze.cpp — papr_hip_analyze: the whole papr result for one shard among its peers in one call (the sequence
// of include/papr_hip.h's primitives that bench.py times and bin/papr prints from), so that a step costs the host
// one entry into the library instead of a dozen; and the in-process exchange transport of bin/papr's shard threads.

#include "papr_runtime_internal.h"

using namespace papr_rt;

extern "C" int papr_exchange_selftest_once(papr_exchange *x, papr_hip_ctx *ctx);  // papr_exchange.cpp

extern "C" {

static int papr_hip_analyze_impl(papr_hip_ctx *ctx, papr_exchange *x, int graph, unsigned flags, papr_result *res, float *levels,
                     uint64_t *counts_above, int cap)
{
    if (!ctx || !res || cap < 0 || (cap && (!levels || !counts_above)))
        return PAPR_E_ARG;
    if (!ctx->loaded)
        return fail(ctx, PAPR_E_STATE, "papr_hip_analyze called before a shard was loaded");
    int rc0 = PAPR_OK;
    memset(res, 0, sizeof(*res));
    ctx->trace.mark("enter");
    if (x && (rc0 = papr_exchange_selftest_once(x, ctx)) != PAPR_OK) {  // PAPR_XCH_SELFTEST=1: once per handle
        snprintf(ctx->err, sizeof(ctx->err), "%s", papr_exchange_last_error(x));
        return rc0;
    }
    auto xfail = [&](int rc) {  // an exchange failed: its text is the detail
        if (x)
            snprintf(ctx->err, sizeof(ctx->err), "exchange: %s", papr_exchange_last_error(x));
        return rc;
    };
    // ---- pass 1 (+ banded pass 2): one sweep, steered by a 1/64-sample estimate of the whole file's mean ----
    papr_stats local;
    int rc;
    bool swept_path = false;
    // exact-sum mode: the shards' programs are exchanged (if there are peers) and replayed as soon as this shard's is
    // complete — inside whichever call is about to wait for the stream while the GPU still has work queued behind the
    // program (papr_hip_ctx::overlap_work): the single-wait step's final wait, or papr_hip_ccdf_exact's
    double seq = 0.0;
    int crc = PAPR_OK;
    bool replayed = false;
    auto replay = [&](const void *prog, size_t nbytes, bool have) {  // exchange (if any) + the chained replay
        if (x) {
            // a shard that could not build its program sends an empty one: the chain then fails on EVERY rank alike
            static const unsigned char none[8] = {0};
            crc = papr_exchange_exact_sum(x, have ? prog : none, have ? nbytes : 0, &seq);
        } else if (have) {
            const void *progs[1] = {prog};
            const size_t sizes[1] = {nbytes};
            crc = papr_exact_chain(progs, sizes, 1, &seq);
        }
    };
    auto replay_when_ready = [&] {
        if (ctx->xprog_ready) {
            // peers, single-wait step: every rank's program crossed the exchange in the stream and lies in its slot of
            // h_xprog_all (papr_sweep_rt.cpp) — replayed here in rank (= file) order, on every rank alike.  A slot that is
            // marked (a program that outgrew it, or is not final) or is no program: nothing is replayed, and every rank —
            // they all see the same slots — exchanges the programs on the host further down.
            ctx->xprog_ready = false;
            const int world = ctx->xprog_world;
            std::vector<const void *> progs((size_t)world);
            std::vector<size_t> sizes((size_t)world);
            bool usable = !env_int("PAPR_EXACT_HOST_ASSEMBLY", 0);
            for (int r = 0; r < world; r++) {
                const unsigned char *slot = ctx->h_xprog_all + ctx->xprog_offs[(size_t)r];
                const size_t room = ctx->xprog_sizes[(size_t)r];
                papr_exact_header h;
                memcpy(&h, slot, sizeof(h));
                if (h.magic != PAPR_EXACT_MAGIC) {
                    usable = false;
                    continue;
                }
                const size_t want = sizeof(h) + (size_t)h.ngroups * sizeof(papr_exact_group_rec) +
                                    (size_t)h.nmixed * sizeof(papr_exact_mixed_rec) + (size_t)h.nraw * sizeof(papr_exact_raw_rec) +
                                    (size_t)h.tail_samples * 8;
                // a program that outgrew its slot: the slot grows to what it needed and a quarter (from the next step on;
                // every rank reads the same header and does the same)
                if (want > room && !env_int("PAPR_XPROG_SLOT_KB", 0))
                    ctx->xprog_sizes[(size_t)r] = std::min<size_t>((want + want / 4 + 65535) & ~(size_t)65535, (size_t)64 << 20);
                if (h.reserved != 0 || want > room)
                    usable = false;
                progs[(size_t)r] = slot;
                sizes[(size_t)r] = want;
            }
            if (!usable)
                return;
            crc = papr_exact_chain(progs.data(), sizes.data(), world, &seq);
            replayed = true;
            return;
        }
        if (*ctx->h_redo_count > kCapRedo)
            return;  // too many tiles to rebuild: the program is not final yet
        const size_t nbytes = swept_program_bytes(ctx);
        if (!nbytes)
            return;  // the device-side gather overflowed its lists: assembled by the host afterwards
        ctx->trace.mark("program_here");
        replay(ctx->h_program, nbytes, true);
        ctx->trace.mark("replayed");
        replayed = true;
    };
    bool fused = false;
    PeerStep peer;
    if (!(flags & PAPR_ANALYZE_TWO_PASS)) {
        const bool alone = papr_exchange_is_identity(x);
        if (ctx->exact)  // (with peers: only the single-wait step over an in-stream exchange leaves work for it)
            ctx->overlap_work = replay_when_ready;
        // estimate, guess (on the device) and sweep in one sequence of launches, one wait (papr_sweep_rt.cpp) — alone, or
        // with the exchanges as collectives on the stream when the transport has them (RCCL)
        rc = stats_sweep_fused(ctx, alone ? nullptr : x, graph, graph ? 48.0 : 60.0,
                               (flags & PAPR_ANALYZE_SPOIL_GUESS) ? 1.03f : 1.0f, &local, &fused, &peer);
        ctx->overlap_work = nullptr;
        if (rc)
            return rc;
        swept_path = fused;
    }
    if (fused) {
        // (done)
    } else if (!(flags & PAPR_ANALYZE_TWO_PASS)) {
        papr_stats est, est_total;
        double est_before = 0.0;
        rc = papr_hip_estimate(ctx, &est);
        if (rc)
            return rc;
        est_total = est;
        if (x && (rc = papr_exchange_stats(x, &est, &est_total, &est_before, nullptr)) != PAPR_OK)
            return xfail(rc);
        std::vector<float> guess;
        int nguess = 0;
        try {
            guess.resize(PAPR_HIP_MAX_LEVELS);
        } catch (...) {
            return fail(ctx, PAPR_E_NOMEM, "out of host memory");
        }
        if (est_total.n)
            nguess = papr_guess_levels(&est_total, graph, graph ? 48.0 : 60.0, guess.data(), PAPR_HIP_MAX_LEVELS);
        if (flags & PAPR_ANALYZE_SPOIL_GUESS)
            for (int j = 0; j < nguess; j++)
                guess[(size_t)j] *= 1.03f;
        (void)papr_hip_set_band(ctx, papr_sweep_band_for(&est_total));
        if (ctx->exact)
            (void)papr_hip_set_exact_hint(ctx, std::isfinite(est_before) && est_before >= 0.0 ? est_before : 0.0);
        rc = papr_hip_stats_sweep(ctx, guess.data(), nguess, &local);  // falls back to plain pass 1 by itself
        swept_path = true;
    } else {
        rc = papr_hip_stats(ctx, &local);
    }
    if (rc)
        return rc;
    ctx->trace.mark("stats_done");
    // ---- exchange 1 + host scalars ----
    papr_stats total = local;
    double before = 0.0;
    if (peer.global) {  // (the records crossed in the stream)
        total = peer.total;
        before = peer.before;
    } else if (x && (rc = papr_exchange_stats(x, &local, &total, &before, nullptr)) != PAPR_OK) {
        return xfail(rc);
    }
    double mean = 0.0;
    float papr = 0.f;
    int L = papr_levels(&total, graph, &mean, &papr, nullptr, 0);
    if (L > cap || L > PAPR_HIP_MAX_LEVELS)
        return fail(ctx, PAPR_E_LIMIT, "%d levels exceed the caller's capacity (%d)", L, std::min(cap, PAPR_HIP_MAX_LEVELS));
    (void)papr_levels(&total, graph, nullptr, nullptr, levels, L);
    ctx->trace.mark("levels");
    // ---- pass 2 (the stash recount when the sweep resolves) and, in exact-sum mode, the sequential sum ----
    bool counted = false;
    if (ctx->exact && std::isfinite(total.sum)) {
        const void *program = nullptr;
        size_t bytes = 0;
        // One-read form: the program is complete before the stash recount has run, so its replay (0.07 ms of dependent
        // additions for a 10 GiB shard) is done inside that window instead of behind it
        if (!replayed)
            ctx->overlap_work = replay_when_ready;
        const int xrc = papr_hip_ccdf_exact(ctx, levels, L, counts_above, before, total.n, &program, &bytes);
        ctx->overlap_work = nullptr;
        ctx->trace.mark("ccdf_exact");
        counted = xrc == PAPR_OK;
        if (!replayed) {
            if (x || xrc == PAPR_OK)
                replay(program, bytes, xrc == PAPR_OK);
            else
                crc = xrc;
        }
        if (crc != PAPR_OK && env_int("PAPR_EXACT_DEBUG", 0))  // (the tree sum stands in; why, for whoever asks)
            fprintf(stderr, "papr: the sequential sum was not reproduced (code %d): %s\n", crc, ctx->err);
        if (crc == PAPR_OK) {
            total.sum = seq;
            res->exact_sum = 1;
            std::vector<float> again;
            try {
                again.resize((size_t)std::min(std::max(papr_levels(&total, graph, nullptr, nullptr, nullptr, 0), 1), PAPR_HIP_MAX_LEVELS));
            } catch (...) {
                return fail(ctx, PAPR_E_NOMEM, "out of host memory");
            }
            const int L2 = papr_levels(&total, graph, &mean, &papr, again.data(), (int)again.size());
            if (L2 > cap || L2 > PAPR_HIP_MAX_LEVELS)
                return fail(ctx, PAPR_E_LIMIT, "%d levels exceed the caller's capacity (%d)", L2, std::min(cap, PAPR_HIP_MAX_LEVELS));
            if (L2 != L || memcmp(again.data(), levels, (size_t)L * sizeof(float)) != 0) {
                // the exact sum moved a float threshold (rare): count again — from the stash if the sweep holds
                L = L2;
                memcpy(levels, again.data(), (size_t)L * sizeof(float));
                counted = false;
                res->pass2_reruns = 1;
            }
        }
    }
    if (!counted) {
        rc = papr_hip_ccdf(ctx, levels, L, counts_above);
        if (rc)
            return rc;
    }
    // ---- exchange 2 (unless the counters are the file's already: they were added up in the stream) ----
    if (x && !ctx->counts_global && (rc = papr_exchange_counts(x, counts_above, L)) != PAPR_OK)
        return xfail(rc);
    ctx->counts_global = false;
    res->total = total;
    res->mean = mean;
    res->papr = papr;
    res->nlevels = L;
    res->swept = swept_path ? ctx->sweep_info.swept : 0;
    res->resolved = swept_path ? ctx->sweep_info.resolved : 0;
    res->reason = swept_path ? ctx->sweep_info.reason : PAPR_SWEEP_NONE;
    res->exact_redo_tiles = ctx->sweep_info.exact_redo_tiles;
    res->band_log2 = ctx->sweep_info.band_log2;
    ctx->trace.mark("leave");
    ctx->trace.dump();
    return PAPR_OK;
}

int papr_hip_analyze(papr_hip_ctx *ctx, papr_exchange *x, int graph, unsigned flags, papr_result *res, float *levels,
                     uint64_t *counts_above, int cap)
{
    return guarded(ctx, [&] { return papr_hip_analyze_impl(ctx, x, graph, flags, res, levels, counts_above, cap); });
}

}  // extern "C"
