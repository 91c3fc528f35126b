/*
 * papr_exchange.h — the exchange between the shards of one file (one process, or one thread, per GPU): part of
 * libpaprhip.so's C ABI, used together with include/papr_hip.h (papr_hip_analyze takes a papr_exchange).
 */
#ifndef PAPR_EXCHANGE_H
#define PAPR_EXCHANGE_H

#include "papr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the exchange between shards: one process per GPU ---------------------------------
 * The sample axis shards with no bulk exchange; what crosses GPUs is three tiny, latency-bound messages:
 *   papr_exchange_stats     after pass 1 / the sweep (also for the mean estimate before it): all-gather of one
 *                           96-byte papr_stats per rank, folded in rank (= file) order with papr_stats_merge on
 *                           every rank — first-index extrema are not a reduction op RCCL has, and a fixed order
 *                           keeps the double sum identical everywhere.  Also returns the sum of the ranks in
 *                           front of this one (what the exact-sum path and its hint want).
 *   papr_exchange_counts    after pass 2: all-reduce (sum) of the per-level counters, as 64-bit integers
 *   papr_exchange_exact_sum exact-sum mode: all-gather of the shards' sum programs (<= ~1 MB), chained in rank
 *                           order with papr_exact_chain -> the reference's sequential sum on every rank
 * Transports: RCCL over xGMI (papr_exchange_open_rccl: ncclAllGather / ncclAllReduce on the context's stream,
 * device staging buffers, one stream synchronisation per exchange; the 128-byte ncclUniqueId from
 * papr_exchange_unique_id on rank 0 is handed to the other ranks by whatever launched them), or any pair of
 * collectives the caller supplies (papr_exchange_open_ops: the tests run the same code over gloo on CPUs). */
typedef struct papr_exchange_ops {
    void *user;
    int (*allgather)(void *user, const void *send, void *recv, size_t bytes_per_rank); /* recv: world x bytes_per_rank */
    int (*allreduce_sum_u64)(void *user, uint64_t *buf, size_t count);                 /* in place */
} papr_exchange_ops;
#define PAPR_EXCHANGE_ID_BYTES 128
int papr_exchange_unique_id(void *id /* PAPR_EXCHANGE_ID_BYTES */);
int papr_exchange_open_rccl(papr_exchange **x, papr_hip_ctx *ctx, const void *id, int rank, int world);
int papr_exchange_open_ops(papr_exchange **x, const papr_exchange_ops *ops, int rank, int world);
void papr_exchange_close(papr_exchange *x); /* before or after papr_hip_close of the context it was opened / bound with: either order */
const char *papr_exchange_last_error(const papr_exchange *x); /* x may be NULL: last open error */
int papr_exchange_stats(papr_exchange *x, const papr_stats *local, papr_stats *total, double *sum_before,
                        papr_stats *all /* world records in rank order, or NULL */);
int papr_exchange_counts(papr_exchange *x, uint64_t *counts, int n);
int papr_exchange_exact_sum(papr_exchange *x, const void *program, size_t bytes, double *sum);

/* An in-process transport for papr_exchange: n handles for n threads of ONE process that each drive one GPU (what
 * bin/papr does): the same exchange calls, met at a barrier instead of on a wire. xs receives n handles. */
int papr_exchange_open_local(papr_exchange **xs, int n);
/* The same n handles for n threads of one process, with RCCL underneath ("partial peak / mean / histogram
 * reduced over RCCL / xGMI"; what bin/papr uses when it drives more than one GPU, PAPR_XCH=threads keeps the plain hub):
 * every thread, once it has opened its context, calls papr_exchange_bind(x, ctx) — all n of them, at about the same time:
 * it is ncclCommInitRank on the thread's own device, which returns when every rank has joined.  From then on the step's
 * exchanges are ncclAllGather / ncclAllReduce on device buffers, queued on each context's stream between the kernels that
 * produce and consume them (papr_hip_analyze: one wait per step); the few host-level exchanges that remain (agreeing on a
 * path, programs that outgrew their slot) still meet at the in-process hub.  If two shards share a device (one
 * communicator cannot hold two ranks of one GPU) bind leaves the handles as papr_exchange_open_local made them.
 * papr_exchange_bind is a no-op for handles of the other transports; papr_exchange_is_rccl tells what a handle became. */
int papr_exchange_open_rccl_local(papr_exchange **xs, int n);
int papr_exchange_bind(papr_exchange *x, papr_hip_ctx *ctx);
/* The same with the communicators coming up BESIDE the ingest — what bin/papr does: ncclCommInitRank costs seconds around a
 * step of milliseconds, and the first collective is only needed when the shards are loaded.  _open_rccl_local_async returns
 * at once with n handles that are papr_exchange_open_local's (host-level exchanges meet at the hub from the start) and has
 * started n threads that load librccl, select devices[r] and join the communicator; it does not fail for want of RCCL.
 * Every shard's thread later calls papr_exchange_adopt_rccl(x, ctx, ...) — all n at the same point of their sequence: it
 * waits up to timeout_s (0: not at all) for this rank's communicator, the threads agree through the hub, and either ALL
 * handles take their communicators (papr_exchange_is_rccl: the step's exchanges are collectives on the contexts' streams
 * from then on) or none does — librccl missing, a failed or slow ncclCommInitRank: the handles stay the hub's and
 * papr_exchange_last_error(x) says why (nothing is printed: bin/papr prints it when RCCL was asked for by name).  Always PAPR_OK unless the hub itself was cancelled.  *setup_s: how long this rank's
 * set-up thread ran (0 if it is not done), *waited_s: how long this call waited for it.  Shards that share a device:
 * no threads are started, adopt is a no-op.  (Tests: PAPR_XCH_BIND_FAIL=all|<rank> injects a failure,
 * PAPR_XCH_BIND_DELAY_MS a slow set-up.) */
int papr_exchange_open_rccl_local_async(papr_exchange **xs, int n, const int *devices);
int papr_exchange_adopt_rccl(papr_exchange *x, papr_hip_ctx *ctx, double timeout_s, double *setup_s, double *waited_s);
int papr_exchange_is_rccl(const papr_exchange *x);
/* A rank that cannot go on cancels the exchange so that its peers are released instead of waiting for it: the threads of
 * the in-process transports get PAPR_E_STATE from their pending and future exchange calls, and RCCL communicators — this
 * rank's, and with papr_exchange_open_rccl_local every thread's — are aborted (ncclCommAbort), which ends collectives that
 * are already queued on the peers' streams.  No-op for caller-supplied collectives.  libpaprhip calls it itself when a
 * step fails locally after its collectives have begun (papr_hip_analyze: FAILURE WITH PEERS). */
void papr_exchange_abort(papr_exchange *x);
/* Self-test: every collective the sharded step uses, once, on tiny buffers whose contents every rank can predict for every
 * other rank — host-level all-gather (96 B) and all-reduce (301 x u64), and, where the transport runs collectives in the
 * context's stream (RCCL; PAPR_XCH_IN_STREAM=2: the stand-ins), the in-stream all-gather, the all-gather-v with unequal
 * sizes (one group of ncclBroadcasts) and the all-reduce on device buffers.  All ranks call it together.  verbose: rank 0
 * prints one stderr line per collective with its microseconds; a failure names the collective on the rank that saw it,
 * cancels the exchange (papr_exchange_abort) and returns PAPR_E_STATE.  ctx may be NULL (host-level collectives only).
 * PAPR_XCH_SELFTEST=1 makes papr_hip_analyze run it once per handle in front of its first step with peers (bin/papr,
 * bench.py): a first run on N GPUs that goes wrong then says which collective did instead of hanging in the step.
 * PAPR_XCH_TIMEOUT_S=<seconds> (off by default): a rank that waits longer than that for its peers — at the in-process hub,
 * in a host-level exchange, in the step's one wait behind in-stream collectives — says so on stderr and cancels the
 * exchange: the wait returns PAPR_E_STATE and the peers are released.  Inside ncclCommInitRank, where there is no
 * communicator to abort yet, the process is ended with status 254 (PAPR_XCH_TIMEOUT_EXIT=0: only the message). */
int papr_exchange_selftest(papr_exchange *x, papr_hip_ctx *ctx, int verbose);

#ifdef __cplusplus
}
#endif
#endif /* PAPR_EXCHANGE_H */
