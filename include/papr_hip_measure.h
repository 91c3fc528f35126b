/*
 * papr_hip_measure.h — the measurement, tuning and test half of libpaprhip.so's C ABI: nothing a caller of the papr
 * path needs (include/papr_hip.h is that), everything bench.py, the tests, tools/ and `PAPR_STATS=1 bin/papr` use to
 * look inside — kernel timing, ingest timing, what the last sweep did, launch-geometry knobs, synthetic and
 * caller-owned shards, and the GPU-free halves of the one-sweep bookkeeping.
 */
#ifndef PAPR_HIP_MEASURE_H
#define PAPR_HIP_MEASURE_H

#include "papr_exchange.h"
#include "papr_hip.h"
#include "papr_synth.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel timing accumulated since the last reset (HIP events on the context's
 * stream, kernel launches only). */
typedef struct papr_hip_timing {
    double stats_ms;  uint64_t stats_launches;  uint64_t stats_bytes;
    double ccdf_ms;   uint64_t ccdf_launches;   uint64_t ccdf_bytes;
    double exact_ms;  uint64_t exact_launches;  uint64_t exact_bytes; /* exact-sum kernels (classify + segments + groups) */
    double sweep_ms;  uint64_t sweep_launches;  uint64_t sweep_bytes; /* one-sweep kernel (pass 1 + banded pass 2) */
    double aux_ms;    uint64_t aux_launches;    uint64_t aux_bytes;   /* mean-estimate and stash-recount kernels */
} papr_hip_timing;

/* Where the wall time of the last papr_hip_load_file went (seconds). */
typedef struct papr_hip_ingest_timing {
    double total_s;       /* whole call */
    double setup_s;       /* staging buffers, shard allocation, reader threads */
    double read_s;        /* main thread blocked on the file readers */
    double buffer_wait_s; /* main thread blocked on a pinned buffer still being copied */
    double issue_s;       /* hipMemcpyAsync / kernel launch calls */
    double drain_s;       /* final wait for copies + pass-1 kernels + finalize */
    uint64_t bytes, chunks;
    int reader_threads;
    int resident;         /* 1 = shard kept in HBM, 0 = will be re-streamed for pass 2 */
    int o_direct;         /* 1 = the file was read with O_DIRECT (not in the page cache, or PAPR_O_DIRECT=1) */
    int numa_bound;       /* 1 = reader threads and pinned staging buffers sit on the GPU's NUMA node (PAPR_NUMA=0 disables) */
    int io_uring;         /* 1 = the O_DIRECT reads went through one io_uring instead of the reader threads (PAPR_IO_URING=0 disables) */
    int file_passes;      /* whole passes over the shard's file range so far: 1 after the ingest, more when later calls had to re-stream it */
} papr_hip_ingest_timing;

/* Launch geometry knobs.  0 always means "built-in default" (chosen from the
 * 10 GiB sweeps in DESIGN.md section 6); variant and map fields therefore hold
 * id + 1.  Also settable with the PAPR_HIP_TUNE environment variable, e.g.
 * "sblocks=512,svariant=1,smap=0,cblocks=512,cvariant=13,cmap=0,nt=1" (= the defaults on a 256-CU device)
 * ("blocks=" / "variant=" / "map=" set both passes; "wblocks=" / "wvariant=" / "wmap=" / "band=" / "ratio="
 * the one-sweep kernel). */
typedef struct papr_hip_tuning {
    int stats_blocks;   /* pass 1: workgroups per launch */
    int stats_variant;  /* pass 1: kernel geometry variant id + 1 (block x unroll x prefetch form, papr_kernels.hip) */
    int stats_map;      /* pass 1: tile mapping id + 1 (ids: 0 grid-stride, 1 span per workgroup, 2 span per XCD) */
    int ccdf_blocks;    /* pass 2: workgroups per launch */
    int ccdf_variant;   /* pass 2: kernel geometry variant id + 1 */
    int ccdf_map;       /* pass 2: tile mapping id + 1 */
    int nontemporal;    /* 0/1 = default (nontemporal loads), 2 = plain loads */
    int hist_copies;    /* LDS histogram copies per workgroup (1..waves) */
    int flags;          /* bit 0: force the binary-search form of pass 2 (tests) */
    int sweep_blocks;   /* one-sweep kernel: workgroups per launch */
    int sweep_variant;  /* one-sweep kernel: geometry variant id + 1 */
    int sweep_map;      /* one-sweep kernel: tile mapping id + 1 */
    int sweep_band_log2;/* half-width of a threshold band in float bit patterns, log2 (default 14; 8..20) */
    int estimate_ratio; /* papr_hip_estimate reads one 2048-sample tile out of this many (default 64) */
    int reserved;
} papr_hip_tuning;

int papr_hip_abi_version(void);                        /* PAPR_HIP_ABI_VERSION the library was built from */
int papr_hip_device_name(const papr_hip_ctx *ctx, char *buf, int buflen);
int papr_hip_set_tuning(papr_hip_ctx *ctx, const papr_hip_tuning *t);
/* 1 if this build of the library carries the one-sweep kernel form with that id: the product has two (111:
 * papr_sweep_kernel, 131: papr_sweep3_kernel for exact-sum mode); `make MEASURE=1` adds the laboratory's
 * (csrc/measure/papr_sweep_lab.hip: other geometries, stash forms, ablations, the predecessors) */
int papr_hip_sweep_variant_built(int variant);
/* Kernel timing (papr_hip_get_timing): 0 off, 1 every timed kernel, 2 only the kernels that read the shard (pass 1,
 * pass 2, the sweep, the exact-sum pass) — a timed kernel carries a completion signal of its own, which costs the stream
 * ~5 us on either side of it (profiles/r02_step_timeline.txt), so a benchmark times the small estimate / recount
 * kernels in separate steps.  Also resets the counters. */
int papr_hip_set_timing(papr_hip_ctx *ctx, int enabled);
int papr_hip_get_timing(papr_hip_ctx *ctx, papr_hip_timing *out);
/* The same launches one by one, in dispatch order: durations (ms) of the timed launches of one class (0 pass 1, 1 pass 2,
 * 2 exact-sum kernels, 3 the sweep, 4 estimate / recount) into ms[0 .. cap); returns how many there were (may exceed cap),
 * or a negative PAPR_E_* code.  A benchmark reports min / median / max of its dominant kernel from this. */
int papr_hip_get_timing_launches(papr_hip_ctx *ctx, int kind, float *ms, int cap);

int papr_hip_get_ingest_timing(const papr_hip_ctx *ctx, papr_hip_ingest_timing *out);

/* Copy nsamples IQ pairs from host memory into the shard. */
int papr_hip_upload(papr_hip_ctx *ctx, const float *iq, uint64_t nsamples, uint64_t base_index);

/* Use caller-owned device memory (16-byte aligned, on this context's GPU) as
 * the shard, without copying.  The memory must stay valid until the next
 * load/upload/adopt/close. */
int papr_hip_adopt(papr_hip_ctx *ctx, void *device_iq, uint64_t nsamples, uint64_t base_index);

/* Fill the shard with synthetic samples [first_index, first_index+nsamples)
 * of the stream defined by include/papr_synth.h.  If a buffer of at least that
 * size was adopted it is filled in place, otherwise the context allocates. */
int papr_hip_generate(papr_hip_ctx *ctx, const papr_synth_spec *spec, uint64_t first_index, uint64_t nsamples);

/* Copy shard samples [first, first+nsamples) (shard-relative) back to the host (tests). */
int papr_hip_download(papr_hip_ctx *ctx, float *iq, uint64_t first, uint64_t nsamples);

/* what the last one-sweep pass / papr_hip_ccdf did */
typedef struct papr_hip_sweep_info {
    uint64_t stash_samples;  /* in-band samples the last sweep produced */
    uint64_t stash_capacity;
    uint64_t estimate_samples; /* samples the last papr_hip_estimate read */
    int swept;               /* last papr_hip_stats_sweep: 1 = banded + stashed, 0 = plain pass 1 */
    int resolved;            /* last papr_hip_ccdf: 1 = answered from the sweep, 0 = read the shard again */
    int reason;              /* PAPR_SWEEP_*: why not */
    int band_log2;
    uint32_t exact_redo_tiles; /* exact-sum mode, last papr_hip_ccdf_exact / _exact_program after a sweep: 2048-sample tiles whose
                                * speculated running-sum binade was wrong and whose rounding functions were rebuilt */
    uint32_t gave_up;        /* waves that gave the sweep up for their workgroup because most of what it folded was in band
                              * (constant-envelope captures); any > 0 shows as PAPR_SWEEP_STASH_FULL */
    int kernel_variant;      /* id of the kernel form the last sweep was launched as (papr_sweep.hip's tables); measurements
                              * quote it so that numbers taken with another form are recognised as stale */
    int xcd_first;           /* XCD the context's stream puts workgroup 0 of a launch on (-1: not asked yet): its parity decides
                              * which workgroups take the sweep kernels' XCD skew */
} papr_hip_sweep_info;
int papr_hip_get_sweep_info(const papr_hip_ctx *ctx, papr_hip_sweep_info *out);
/* When every workgroup of the last single-wait step's sweep launch was done, and where it ran: bits 27:0 the 100 MHz real-time
 * counter (10 ns ticks) as the workgroup left, bits 31:28 the XCD it ran on (HW_REG_XCC_ID), in workgroup order — the kernels are ONE persistent workgroup per CU over a static
 * share of the shard, so the launch lasts as long as its slowest workgroup, and the spread says how much of it the others
 * idle.  Returns the number of workgroups (writes min(that, cap) ticks), 0 if no sweep was launched, or a negative code. */
int papr_hip_get_wg_finish(papr_hip_ctx *ctx, uint32_t *ticks, int cap);

/* The GPU-free halves of the one-sweep bookkeeping — what the runtime itself calls, exported so that the
 * speculation's logic can be checked without a GPU (tests/test_sweep_host.py):
 *   papr_level_key      smallest bit pattern of a non-negative float that is > level (the integer compare
 *                       `bits(v) >= key` is the reference's float compare `v > level`); 0 for negative levels,
 *                       UINT32_MAX when nothing can exceed the level (NaN, +Inf)
 *   papr_sweep_bands    guessed levels -> unique ascending keys and the band edges key -+ 2^band_log2
 *                       (edges[2j], edges[2j+1]); returns the number of keys, or 0 when the guess has no band
 *                       form (no usable level, denormal/huge thresholds, bands that touch)
 *   papr_sweep_resolve  counts_above[l] for the true levels from what a sweep left behind: above_band[j] =
 *                       samples at or above band j's upper edge, stash_above[l] = stash powers > levels[l].
 *                       Returns 1, or 0 if some true level lies in no band (then nothing is written). */
uint32_t papr_level_key(float level);
int papr_sweep_bands(const float *guess_levels, int nlevels, int band_log2, uint32_t *keys, uint32_t *edges);
int papr_sweep_resolve(const uint32_t *guess_keys, int nguess, int band_log2, const uint64_t *above_band,
                       const float *levels, int nlevels, const uint64_t *stash_above, uint64_t *counts_above);

typedef struct papr_exchange_timing { /* host wall time spent inside the exchanges since open / the last reset */
    uint64_t stats_calls, counts_calls, exact_calls;
    double stats_us, counts_us, exact_us;
    uint64_t in_stream_calls; /* collectives queued on the context's stream between kernels (RCCL transport, papr_hip_analyze's
                               * single-wait step): no host staging and no wait of their own, so no time to report */
} papr_exchange_timing;
int papr_exchange_get_timing(papr_exchange *x, papr_exchange_timing *out, int reset);

#ifdef __cplusplus
}
#endif
#endif /* PAPR_HIP_MEASURE_H */
