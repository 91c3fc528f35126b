// papr_kernels.h — shared between the device code (papr_kernels.hip) and the
// host runtime (papr_runtime.cpp, papr_ingest.cpp, papr_sweep_rt.cpp, papr_exact_rt.cpp).  Internal: not part of the C ABI.
#ifndef PAPR_KERNELS_H
#define PAPR_KERNELS_H

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "papr_synth.h"

// Launch geometry.  The default streaming variant uses 4-wave workgroups whose
// loop iteration consumes one tile = 8 coalesced 4 KiB rows (256 lanes x 16 B),
// i.e. 32 KiB = 4096 IQ samples; other (block, unroll) variants exist for
// measurement (papr_variant_geometry).  PAPR_BLOCK is the workgroup size of
// the small helper kernels; PAPR_TILE_SAMPLES_MAX bounds every variant's tile
// (shard slack, chunk alignment).
#define PAPR_BLOCK 256
#define PAPR_TILE_SAMPLES_MAX 8192

// Exact-sum mode geometry (papr_exact.hip): pass 1 runs as 256 threads x 4
// loads, i.e. tiles of 2048 samples (16 KiB) with 4 per-wave sums each; the
// rounding-function kernel works on segments of half a tile (one wave, 16
// samples per lane); segment functions are pre-composed in groups of 128 tiles.
#define PAPR_EXACT_TILE_SAMPLES 2048
#define PAPR_EXACT_TILE_WAVES 4
#define PAPR_EXACT_SEG_SAMPLES 1024
#define PAPR_EXACT_GROUP_TILES 128
#define PAPR_EXACT_AMBIG (-2147483647 - 1) /* tile_E: entry binade not provable, or the tile crosses a binade */
#define PAPR_EXACT_ZERO (-2147483647)      /* tile_E: every power in the tile is +0: the running sum cannot change */

// One-sweep mode (papr_sweep.hip): the mean estimate reads one 2048-sample tile out of every
// `ratio`; every wave of the sweep kernel compacts in-band powers into an LDS slice of
// papr_sweep_slice_floats(unroll) floats (1.5 x what one tile can add: it spills once a third is
// used; LDS per workgroup decides how many waves a CU holds) before spilling to HBM.
#define PAPR_ESTIMATE_TILE_SAMPLES 2048
constexpr uint32_t papr_sweep_slice_floats(int unroll)
{
    return 3u * (uint32_t)unroll * 64u;
}

#define PAPR_MAP_GRID_STRIDE 0
#define PAPR_MAP_BLOCK_SPAN 1
#define PAPR_MAP_XCD_SPAN 2

// One workgroup's (or the final) pass-1 record.
// val/idx order: peak power, re_pos, re_neg, im_pos, im_neg.
struct papr_partial {
    double sum;
    uint64_t idx[5];
    float val[5];
    uint32_t pad;
};

struct papr_ccdf_params {
    uint32_t shift;       // LUT: bit patterns per cell = 1 << shift
    uint32_t cell_lo;     // LUT: first cell in the table
    uint32_t ncells;      // LUT: cells in the table
    uint32_t nkeys;       // unique thresholds (histogram has nkeys + 1 bins)
    uint32_t above_lo;    // LUT: first bit pattern past the table
    uint32_t above_count; // LUT: patterns in [above_lo, +Inf] (0 when the table reaches past +Inf)
    uint32_t table_words; // 32-bit words of table to stage into LDS
    uint32_t copies;      // LDS histogram copies per workgroup
    uint32_t search_step; // search: largest power of two <= nkeys
};

// exact-sum mode: one pre-composed group of PAPR_EXACT_GROUP_TILES tiles
struct papr_exact_group {
    int32_t E;     // binade of the running sum throughout the group, PAPR_EXACT_ZERO, or PAPR_EXACT_AMBIG (= mixed)
    int32_t pad;
    double D0, D1; // total increment for even / odd entry parity
};

struct papr_exact_plan {
    uint32_t nmixed, nraw, overflow, pad;
};

/* what the pack kernel needs to place a raw tile's 16-sample runs in their binades (an approximate prefix at the tile):
 * the sums papr_exact_classify scanned — kind 1: seg_D (two (D0, D1) per tile), kind 2: pass 1's four wave sums per tile,
 * 0: none (the runs then travel without pairs) — its block sums, and what lies in front of the shard */
struct papr_exact_prefix_src {
    const double *sums;
    const double *block_sums;
    const double *before_dev;
    double before;
    int kind;
    int groups_in_place; /* the program's group table was written by papr_launch_exact_groups (its `program` argument) */
};
void papr_launch_exact_pack(hipStream_t st, const papr_exact_group *groups, uint64_t ngroups, const int32_t *tile_E,
                            uint64_t ntiles, const void *seg_D, const void *data, const void *raw_store,
                            const void *tail_src, uint64_t nsamples, uint32_t tail_samples, uint32_t *mixed_list,
                            uint32_t cap_mixed, uint32_t *raw_list, uint32_t cap_raw, papr_exact_plan *plan,
                            unsigned char *out_mapped,
                            const uint32_t *count_src = nullptr /* one word (the redo count) copied along ... */,
                            uint32_t *count_dst = nullptr /* ... to mapped host memory */,
                            uint64_t out_cap = 0 /* bytes `out` can hold (0: as much as any program needs) */,
                            uint32_t redo_cap = 0 /* with count_src: a larger count marks the program as not final */,
                            papr_exact_prefix_src prefix = papr_exact_prefix_src{nullptr, nullptr, nullptr, 0.0, 0, 0});
/* peers: the used bytes of every rank's program slot (device, after the all-gather) into the same slot of mapped host memory */
struct papr_xprog_layout {
    uint32_t world, pad;
    uint64_t offs[65];  /* slot r = bytes [offs[r], offs[r + 1]) (8-byte aligned; at most 64 ranks) */
};
void papr_launch_exact_programs_to_host(hipStream_t st, const void *slots, const papr_xprog_layout &lay, void *host_mapped);
/* ambig_* may be null (resident shards); otherwise the unprovable tiles are also listed, ascending, in ambig_sorted */
void papr_launch_exact_classify(hipStream_t st, const double *tile_wave_sums, uint64_t ntiles, double *block_sums,
                                double before, double delta, int32_t *tile_E, uint32_t *ambig_list, uint32_t ambig_cap,
                                uint32_t *ambig_count, uint32_t *ambig_sorted);
/* one-read sweep (papr_sweep2_kernel<EXACT>): speculated binades before it, true classification + redo after it */
void papr_launch_exact_spec(hipStream_t st, const double *group_sums, uint64_t ngroups, uint32_t ratio, double scale,
                            double before, double *group_prefix, uint64_t ntiles, int32_t *spec,
                            const double *before_dev = nullptr /* overrides `before` (peers: known on the device only) */,
                            bool scan_done = false /* papr_launch_guess_bands already made group_prefix (its spec_* arguments) */);
void papr_launch_exact_fill_spec(hipStream_t st, int32_t *spec, uint64_t ntiles, int32_t E);
/* ambig_* may be null (resident shards); otherwise the unprovable tiles are also listed, ascending, in ambig_sorted */
void papr_launch_exact_classify_swept(hipStream_t st, const void *seg_D, uint64_t ntiles, double *block_sums, double before,
                                      double delta, int32_t *tile_E, const int32_t *spec, uint32_t *redo_list,
                                      uint32_t redo_cap, uint32_t *redo_count, uint32_t *ambig_list, uint32_t ambig_cap,
                                      uint32_t *ambig_count, uint32_t *ambig_sorted,
                                      const double *before_dev = nullptr /* overrides `before` ... */,
                                      const unsigned long long *n_total_dev = nullptr /* ... and `delta` (from the file's length) */,
                                      uint64_t n_shard = 0);
/* compact != 0: the k-th listed tile's samples are the k-th tile of `data` (tiles read back from the file) */
void papr_launch_exact_redo(hipStream_t st, int blocks, const void *data, const int32_t *tile_E, void *seg_D,
                            const uint32_t *tile_list, const uint32_t *tile_count, uint32_t list_cap, int compact);
void papr_launch_exact_segsums_to_tilesums(hipStream_t st, const void *seg_D, uint64_t ntiles, double *tile_wave_sums);
void papr_launch_exact_capture(hipStream_t st, const void *chunk, uint64_t chunk_tile0, uint64_t chunk_ntiles,
                               const uint32_t *sorted, const uint32_t *count, uint32_t cap, void *raw_store);
void papr_launch_exact_segments(hipStream_t st, int blocks, const void *data, uint64_t nsegs, const int32_t *tile_E,
                                void *seg_D);
size_t papr_exact_transpose_lds_bytes(void); /* of the fused sweep's workgroup */
int papr_exact_fused_waves(void);
void papr_launch_exact_segments_ccdf(hipStream_t st, int blocks, const void *data, uint64_t nsegs,
                                     const int32_t *tile_E, void *seg_D, const void *tail, uint32_t tail_samples,
                                     const uint32_t *table, const papr_ccdf_params &P, size_t lds_table_bytes,
                                     unsigned long long *ghist);
void papr_launch_exact_groups(hipStream_t st, const int32_t *tile_E, uint64_t ntiles, const void *seg_D,
                              uint64_t ngroups, papr_exact_group *out,
                              unsigned char *program = nullptr /* also into this sum program's group table */);

int papr_variant_geometry(int variant, int *block, int *unroll); /* 0, or -1 for an unknown variant */
void papr_launch_stats(hipStream_t st, int variant, int blocks, bool nt, const void *data, uint64_t ntiles,
                       uint64_t base_index, int map, papr_partial *out);
void papr_launch_stats_tilesums(hipStream_t st, int blocks, const void *data, uint64_t ntiles, uint64_t base_index,
                                int map, papr_partial *out, double *tile_sums, uint64_t tile_offset);
void papr_launch_stats_finalize(hipStream_t st, const void *tail, uint32_t tail_samples, uint64_t tail_base_index,
                                const papr_partial *partials, uint32_t npartials, papr_partial *result,
                                papr_partial *result_dev = nullptr /* a second copy of the record, in device memory */,
                                const unsigned long long *copy_src = nullptr /* copy_words words copied to copy_dst ... */,
                                unsigned long long *copy_dst = nullptr /* ... (mapped host memory) on the way */,
                                uint32_t copy_words = 0);
void papr_launch_first_nan(hipStream_t st, int blocks, const void *data, uint64_t nsamples, uint64_t base_index,
                           unsigned long long *key);
// bytes (a multiple of 8) from mapped pinned host memory to device memory, by a kernel (the ingest's H2D leg)
void papr_launch_pull(hipStream_t st, const void *src_mapped, void *dst, uint64_t bytes);
void papr_launch_ccdf(hipStream_t st, int variant, int blocks, bool nt, bool lut, size_t lds_bytes, const void *data,
                      uint64_t ntiles, int map, const void *tail, uint32_t tail_samples, const uint32_t *table,
                      const papr_ccdf_params &P, unsigned long long *ghist);
void papr_launch_generate(hipStream_t st, int blocks, void *out, uint64_t nsamples, uint64_t first_index,
                          const papr_synth_spec &spec);
/* one-sweep mode (papr_sweep.hip) */
/* Kernel timing without marker packets: the NEXT papr_launch_estimate / _sweep / _sweep2 / _ccdf_power call of this
 * thread binds the two events to its dispatch (hipExtLaunchKernelGGL: both take the kernel's own start and end from its
 * completion signal), once.  hipEventRecord in front of and behind a kernel costs a barrier packet each (~5 us of
 * stream time per record on this chip, profiles/r02_step_timeline.txt). */
struct papr_launch_timer {
    hipEvent_t start, stop;
};
void papr_time_next_launch(const papr_launch_timer *t);
void papr_launch_estimate(hipStream_t st, int blocks, const void *data, uint64_t ngroups, uint32_t ratio,
                          papr_partial *out, double *group_sums /* may be null: 4 sampled sums per group */,
                          double *block_sq /* may be null: per workgroup, sum of squared piece sums */);
/* papr_guess_bands_kernel: what the host half of the speculation (papr_guess_levels, papr_sweep_band_for,
 * papr_sweep_bands, the compact LUT plan) produces, made on the device between the estimate kernel and the sweep
 * kernel so that the step needs no host round trip there.  Written to device memory (the sweep kernel reads P) and to
 * mapped host memory (the host reads the rest after the sweep). */
#define PAPR_GUESS_MAX_BANDS 512
struct papr_guess_out {
    papr_ccdf_params P;     /* LUT form of the band edges (nkeys = 2 * nbands); an empty table if !ok */
    uint32_t ok;            /* 1: bands, table and keys are valid; 0: the guess has no band form (plain pass 1 ran) */
    uint32_t band_log2;
    uint32_t nbands;
    uint32_t pad;
    double est_sum;         /* sampled sum scaled to the shard */
    double est_rel_se;      /* relative standard error of the estimated mean */
    double est_before;      /* peers: the estimated sum of the shards in front of this one */
    uint32_t gkeys[PAPR_GUESS_MAX_BANDS]; /* keys of the guessed thresholds, ascending */
};
/* With peers (papr_exchange over RCCL) the step's exchanges are collectives queued on the stream, between these:
 *   papr_est_record_kernel  one 48-byte record of this shard's estimate -> all-gather -> papr_guess_bands_kernel(recs)
 *   papr_record_merge_kernel after the all-gather of the shards' pass-1 records: the ordered merge (rank = file order;
 *                           papr_stats_merge's rules), the sum of the shards in front of this one, the file's length
 *   papr_xpack_kernel       the sweep's bins, the recount's bins and four flags as ONE vector for the all-reduce */
struct papr_est_record {
    double S, sq;                          /* sampled sum; sum of squared piece sums */
    unsigned long long sampled, n, pieces; /* samples read; samples in the shard; pieces (4 per group) */
    unsigned long long ratio;              /* 1 = everything was read */
    unsigned long long flags;              /* the shard's PAPR_FLAG_* (odd tail) */
    unsigned long long pad;
};
struct papr_peer_out {
    papr_partial total;            /* the file's pass-1 record */
    double before;                 /* sum of the shards in front of this one */
    unsigned long long n_total;    /* samples in the file */
    unsigned long long nan_ranks;  /* shards whose sum came out NaN (then the host path takes over, on every rank) */
    unsigned long long flags;      /* OR of the shards' flags */
};
#define PAPR_XVEC_FLAGS 4 /* [0] stash overflow / give-up on some rank, [1] guess without a band form, [2] no speculated table, [3] reserved */
void papr_launch_est_record(hipStream_t st, const papr_partial *est_partials, const double *est_sq, uint32_t est_blocks,
                            uint64_t ngroups, uint64_t sampled, uint64_t nsamples, uint32_t ratio, uint32_t flags,
                            papr_est_record *out);
void papr_launch_record_merge(hipStream_t st, const papr_partial *recs, const papr_est_record *est, uint32_t world, uint32_t rank,
                              papr_partial *total_dev, unsigned long long *n_total_dev, papr_peer_out *out_host);
void papr_launch_xpack(hipStream_t st, const unsigned long long *sweep_hist, uint32_t sweep_words, const unsigned long long *seg_fill,
                       uint32_t nsegs, uint64_t seg_cap, const unsigned long long *gave_up, const unsigned long long *recount_hist,
                       uint32_t recount_words, const struct papr_guess_out *guess, const struct papr_true_out *tru,
                       unsigned long long *vec);
#define PAPR_POW_TABLE 2048 /* entries per mode of the pow(10, x_j) table (papr_host.c: PAPR_POW_CACHE) */
void papr_launch_guess_bands(hipStream_t st, const papr_partial *est_partials, const double *est_sq, uint32_t est_blocks,
                             uint64_t ngroups, uint64_t sampled, uint64_t nsamples, uint32_t ratio, int graph, float max_db,
                             float spoil, int band_override, uint32_t copies, int compact /* LUT form: two edges per cell */,
                             uint32_t soft_lds /* bytes table + histogram copies should stay under */, uint32_t *table,
                             uint32_t table_cap_words, papr_guess_out *out_dev, papr_guess_out *out_host,
                             unsigned long long *zero /* words the kernel clears on its way */, uint32_t zero_words,
                             const papr_est_record *recs = nullptr /* all shards' estimate records, rank order (peers) */,
                             uint32_t nrecs = 0, uint32_t my_rank = 0,
                             const double *spec_group_sums = nullptr /* exact-sum mode without peers: a SECOND workgroup scans the */,
                             uint64_t spec_ngroups = 0 /* estimate's per-group sums for the binade speculation meanwhile */,
                             double spec_scale = 0.0, double *spec_group_prefix = nullptr,
                             const double *pow_tab = nullptr /* [2][PAPR_POW_TABLE]: the host libm's pow(10, x_j) (papr_host.c) */);
/* the product form of the sweep: papr_sweep_kernel = 512 threads x 8 loads per lane (64 KiB tiles), one persistent
 * workgroup per CU, 12 KiB of stash slice per wave; its variant id, and papr_sweep3_kernel's (exact-sum mode) */
#define PAPR_SWEEP_THREADS 512
#define PAPR_SWEEP_LOADS 8
#define PAPR_SWEEP_SLICE_FLOATS 3072u
#define PAPR_SWEEP_VARIANT 111
#define PAPR_SWEEP3_VARIANT 131
uint32_t papr_sweep_xcd_skew_rounds(uint64_t ntiles, int blocks, int kind);
#define PAPR_MAP_EVEN_SLOW (1 << 30)  /* papr_launch_sweep's map: workgroup 0 sits on an odd XCD, the EVEN workgroups take the skew */
void papr_launch_xcd_probe(hipStream_t st, unsigned long long *out);
int papr_sweep_variant(int variant); /* the sweep geometry used for a variant id, or -1 */
#define PAPR_SWEEP_VARIANT_IS_LUT2(v) (((v) >= 20 && (v) <= 29) || ((v) >= 70 && (v) <= 79) || (v) == 18 || (v) == 38) /* compact table: papr_sweep_kernel<LUT2>, papr_sweep_split_kernel */
#define PAPR_SWEEP_VARIANT_IS_PERSISTENT(v) ((v) == PAPR_SWEEP_VARIANT || (v) == 40 || (v) == 114 || ((v) >= 120 && (v) <= 129) || ((v) >= 140 && (v) <= 154)) /* launched as ONE workgroup per CU (512 threads x 8 loads per lane) */
#define PAPR_SWEEP_VARIANT_HAS_HIST_SETS(v) (((v) >= 84 && (v) <= 85) || ((v) >= 87 && (v) <= 89)) /* SMODE bit 2 (measurement variants) */
int papr_sweep_geometry(int variant, int *threads, uint64_t *tile_samples, size_t *stash_lds); /* 0, or -1 */
void papr_launch_sweep(hipStream_t st, int variant, int blocks, size_t lds_bytes, const void *data, uint64_t ntiles,
                       uint64_t base_index, int map, papr_partial *out, const void *tail, uint32_t tail_samples,
                       const uint32_t *table, const papr_ccdf_params &P, unsigned long long *ghist, float *stash,
                       unsigned long long *seg_counts, uint64_t seg_cap, unsigned long long *gave_up,
                       unsigned long long *seg_real /* per workgroup: powers stashed, without padding */,
                       const papr_ccdf_params *Pdev /* null, or the table geometry in device memory (overrides P) */);
/* (the geometry follows from the chip: 2 workgroups of 1024 threads per CU, segments split or shared to match) */
void papr_launch_ccdf_power(hipStream_t st, int num_cus, bool lut, size_t lds_bytes, const float *stash,
                            const unsigned long long *seg_counts, uint64_t seg_cap, uint32_t nsegs,
                            const uint32_t *table, const papr_ccdf_params &P, unsigned long long *ghist,
                            const papr_ccdf_params *Pdev /* null, or the table geometry in device memory (overrides P) */);
/* papr_true_table_kernel: the reference's level table from the pass-1 record, with the device's libm, and the recount
 * LUT for it — a speculation the host checks against papr_levels bit for bit */
#define PAPR_TRUE_MAX_LEVELS 1024
struct papr_true_out {
    papr_ccdf_params P;
    uint32_t ok;       /* 1: levels and table are there (normal, increasing levels with a LUT form) */
    uint32_t nlevels;
    uint32_t pad[2];
    float levels[PAPR_TRUE_MAX_LEVELS];  /* (host copy only) */
};
void papr_launch_true_table(hipStream_t st, const papr_partial *result, uint64_t nsamples, int graph, uint32_t copies,
                            uint32_t soft_lds, uint32_t *table, uint32_t table_cap_words, papr_true_out *out_dev,
                            papr_true_out *out_host, unsigned long long *zero, uint32_t zero_words,
                            const unsigned long long *gave_up /* the sweep's give-up counter: non-zero = nothing to recount */,
                            const unsigned long long *nsamples_dev = nullptr /* peers: the file's length, in device memory */,
                            const double *pow_tab = nullptr /* as papr_launch_guess_bands */);
void papr_sweep_prepare_device(void);
#ifdef PAPR_MEASURE /* measure/papr_sweep_lab.hip: every other kernel form, behind the same variant ids */
int papr_lab_sweep_variant(int variant);
int papr_lab_sweep_geometry(int variant, int *threads, uint64_t *tile_samples, size_t *stash_lds);
void papr_lab_launch_sweep(hipStream_t st, int variant, int blocks, size_t lds_bytes, const void *data, uint64_t ntiles,
                           uint64_t base_index, int map, papr_partial *out, const void *tail, uint32_t tail_samples,
                           const uint32_t *table, const papr_ccdf_params &P, unsigned long long *ghist, float *stash,
                           unsigned long long *seg_counts, uint64_t seg_cap, unsigned long long *gave_up,
                           unsigned long long *seg_real, const papr_ccdf_params *Pdev);
void papr_lab_prepare_device(void);
#endif

// ---- one-sweep kernel, second generation (papr_sweep.hip: papr_sweep2_kernel) -----------------------------------
// Every WAVE owns whole segments of 64 * U float4 (U = 8: 1024 samples, 8 KiB): wave w of workgroup b takes segment
// (it * gridDim + b) * WAVES + w.  The band-edge LUT is "compact": a cell of 2^shift bit patterns may hold up to TWO
// edges — entry = { below << 22 | off1, off2 } with off = edge & (2^shift - 1), PAPR_LUT2_NEVER where there is none — so
// the cell size no longer has to shrink with the band width.
#define PAPR_LUT2_OFF_BITS 22
#define PAPR_LUT2_NEVER 0x3FFFFFu
#define PAPR_LUT2_MAX_SHIFT 21
#define PAPR_LUT2_MAX_EDGES 1022       /* `below` has 10 bits */
#define PAPR_SWEEP2_SPILL 256u         /* floats per full stash spill: one 16-byte store per lane */
#define PAPR_STASH_PAD_BITS 0x7FC00000u /* quiet NaN: pads a partial spill to 16 bytes; the recount ignores NaN */

struct papr_sweep2_params {
    const void *data;             // first sample of the launch (16-byte aligned)
    uint64_t nsegs;               // whole segments in the launch
    uint64_t base_index;          // global index of sample 0 of `data`
    papr_partial *out;            // one record per workgroup
    const void *tail;             // the < 1 segment (exact mode: < 1 tile) remainder, binned by the last workgroup
    uint32_t tail_samples;
    const uint32_t *table;        // compact LUT incl. the two sentinel cells
    papr_ccdf_params P;           // shift, cell_lo, ncells, nkeys (= edges), table_words, copies
    const papr_ccdf_params *Pdev; // null, or the same in device memory (papr_guess_bands_kernel's output; overrides P)
    unsigned long long *ghist;    // nkeys + 1 bins
    float *stash;                 // one segment of seg_cap floats per workgroup
    unsigned long long *seg_slots;// per workgroup: floats used in its stash segment (incl. padding; multiple of 4)
    unsigned long long *seg_real; // per workgroup: in-band powers stashed (the invariant: == sum of the odd bins)
    unsigned long long *gave_up;  // one counter: how often a wave gave the sweep up for its workgroup (papr_sweep.hip)
    uint64_t seg_cap;
    // exact-sum mode
    const int32_t *tile_E_spec;   // per 2048-sample tile: speculated binade of the running sum, or PAPR_EXACT_AMBIG
    void *seg_D;                  // per segment: double2 (D0, D1); D0 doubles as the segment's sum
    uint64_t seg_offset;          // index of the launch's first segment within the shard (chunked launches)
    uint32_t lds_bytes;           // papr_sweep3_kernel: the launch's dynamic LDS (set by its launch wrapper)
    uint32_t fine_table;          // papr_sweep3_kernel: more than 64 bands (the 0.1 dB table) — selects the kernel form
    uint32_t xcd_skew;            // papr_sweep3_kernel: period of the walk's XCD skew in rounds (0: none; set by papr_launch_sweep3);
                                  // bit 31 (the caller's): workgroup 0 sits on an odd XCD, the EVEN workgroups take the skew
};
int papr_sweep2_geometry(int variant, int *threads, uint64_t *seg_samples, size_t *lds_fixed, int *exact);
// ---- the exact-sum sweep, third form (papr_sweep.hip: papr_sweep3_kernel) ----------------------------------------
// Same parameter block and segment numbering as papr_sweep2_kernel<EXACT>; the band-edge table is papr_sweep_kernel's
// (ONE edge per cell, a sentinel cell at either end, nkeys + 2 bins), the stash leaves through per-wave LDS slices.
#define PAPR_SWEEP3_SLICE_FLOATS 1408u /* per wave: what the 160 KiB leave beside 8 x 8 KiB of transposition buffers and a 40 KiB table */
int papr_sweep3_geometry(int variant, int *threads, size_t *lds_fixed, int *exact = nullptr); /* 0, or -1 if `variant` is not one of its ids */
void papr_launch_sweep3(hipStream_t st, int variant, int blocks, size_t lds_bytes, const papr_sweep2_params &p);
void papr_launch_sweep2(hipStream_t st, int variant, int blocks, size_t lds_bytes, const papr_sweep2_params &p);
#ifdef PAPR_MEASURE
int papr_lab_sweep2_geometry(int variant, int *threads, uint64_t *seg_samples, size_t *lds_fixed, int *exact);
void papr_lab_launch_sweep2(hipStream_t st, int variant, int blocks, size_t lds_bytes, const papr_sweep2_params &p);
#endif
int papr_ccdf_max_dynamic_lds(void);
void papr_kernels_prepare_device(void); /* call once per device after hipSetDevice */
void papr_exact_prepare_device(void);

#endif
