// papr_sweep.hip — ONE read of the shard for both passes of papr (gfx950 / MI355X).
//
// The reference reads the file twice because pass 2's thresholds are mean * 10^(dB/10) and the mean
// is only known after pass 1 (papr.c:131-141 -> :145-152).  Both passes are HBM-bound here, so the
// second read is half of the job.  The one-sweep scheme keeps the result exact and drops it:
//
//   1. papr_estimate_kernel      sums a pseudo-random 1/ratio of the shard's 1 KiB rows (1/64 by
//                                default): a mean good to ~1e-4 relative
//   2. the host turns that guess into the level table the reference would build from it and widens
//      every threshold into a BAND of +-2^w float bit patterns (default w = 14, i.e. +-0.1 .. 0.2 %)
//   3. papr_sweep_kernel         pass 1 as papr_stats_kernel does it (same trackers; same geometry =>
//                                same sum), and in the same read every power is binned against
//                                the band edges: even bins lie BETWEEN bands, so whichever way the
//                                true threshold falls inside its band those samples are already
//                                decided; the few per cent that land INSIDE a band (odd bins) are
//                                appended to a stash of float powers in HBM
//   4. once the true mean — hence the true table — is known, papr_ccdf_power_kernel bins just the
//      stash against it.  counts_above[j] = (even bins above band j) + (stash powers > level j).
//
// If a true threshold falls outside its band, the stash overflows or the table has no LUT form, the
// runtime simply runs the classic pass 2 (papr_ccdf_kernel): speculation never changes a result, it
// only decides how many bytes are read.  Typical extra traffic: 0.5 % of a pass (default table),
// 4.6 % (-g), instead of 100 %.
//
// The stash is filled without workgroup barriers and without global atomics: every workgroup owns one
// segment of the HBM stash, every wave owns a slice of LDS in which its lanes get slots for their in-band
// powers (default kernel: by ballot + mbcnt, without a branch; older forms: with a returning LDS atomic);
// when the slice is about to run out the wave reserves a range of the workgroup's segment with one LDS
// atomic and writes it out coalesced.  (A single global counter serialises at ~11 ns per reservation across
// the 8 XCDs — measured: it doubled the kernel time of the 0.1 dB table.)
//
// The product kernels are papr_sweep_kernel (512 threads x 8 loads per lane, ONE persistent workgroup per CU, next-tile
// prefetch, branch-free per-sample code) and, in exact-sum mode, papr_sweep3_kernel (the same on wave-private segments,
// plus the sequential sum's rounding-function pairs).  Every other geometry / stash form / ablation that was measured on
// the way is built by `make MEASURE=1` only (measure/papr_sweep_lab.hip) and selected through the same variant ids.

#include "papr_sweep_dev.h"

// =============================================================================
// 1. mean estimate from a 1/ratio sample of the tiles
// =============================================================================
// Group g = tiles [g*ratio, (g+1)*ratio) of 16 KiB.  A workgroup reads one tile's worth of it, but every 1 KiB
// row (one wave-wide 16-byte load) from a tile of the group chosen by a hash of (g, row): no periodic structure
// in the capture can alias with the sampling, and a bursty capture is sampled in 16 x more independent places
// than whole tiles would give (the error of the mean is sigma(piece means) / sqrt(pieces)).
// Output: one papr_partial per workgroup with only `sum` set (merged by papr_stats_finalize like pass-1 partials).
// `group_sums` (may be null): 4 doubles per group, one per wave — their sum is the group's sampled sum (the exact
// one-read sweep speculates each tile's running-sum binade from them, papr_exact.hip).
// `block_sq` (may be null): per workgroup, the sum over its (group, wave) pieces of the piece's sum SQUARED — with
// the total that gives the scatter of the pieces, i.e. the standard error of the estimate (the host sizes the
// threshold bands from it: a bursty capture gets wider bands than a stationary one).
__global__ __launch_bounds__(PAPR_BLOCK) void papr_estimate_kernel(const float4 *__restrict__ data, uint64_t ngroups,
                                                                    uint32_t ratio, papr_partial *__restrict__ out,
                                                                    double *__restrict__ group_sums,
                                                                    double *__restrict__ block_sq)
{
    constexpr int U = PAPR_ESTIMATE_TILE_SAMPLES / (2 * PAPR_BLOCK);
    constexpr int kRows = U * (PAPR_BLOCK / kWave);
    constexpr uint64_t TILE_F4 = (uint64_t)PAPR_BLOCK * U;
    const uint32_t wave = threadIdx.x / kWave;
    double sum = 0.0, sq = 0.0;
    for (uint64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        float4 x[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t row = g * kRows + (uint64_t)u * (PAPR_BLOCK / kWave) + wave;
            const uint64_t tile = g * ratio + papr_estimate_pick(row, ratio);
            x[u] = load16<true>(data + tile * TILE_F4 + (uint64_t)u * PAPR_BLOCK + threadIdx.x);
        }
        double gsum = 0.0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            gsum += (double)power_of(x[u].x, x[u].y);
            gsum += (double)power_of(x[u].z, x[u].w);
        }
        sum += gsum;
        const double ws = wave_reduce_sum(gsum);  // this wave's piece of the group (valid in lane 0)
        sq += ws * ws;
        if (group_sums && (threadIdx.x & (kWave - 1)) == 0)
            group_sums[g * (PAPR_BLOCK / kWave) + wave] = ws;
    }
    if (block_sq) {
        __shared__ double sh_sq[PAPR_BLOCK / kWave];
        if ((threadIdx.x & (kWave - 1)) == 0)
            sh_sq[wave] = sq;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int wq = 0; wq < PAPR_BLOCK / kWave; wq++)
                tot += sh_sq[wq];
            block_sq[blockIdx.x] = tot;
        }
    }
    LaneStats s;
    s.sum = sum;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        s.val[k] = 0.f;
        s.idx[k] = 0;
    }
    block_reduce_stats<PAPR_BLOCK / kWave>(s);
    if (threadIdx.x == 0) {
        papr_partial q;
        q.sum = s.sum;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            q.idx[k] = 0;
            q.val[k] = 0.f;
        }
        q.pad = 0;
        out[blockIdx.x] = q;
    }
}

// =============================================================================
// 2. the guess, on the device
// =============================================================================
// One workgroup between the estimate kernel and the sweep kernel: mean estimate and its standard error from the
// estimate's partial records, band width (papr_sweep_band_for), guessed level table (papr_guess_levels; the GUESS
// needs no libm exactness — the bands absorb its error, and the TRUE table is still built by papr_levels on the host
// after the sweep), band edges (papr_sweep_bands) and their compact LUT (plan_compact_lut / fill_compact_lut of
// papr_sweep_rt.cpp, same layout).  Everything the host needs afterwards goes to mapped host memory.
namespace {
// sum of n doubles by a 1024-thread workgroup, in a fixed order (deterministic); every thread gets the result
__device__ __forceinline__ double strided(const double *v, uint32_t k) { return v[k]; }
__device__ __forceinline__ double strided(const papr_partial *v, uint32_t k) { return v[k].sum; }
template <typename T>
__device__ __forceinline__ double block_sum_1024(const T *v, uint32_t n, double *red)
{
    const uint32_t t = threadIdx.x;
    double a = 0.0;
    for (uint32_t b = t; b < n; b += 1024)
        a += strided(v, b);
    red[t] = a;
    __syncthreads();
    for (uint32_t w = 512; w > 0; w >>= 1) {
        if (t < w)
            red[t] += red[t + w];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}
}  // namespace

// this shard's estimate as one record, for the all-gather between the estimate kernel and papr_guess_bands_kernel
__global__ __launch_bounds__(1024) void papr_est_record_kernel(const papr_partial *__restrict__ est_partials,
                                                                const double *__restrict__ est_sq, uint32_t est_blocks,
                                                                uint64_t ngroups, uint64_t sampled, uint64_t nsamples, uint32_t ratio,
                                                                uint32_t flags, papr_est_record *__restrict__ out)
{
    __shared__ double red[1024];
    const double S = block_sum_1024(est_partials, est_blocks, red);
    const double sq = block_sum_1024(est_sq, est_blocks, red);
    if (threadIdx.x == 0) {
        papr_est_record e;
        e.S = S;
        e.sq = sq;
        e.sampled = sampled;
        e.n = nsamples;
        e.pieces = 4ull * ngroups;
        e.ratio = ratio;
        e.flags = flags;
        e.pad = 0;
        *out = e;
    }
}

void papr_launch_est_record(hipStream_t st, const papr_partial *est_partials, const double *est_sq, uint32_t est_blocks,
                            uint64_t ngroups, uint64_t sampled, uint64_t nsamples, uint32_t ratio, uint32_t flags,
                            papr_est_record *out)
{
    hipLaunchKernelGGL(papr_est_record_kernel, dim3(1), dim3(1024), 0, st, est_partials, est_sq, est_blocks, ngroups, sampled,
                       nsamples, ratio, flags, out);
}

// the shards' pass-1 records (rank = file order) folded as papr_stats_merge folds them: sums added in rank order,
// trackers by "more extreme value, else smaller index" from the reference's initial state (0.0 at index 0)
__global__ void papr_record_merge_kernel(const papr_partial *__restrict__ recs, const papr_est_record *__restrict__ est,
                                         uint32_t world, uint32_t rank, papr_partial *__restrict__ total_dev,
                                         unsigned long long *__restrict__ n_total_dev, papr_peer_out *__restrict__ out_host)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    papr_partial acc;
    acc.sum = 0.0;
    acc.pad = 0;
    for (int k = 0; k < 5; k++) {
        acc.val[k] = 0.f;
        acc.idx[k] = 0;
    }
    double before = 0.0;
    unsigned long long n = 0, nans = 0, flags = 0;
    for (uint32_t r = 0; r < world; r++) {
        const papr_partial q = recs[r];
        flags |= est[r].flags;
        if (r == rank)
            before = acc.sum;
        acc.sum = acc.sum + q.sum;
        nans += q.sum != q.sum ? 1ull : 0ull;
        n += est[r].n;
        for (int k = 0; k < 5; k++) {
            const bool is_min = k == 2 || k == 4;
            const bool more = is_min ? q.val[k] < acc.val[k] : q.val[k] > acc.val[k];
            if (more || (q.val[k] == acc.val[k] && q.idx[k] < acc.idx[k])) {
                acc.val[k] = q.val[k];
                acc.idx[k] = q.idx[k];
            }
        }
    }
    *total_dev = acc;
    n_total_dev[0] = n;
    reinterpret_cast<double *>(n_total_dev)[1] = before;  // (exact-sum mode: the classification's prefix starts here)
    out_host->total = acc;
    out_host->before = before;
    out_host->n_total = n;
    out_host->nan_ranks = nans;
    out_host->flags = flags;
}

void papr_launch_record_merge(hipStream_t st, const papr_partial *recs, const papr_est_record *est, uint32_t world, uint32_t rank,
                              papr_partial *total_dev, unsigned long long *n_total_dev, papr_peer_out *out_host)
{
    hipLaunchKernelGGL(papr_record_merge_kernel, dim3(1), dim3(64), 0, st, recs, est, world, rank, total_dev, n_total_dev, out_host);
}

// everything the ranks have to add up after the recount, as one vector: [sweep bins | recount bins | flags]
__global__ __launch_bounds__(1024) void papr_xpack_kernel(const unsigned long long *__restrict__ sweep_hist, uint32_t sweep_words,
                                                           const unsigned long long *__restrict__ seg_fill, uint32_t nsegs,
                                                           uint64_t seg_cap, const unsigned long long *__restrict__ gave_up,
                                                           const unsigned long long *__restrict__ recount_hist,
                                                           uint32_t recount_words, const papr_guess_out *__restrict__ guess,
                                                           const papr_true_out *__restrict__ tru,
                                                           unsigned long long *__restrict__ vec)
{
    __shared__ uint32_t s_over;
    const uint32_t t = threadIdx.x;
    if (t == 0)
        s_over = 0;
    __syncthreads();
    for (uint32_t k = t; k < sweep_words; k += 1024)
        vec[k] = sweep_hist[k];
    for (uint32_t k = t; k < recount_words; k += 1024)
        vec[sweep_words + k] = recount_hist[k];
    uint32_t over = 0;
    for (uint32_t b = t; b < nsegs; b += 1024)
        over |= seg_fill[b] > seg_cap ? 1u : 0u;
    if (over)
        atomicOr(&s_over, 1u);
    __syncthreads();
    if (t == 0) {
        unsigned long long *f = vec + sweep_words + recount_words;
        f[0] = (s_over || *gave_up != 0) ? 1ull : 0ull;
        f[1] = guess->ok ? 0ull : 1ull;
        f[2] = tru->ok ? 0ull : 1ull;
        f[3] = 0ull;
    }
}

void papr_launch_xpack(hipStream_t st, const unsigned long long *sweep_hist, uint32_t sweep_words, const unsigned long long *seg_fill,
                       uint32_t nsegs, uint64_t seg_cap, const unsigned long long *gave_up, const unsigned long long *recount_hist,
                       uint32_t recount_words, const papr_guess_out *guess, const papr_true_out *tru, unsigned long long *vec)
{
    hipLaunchKernelGGL(papr_xpack_kernel, dim3(1), dim3(1024), 0, st, sweep_hist, sweep_words, seg_fill, nsegs, seg_cap, gave_up,
                       recount_hist, recount_words, guess, tru, vec);
}

__global__ __launch_bounds__(1024) void papr_guess_bands_kernel(
    const papr_partial *__restrict__ est_partials, const double *__restrict__ est_sq, uint32_t est_blocks, uint64_t ngroups,
    uint64_t sampled, uint64_t nsamples, uint32_t ratio, int graph, float max_db, float spoil, int band_override,
    uint32_t copies, int compact, uint32_t soft_lds, uint32_t *__restrict__ table, uint32_t table_cap_words,
    papr_guess_out *__restrict__ out_dev, papr_guess_out *__restrict__ out_host, unsigned long long *__restrict__ zero,
    uint32_t zero_words, const papr_est_record *__restrict__ recs, uint32_t nrecs, uint32_t my_rank,
    const double *__restrict__ spec_group_sums, uint64_t spec_ngroups, double spec_scale, double *__restrict__ spec_group_prefix,
    const double *__restrict__ pow_tab)
{
    if (blockIdx.x == 1) {
        // exact-sum mode without peers: the scan of the estimate's per-group sums (first half of the binade speculation,
        // papr_exact.hip) has nothing to do with the guess — it runs beside it instead of behind it
        __shared__ double sh_scan[1024 / kWave];
        papr_exact_spec_scan_body(spec_group_sums, spec_ngroups, spec_scale, 0.0, spec_group_prefix, sh_scan);
        return;
    }
    constexpr uint32_t kNeverHi = 0xFFFFFFFFu;
    for (uint32_t w = threadIdx.x; w < zero_words; w += 1024)
        zero[w] = 0;  // (the sweep's histogram and segment counters: saves a memset between the launches)
    __shared__ double red[1024];
    __shared__ uint32_t keys[PAPR_GUESS_MAX_BANDS];
    __shared__ uint32_t edges[2 * PAPR_GUESS_MAX_BANDS];
    __shared__ uint32_t s_m, s_bad;
    const uint32_t t = threadIdx.x;
    double mean, rel, est_sum, est_before = 0.0;
    if (recs) {
        // ---- peers: every shard's record (rank = file order), the same on every rank: the FILE's mean, the largest of
        // the shards' relative standard errors (papr_stats_merge keeps the maximum), what lies in front of this shard ----
        double tot = 0.0, ntot = 0.0, relmax = 0.0, mine = 0.0;
        for (uint32_t r = 0; r < nrecs; r++) {
            const papr_est_record e = recs[r];
            const double scaled = e.sampled ? e.S * ((double)e.n / (double)e.sampled) : 0.0;
            const double pieces = (double)e.pieces;
            const double var_total = pieces > 1.0 ? pieces / (pieces - 1.0) * fmax(0.0, e.sq - e.S * e.S / pieces) : 0.0;
            const double rel_r = (e.S > 0.0 && e.ratio > 1) ? (double)(float)(sqrt(var_total) / e.S) : 0.0;  // (the host path carries it as a float)
            if (r < my_rank)
                est_before += scaled;
            if (r == my_rank)
                mine = scaled;
            tot += scaled;
            ntot += (double)e.n;
            relmax = rel_r > relmax || !(rel_r == rel_r) ? rel_r : relmax;
        }
        mean = ntot > 0.0 ? tot / ntot : 0.0;
        rel = relmax;
        est_sum = mine;
    } else {
        // ---- the estimate: sum and sum of squared piece sums (fixed order: deterministic) ----
        const double S = block_sum_1024(est_partials, est_blocks, red);
        const double sq = block_sum_1024(est_sq, est_blocks, red);
        const double pieces = 4.0 * (double)ngroups;
        const double var_total = pieces > 1.0 ? pieces / (pieces - 1.0) * fmax(0.0, sq - S * S / pieces) : 0.0;
        rel = (S > 0.0 && ratio > 1) ? sqrt(var_total) / S : 0.0;
        mean = sampled ? S / (double)sampled : 0.0;
        est_sum = sampled ? S * ((double)nsamples / (double)sampled) : 0.0;
    }
    // ---- band half-width: 4.5 standard errors, 2^10 .. 2^20 (papr_sweep_band_for) ----
    int band = 10;
    {
        const double want = 4.5 * rel * 16777216.0;
        while (band < 20 && (double)(1u << band) < want)
            band++;
        if (!(rel >= 0.0) || !(rel < 1.0))
            band = 14;
        if (band_override > 0)
            band = band_override;
        else if (!compact && band < 14)
            band = 14;  // (the one-edge table's cells are as narrow as the bands: below 2^14 a 20-octave table does not fit)
    }
    // ---- guessed thresholds and their keys ----
    uint32_t nl = graph ? (uint32_t)(max_db * 10.0f) + 1u : (uint32_t)max_db + 1u;
    nl = nl > PAPR_GUESS_MAX_BANDS ? PAPR_GUESS_MAX_BANDS : nl;
    if (t == 0)
        s_m = (mean > 0.0 && mean <= 3e38) ? nl : 0u;
    __syncthreads();
    const bool have = s_m != 0;
    __syncthreads();
    uint32_t key = kNeverHi;
    if (t < nl && have) {
        // (the host libm's pow(10, x_t), computed once per process, where the runtime has uploaded it: no pow() here)
        const float db = graph ? (float)t * 0.1f : (float)t;
        const double p10 = pow_tab && t < PAPR_POW_TABLE ? pow_tab[(graph ? PAPR_POW_TABLE : 0) + t] : pow(10.0, (double)(db / 10.0f));
        const float lv = (float)(p10 * mean) * spoil;
        const uint32_t bits = __float_as_uint(lv);
        key = (lv != lv || bits >= 0x7F800000u) ? kNeverHi : (lv <= 0.0f ? (lv < 0.0f ? 0u : 1u) : bits + 1u);
        keys[t] = key;
    }
    __syncthreads();
    if (t < nl && have && (key == kNeverHi || (t > 0 && key <= keys[t - 1])))
        atomicMin(&s_m, t);  // the table ends in front of the first key that is not a finite step up
    __syncthreads();
    const uint32_t m = s_m;
    // ---- widest band (down to three steps narrower) whose bands are normal floats and do not touch ----
    uint32_t half = 0;
    for (int w = band; m && w >= (band - 3 > 8 ? band - 3 : 8) && !half; w--) {
        if (t == 0)
            s_bad = 0;
        __syncthreads();
        const uint32_t h = 1u << w;
        if (t < m) {
            const uint32_t g = keys[t];
            if (g < 0x00800000u + h || g >= 0x7F800000u - h || (t && g - h <= keys[t - 1] + h))
                s_bad = 1;
        }
        __syncthreads();
        if (!s_bad) {
            half = h;
            band = w;
        }
        __syncthreads();
    }
    const uint32_t n = half ? 2 * m : 0;  // edges
    if (t < m && half) {
        edges[2 * t] = keys[t] - half;
        edges[2 * t + 1] = keys[t] + half;
    }
    __syncthreads();
    // ---- the LUT: the coarsest cell that leaves at most two edges (compact form) / one edge (papr_sweep_kernel's
    // plain form: finish_plan) in any cell ----
    // (two patterns lie in different cells of size 2^s exactly when their highest differing bit is >= s: the coarsest
    // admissible cell is the minimum of that bit over all pairs that must be apart — one reduction, no search)
    int shift = -1;
    const uint32_t reach = compact ? 2u : 1u;  // edges[i + reach] must not share a cell with edges[i]
    if (t == 0)
        s_bad = 31;
    __syncthreads();
    for (uint32_t i = t; i + reach < n; i += 1024)
        atomicMin(&s_bad, 31u - (uint32_t)__clz((int)(edges[i + reach] ^ edges[i])));
    __syncthreads();
    if (n) {
        const int top = compact ? PAPR_LUT2_MAX_SHIFT : 23;
        shift = (int)s_bad < top ? (int)s_bad : top;
        if (shift < 8)
            shift = -1;
    }
    __syncthreads();
    papr_ccdf_params P;
    P.shift = 8;
    P.cell_lo = 1;
    P.ncells = 0;
    P.nkeys = 0;
    P.above_lo = P.above_count = 0;
    P.table_words = 4;
    P.copies = copies;
    P.search_step = 0;
    uint32_t ok = 0;
    if (shift >= 0 && n <= PAPR_LUT2_MAX_EDGES) {
        const uint32_t c0 = edges[0] >> shift, c1 = edges[n - 1] >> shift;
        const uint64_t ncells = (uint64_t)c1 - c0 + 1;
        const uint64_t words = compact ? ((2 * (ncells + 2) + 3) & ~3ull) : 2 * (ncells + 2);
        if ((compact ? (ncells + 2) * 8 <= 48 * 1024 : ncells * 8 <= 40 * 1024) && words <= table_cap_words) {
            P.shift = (uint32_t)shift;
            P.cell_lo = c0;
            P.ncells = (uint32_t)ncells;
            P.nkeys = n;
            P.table_words = (uint32_t)words;
            // histogram copies: fewer for big tables (as finish_plan: the workgroup's share of the LDS)
            while (P.copies > 1 && (size_t)P.table_words * 4 + (size_t)P.copies * (n + 2) * 4 > soft_lds)
                P.copies--;
            ok = 1;
        }
    }
    // ---- the table: one entry per cell + a sentinel at either end ----
    //   compact (fill_compact_lut): { edges below << 22 | offset of the 1st edge inside, offset of the 2nd }
    //   plain (sweep_prepare):      { edges below, the edge inside or never }
    for (uint32_t w = t; w < P.table_words; w += 1024)
        table[w] = 0;
    __syncthreads();
    if (t == 0) {
        table[0] = compact ? PAPR_LUT2_NEVER : 0u;  // below everything: no edge below, none inside
        table[1] = kNeverHi;
        table[2 * (P.ncells + 1)] = compact ? ((P.nkeys << PAPR_LUT2_OFF_BITS) | PAPR_LUT2_NEVER) : P.nkeys;  // above every edge
        table[2 * (P.ncells + 1) + 1] = compact ? kNeverHi : 0x7F800001u;  // (plain: NaN patterns land in the trash bin)
    }
    if (ok) {
        const uint32_t mask = (1u << P.shift) - 1u;
        for (uint32_t c = t; c < P.ncells; c += 1024) {
            const uint32_t cell = P.cell_lo + c;
            uint32_t lo = 0, hi = n;  // first edge whose cell is >= `cell`
            while (lo < hi) {
                const uint32_t mid = (lo + hi) / 2;
                if ((edges[mid] >> P.shift) < cell)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            const bool in1 = lo < n && (edges[lo] >> P.shift) == cell;
            const bool in2 = lo + 1 < n && (edges[lo + 1] >> P.shift) == cell;
            if (compact) {
                table[2 * (c + 1)] = (lo << PAPR_LUT2_OFF_BITS) | (in1 ? (edges[lo] & mask) : PAPR_LUT2_NEVER);
                table[2 * (c + 1) + 1] = in2 ? (edges[lo + 1] & mask) : kNeverHi;
            } else {
                table[2 * (c + 1)] = lo;
                table[2 * (c + 1) + 1] = in1 ? edges[lo] : kNeverHi;
            }
        }
    }
    // ---- what the sweep kernel and the host need ----
    if (t == 0) {
        papr_guess_out *outs[2] = {out_dev, out_host};
        for (int k = 0; k < 2; k++) {
            papr_guess_out *o = outs[k];
            o->P = P;
            o->ok = ok;
            o->band_log2 = (uint32_t)band;
            o->nbands = ok ? m : 0u;
            o->pad = 0;
            o->est_sum = est_sum;
            o->est_rel_se = rel;
            o->est_before = est_before;
        }
    }
    for (uint32_t j = t; j < m && ok; j += 1024) {
        out_dev->gkeys[j] = keys[j];
        out_host->gkeys[j] = keys[j];
    }
}

void papr_launch_guess_bands(hipStream_t st, const papr_partial *est_partials, const double *est_sq, uint32_t est_blocks,
                             uint64_t ngroups, uint64_t sampled, uint64_t nsamples, uint32_t ratio, int graph, float max_db,
                             float spoil, int band_override, uint32_t copies, int compact, uint32_t soft_lds, uint32_t *table,
                             uint32_t table_cap_words, papr_guess_out *out_dev, papr_guess_out *out_host,
                             unsigned long long *zero, uint32_t zero_words, const papr_est_record *recs, uint32_t nrecs,
                             uint32_t my_rank, const double *spec_group_sums, uint64_t spec_ngroups, double spec_scale,
                             double *spec_group_prefix, const double *pow_tab)
{
    const bool with_scan = spec_group_sums && spec_group_prefix && spec_ngroups;
    hipLaunchKernelGGL(papr_guess_bands_kernel, dim3(with_scan ? 2 : 1), dim3(1024), 0, st, est_partials, est_sq, est_blocks, ngroups,
                       sampled, nsamples, ratio, graph, max_db, spoil, band_override, copies, compact, soft_lds, table,
                       table_cap_words, out_dev, out_host, zero, zero_words, recs, nrecs, my_rank, spec_group_sums, spec_ngroups,
                       spec_scale, spec_group_prefix, pow_tab);
}

// =============================================================================
// 2b. the true table, speculated on the device
// =============================================================================
// After the sweep the host builds the reference's level table from the pass-1 record with libm (papr_levels) — that
// stays — but waiting for it before the stash recount can start costs a launch + wait round trip.  So one workgroup
// builds the same table right behind the finalize kernel with the device's pow / log10, plans the recount's LUT for it
// (finish_plan's LUT form) and the recount runs on that table at once; the host then compares its own table with this
// one bit for bit (a double-precision pow differs from libm's in the last place once in a few million levels) and uses
// the recount's histogram if they agree, or recounts as before if they do not.  Speculation decides how long the step
// takes, never a count.
__global__ __launch_bounds__(1024) void papr_true_table_kernel(const papr_partial *__restrict__ result, uint64_t nsamples, int graph,
                                                                uint32_t copies, uint32_t soft_lds, uint32_t *__restrict__ table,
                                                                uint32_t table_cap_words, papr_true_out *__restrict__ out_dev,
                                                                papr_true_out *__restrict__ out_host,
                                                                unsigned long long *__restrict__ zero, uint32_t zero_words,
                                                                const unsigned long long *__restrict__ gave_up,
                                                                const unsigned long long *__restrict__ nsamples_dev,
                                                                const double *__restrict__ pow_tab)
{
    if (nsamples_dev)
        nsamples = *nsamples_dev;  // (peers: the file's length, known once the shards' records have been gathered)
    for (uint32_t w = threadIdx.x; w < zero_words; w += 1024)
        zero[w] = 0;  // (the recount's histogram)
    const bool hopeless = *gave_up != 0;  // the sweep gave itself up: its stash is void, nothing to recount
    __shared__ uint32_t keys[PAPR_TRUE_MAX_LEVELS];
    __shared__ uint32_t s_bad;
    const uint32_t t = threadIdx.x;
    const double sum = result->sum;
    const float peak = result->val[0];
    // papr.c:131 / 164, :134 / 165, :136 / 166 (papr_levels)
    const double mean = sum / (double)(long long)nsamples;
    const float papr = (float)(10 * log10((double)peak / mean));
    const float scaled = graph ? papr * 10 : papr;
    const int top = (!(scaled == scaled) || scaled >= 2147483648.0f || scaled < -2147483648.0f) ? INT32_MIN : (int)scaled;
    const uint32_t nl = top < 0 ? 0u : (uint32_t)top + 1u;
    bool ok = nl >= 1 && nl <= PAPR_TRUE_MAX_LEVELS && sum == sum && mean > 0.0 && !hopeless;
    if (t == 0)
        s_bad = 0;
    __syncthreads();
    float level = 0.f;
    uint32_t key = 0;
    if (ok && t < nl) {
        if (pow_tab && t < PAPR_POW_TABLE) {
            // the host libm's own pow(10, x_t) (papr_host.c, uploaded once): this level IS papr_levels' level
            level = (float)(pow_tab[(graph ? PAPR_POW_TABLE : 0) + t] * mean);
        } else if (graph) {
            float tenth_db = 0.0f;  // papr.c:168-173: the float accumulation, step by step
            for (uint32_t j = 0; j < t; j++)
                tenth_db = (float)(tenth_db + 0.1);
            level = (float)(pow(10.0, (double)(tenth_db / 10)) * mean);
        } else {
            level = (float)(pow(10.0, (double)((float)t / 10)) * mean);  // papr.c:138-141
        }
        const uint32_t bits = __float_as_uint(level);
        // the recount's LUT form wants normal, finite, positive levels (anything else: the host does it)
        if (!(level > 0.0f) || bits < 0x00800000u || bits >= 0x7F7FFFFFu)
            s_bad = 1;
        key = bits + 1u;
        keys[t] = key;
    }
    __syncthreads();
    if (ok && t > 0 && t < nl && key <= keys[t - 1])
        s_bad = 1;  // (not strictly increasing: duplicates are the host's business)
    __syncthreads();
    ok = ok && !s_bad;
    const uint32_t m = ok ? nl : 0u;
    // ---- LUT: the coarsest cell that isolates every key (finish_plan) ----
    int shift = -1;
    if (t == 0)
        s_bad = 31;
    __syncthreads();
    for (uint32_t k = t + 1; k < m; k += 1024)
        atomicMin(&s_bad, 31u - (uint32_t)__clz((int)(keys[k] ^ keys[k - 1])));  // highest bit in which neighbours differ
    __syncthreads();
    if (m) {
        shift = (int)s_bad < 23 ? (int)s_bad : 23;
        const uint64_t ncells = (uint64_t)(keys[m - 1] >> shift) - (keys[0] >> shift) + 1;
        if (shift < 8 || ncells * 8 > 40 * 1024)
            shift = -1;
    }
    __syncthreads();
    papr_ccdf_params P;
    P.shift = 8;
    P.cell_lo = 1;
    P.ncells = 0;
    P.nkeys = 0;
    P.above_lo = 0x7F800001u;
    P.above_count = 0;
    P.table_words = 0;
    P.copies = copies;
    P.search_step = 0;
    uint32_t good = 0;
    if (shift >= 0) {
        const uint32_t c0 = keys[0] >> shift, c1 = keys[m - 1] >> shift;
        const uint32_t ncells = c1 - c0 + 1;
        if (2 * ncells <= table_cap_words) {
            P.shift = (uint32_t)shift;
            P.cell_lo = c0;
            P.ncells = ncells;
            P.nkeys = m;
            const uint64_t above = ((uint64_t)c1 + 1) << shift;
            P.above_lo = above <= 0x7F800000u ? (uint32_t)above : 0x7F800001u;
            P.above_count = above <= 0x7F800000u ? 0x7F800001u - P.above_lo : 0u;
            P.table_words = 2 * ncells;
            while (P.copies > 1 && (size_t)P.table_words * 4 + (size_t)P.copies * (m + 1) * 4 > soft_lds)
                P.copies--;
            // (soft_lds is also what the recount was launched with: a table that needs more is the host's business)
            good = (size_t)P.table_words * 4 + (size_t)P.copies * (m + 1) * 4 <= soft_lds ? 1u : 0u;
            // lut[cell] = { keys strictly below this cell, the key inside this cell or never } (upload_ccdf_table)
            for (uint32_t c = t; c < ncells; c += 1024) {
                const uint32_t cell = c0 + c;
                uint32_t lo = 0, hi = m;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) / 2;
                    if ((keys[mid] >> shift) < cell)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                table[2 * c] = lo;
                table[2 * c + 1] = (lo < m && (keys[lo] >> shift) == cell) ? keys[lo] : 0xFFFFFFFFu;
            }
            if (!good) {  // (the recount then runs on an empty table: harmless, unused)
                P.nkeys = 0;
                P.ncells = 0;
                P.table_words = 0;
                P.above_lo = 0x7F800001u;
                P.above_count = 0;
            }
        }
    }
    if (t == 0) {
        papr_true_out *outs[2] = {out_dev, out_host};
        for (int k = 0; k < 2; k++) {
            outs[k]->P = P;
            outs[k]->ok = good;
            outs[k]->nlevels = good ? nl : 0u;
            outs[k]->pad[0] = outs[k]->pad[1] = 0;
        }
    }
    if (good && t < nl)
        out_host->levels[t] = level;
}

void papr_launch_true_table(hipStream_t st, const papr_partial *result, uint64_t nsamples, int graph, uint32_t copies,
                            uint32_t soft_lds, uint32_t *table, uint32_t table_cap_words, papr_true_out *out_dev,
                            papr_true_out *out_host, unsigned long long *zero, uint32_t zero_words,
                            const unsigned long long *gave_up, const unsigned long long *nsamples_dev, const double *pow_tab)
{
    hipLaunchKernelGGL(papr_true_table_kernel, dim3(1), dim3(1024), 0, st, result, nsamples, graph, copies, soft_lds, table,
                       table_cap_words, out_dev, out_host, zero, zero_words, gave_up, nsamples_dev, pow_tab);
}

// =============================================================================
// 3. the sweep: pass 1 + band binning + stash, one read
// =============================================================================
// Doing both passes' arithmetic per sample costs more VALU work than either pass alone (a CU has 64
// lane-ops per clock, i.e. ~44 per sample at full HBM speed), so this kernel trims both halves:
//
//  * trackers per TILE, not per sample: the lane folds the 2U values of a tile with integer max3 on
//    the float bit patterns (signed max finds the largest positive float, unsigned max the most
//    negative one; powers are >= +0), then does ONE strict float compare per tracker per tile and
//    remembers the iteration.  After the loop the lane re-reads that one tile and takes the first
//    slot that holds the value: the same first-occurrence answer as papr_stats_kernel.  NaN bit
//    patterns would win an integer max, but any NaN in I or Q also makes the sum NaN, and then the
//    runtime discards this launch's pass-1 record and runs papr_stats_kernel instead.
//  * the double sum is accumulated exactly as papr_stats_kernel does it (per lane in tile order, then a fixed
//    wave / workgroup / grid tree): the same value whenever the two kernels run the same geometry
//  * branch-free binning: the LUT carries a "below" sentinel cell in front and an "above" one behind,
//    the cell index is clamped with one med3.  The above sentinel sends NaN powers to a trash bin
//    (index nkeys + 1, odd: they also go to the stash, where the recount ignores them).
//
// LDS: [LUT of the band edges | histogram copies | one stash slice per wave].
// `table`/P describe the 2m band edges lo_0 < hi_0 < lo_1 < ... so bin k = #{edges <= bits(power)}:
// k odd <=> inside band (k-1)/2.

// The product form: 512-thread workgroups x 8 loads per lane (64 KiB tiles), launched as ONE persistent workgroup per CU
// over grid-stride tiles — eight waves per CU with eight 16-byte loads each in flight is the shape a stripped kernel
// reads fastest in — with the next tile's loads issued before the current one is folded, and no branch or exec-masked
// region in the per-sample code (with two waves per SIMD nothing would hide them).  Other geometries, stash forms and
// the ablations that led here are built by `make MEASURE=1` only (measure/papr_sweep_lab.hip).
__global__ __launch_bounds__(PAPR_SWEEP_THREADS) void papr_sweep_kernel(const float4 *__restrict__ data, uint64_t ntiles,
                                                                        uint64_t base_index, int map,
                                                                        papr_partial *__restrict__ out,
                                                                        const float2 *__restrict__ tail, uint32_t tail_samples,
                                                                        const uint32_t *__restrict__ table, papr_ccdf_params Parg,
                                                                        unsigned long long *__restrict__ ghist,
                                                                        float *__restrict__ stash,
                                                                        unsigned long long *__restrict__ seg_counts,
                                                                        uint64_t seg_cap, unsigned long long *__restrict__ gave_up,
                                                                        unsigned long long *__restrict__ seg_real,
                                                                        const papr_ccdf_params *__restrict__ Pdev)
{
    constexpr int BLOCK = PAPR_SWEEP_THREADS, U = PAPR_SWEEP_LOADS;
    // the table's geometry: an argument, or — when papr_guess_bands_kernel built the table just before this launch,
    // without the host in between — read from where that kernel left it (wave-uniform loads: scalar registers)
    const papr_ccdf_params P = uniform_params(Pdev, Parg);
    constexpr uint64_t TILE_F4 = (uint64_t)BLOCK * U;
    constexpr uint32_t SLICE = PAPR_SWEEP_SLICE_FLOATS;
    __shared__ unsigned long long seg_fill, seg_real_sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t nbins = P.nkeys + 2;  // + the NaN trash bin
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    uint32_t *hist = tab + P.table_words;
    // (16-byte aligned: the table and the slices are read 16 bytes at a time; the launch reserves the 12 bytes)
    float *slices = reinterpret_cast<float *>(hist + ((P.copies * nbins + 3u) & ~3u));

    const uint32_t t = threadIdx.x;
    for (uint32_t k = t; k < P.table_words; k += BLOCK)
        tab[k] = table[k];
    for (uint32_t k = t; k < P.copies * nbins; k += BLOCK)
        hist[k] = 0;
    if (t == 0) {
        seg_fill = seg_counts[blockIdx.x];  // segments keep filling over the launches of a chunked ingest
        seg_real_sh = seg_real[blockIdx.x];
    }
    __syncthreads();

    const uint2 *lut_biased = reinterpret_cast<const uint2 *>(tab) - ((int32_t)P.cell_lo - 1);
    uint32_t *my = hist + ((t / kWave) % P.copies) * nbins;
    WaveStash ws;
    ws.init(slices + (t / kWave) * SLICE, SLICE, stash + (uint64_t)blockIdx.x * seg_cap, &seg_fill, &seg_real_sh, seg_cap, tab,
            P.table_words, gave_up);
    // cell index straight from the bit pattern: lut_biased[cell] with cell clamped to [cell_lo - 1, cell_lo + ncells]
    const int32_t cell_last = (int32_t)(P.cell_lo + P.ncells);
    int32_t cell_first;  // pinned in a VGPR for the whole kernel (v_med3 takes one scalar operand)
    asm volatile("v_mov_b32 %0, %1" : "=v"(cell_first) : "s"((int32_t)P.cell_lo - 1));
    const uint32_t shift = P.shift;
    auto bin_of = [&](float pw) -> uint32_t {
        const int32_t cell = __float_as_int(pw) >> shift;   // arithmetic shift: sign-bit patterns go below
        const uint2 e = lut_biased[clamp_cell(cell, cell_first, cell_last)];
        return e.x + (__float_as_uint(pw) >= e.y ? 1u : 0u);
    };
    auto count = [&](uint32_t k) {
        // bin 0 (below every band: not counted) adds to this lane's trash word instead of being skipped
        const unsigned long long nz = __ballot(k != 0u);
        const uint32_t a_bin = (uint32_t)(uintptr_t)(lds_u32 *)&my[k];
        const uint32_t a_trash = (uint32_t)(uintptr_t)(lds_u32 *)&ws.buf[ws.trash];
        uint32_t a;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a) : "v"(a_trash), "v"(a_bin), "s"(nz));
        (void)__hip_atomic_fetch_add((lds_u32 *)(uintptr_t)a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto count_and_stash = [&](float pw, uint32_t k) {  // (one sample at a time: the shard's remainder)
        count(k);
        ws.put(pw, (k & 1u) != 0u);
    };

    double sum = 0.0;
    TileTrack tr = {{0.f, 0.f, 0.f, 0.f, 0.f}, {0, 0, 0, 0, 0}};
    auto fold = [&](const float4(&x)[U], uint32_t tile, uint32_t it) {
        const uint64_t now = __builtin_amdgcn_s_memrealtime();  // (for the spill check at the end)
        float pw[2 * U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            // power_of, with the addition written out: hipcc multiplies (I, Q) as a pair (v_pk_mul_f32: the two squares
            // are where the sum needs them) but then also pairs the ADDITIONS of two samples, which costs three
            // v_mov per float4 to line the operands up — as many instructions as it saves
            typedef float f32x2v __attribute__((ext_vector_type(2)));
            const f32x2v a = {x[u].x, x[u].y}, b = {x[u].z, x[u].w};
            const f32x2v aa = a * a, bb = b * b;  // (two IEEE multiplications each; nothing to contract: -ffp-contract=off)
            asm("v_add_f32 %0, %1, %2" : "=v"(pw[2 * u]) : "v"(aa.x), "v"(aa.y));
            asm("v_add_f32 %0, %1, %2" : "=v"(pw[2 * u + 1]) : "v"(bb.x), "v"(bb.y));
        }
#pragma unroll
        for (int u = 0; u < 2 * U; u++)
            sum += (double)pw[u];  // same order as papr_stats_kernel
        track_tile<U>(tr, x, pw, tile);  // (the tracker remembers the TILE: a lane meets its tiles in increasing order)
        uint32_t k[2 * U];
#pragma unroll
        for (int u = 0; u < 2 * U; u++)
            k[u] = bin_of(pw[u]);  // all LUT reads of the tile in flight together
        // count every sample; the tile's in-band powers then go to the stash together (WaveStash::put_tile: slots per lane)
        uint32_t flag[2 * U], cnt = 0;
#pragma unroll
        for (int u = 0; u < 2 * U; u++) {
            count(k[u]);
            flag[u] = k[u] & 1u;
            cnt += flag[u];
        }
        const uint32_t folded = (it + 1) * (uint32_t)(2 * TILE_F4);
        ws.spill_in_step(now, SLICE - kWave, folded);  // (on the chip-wide ticks; put_tile makes the room the tile needs)
        ws.put_tile(pw, flag, cnt, SLICE, folded);
    };

    // which tiles: grid stride with the XCD skew (SkewWalk, papr_sweep_dev.h); `map` carries the skew's period in its upper bits
    SkewWalk walk;
    walk.init(blockIdx.x, gridDim.x, ((uint32_t)map >> 8) & 0xFFFFu, (((uint32_t)map >> 30) & 1u) ^ 1u);
    float4 cur[U], nxt[U];
    uint64_t tile = walk.tile();
    if (tile < ntiles)
        load_tile<BLOCK, U, true>(cur, data + tile * TILE_F4 + t);
    for (uint32_t it = 0; tile < ntiles; it++) {
        walk.advance();
        const uint64_t ntile = walk.tile();
        if (ntile < ntiles)
            load_tile<BLOCK, U, true>(nxt, data + ntile * TILE_F4 + t);  // the next tile's loads fly while this one is folded
        fold(cur, (uint32_t)tile, it);
#pragma unroll
        for (int u = 0; u < U; u++)
            cur[u] = nxt[u];
        tile = ntile;
    }
    const TileWalk w = {0, 1, 0};  // (sweep_record: tile = first + entry * stride — the trackers hold the tile itself)
    // sub-tile remainder of the shard: binned here (its pass-1 part is folded in by papr_stats_finalize)
    if (blockIdx.x == gridDim.x - 1) {
        for (uint32_t k0 = 0; k0 < tail_samples; k0 += BLOCK) {  // wave-uniform trip count
            const bool valid = k0 + t < tail_samples;
            const float2 x = valid ? tail[k0 + t] : make_float2(0.f, 0.f);
            const float pw = power_of(x.x, x.y);
            count_and_stash(pw, valid ? bin_of(pw) : 0u);
            ws.spill_if_above(SLICE - 2 * kWave, ~0u);
        }
    }
    ws.spill_if_above(0, ~0u);

    sweep_record<BLOCK, BLOCK, U>(sum, tr, w, data, base_index, t, out);
    hist_flush<BLOCK>(hist, nbins, P.copies, ghist);  // (starts with a barrier: every wave has spilled)
    if (t == 0) {
        seg_counts[blockIdx.x] = seg_fill;
        seg_real[blockIdx.x] = seg_real_sh;
    }
}

// =============================================================================
// 3b. the exact-sum sweep: the same per-sample code on wave-private segments, plus the pairs of the sequential sum
// =============================================================================
// In exact-sum mode (papr_hip_set_exact; bin/papr's default) the same single read also produces, per 1024-sample
// segment, the rounding-function pair (D0, D1) of papr_exact.hip for a SPECULATED binade of the running sum (from the
// estimate's per-group sums); D0 — exact or not — is the segment's sum, from which the true prefix sums are formed
// afterwards, and the few tiles whose speculated binade turns out wrong are redone (papr_exact_seg_kernel<list>), so
// that the bit-exact sequential sum costs one read as well.
//   * a wave owns whole segments (wave w of workgroup b: segment (it * gridDim + b) * 8 + w); its 64 lanes own 16
//     CONSECUTIVE samples each — the order the reference adds them in — through an XOR-swizzled LDS transposition
//     (conflict-free both ways), and the next segment's loads are issued as soon as the buffer is free again: they fly
//     while this segment is folded, in front of any spill store.  The segment goes from memory straight into that
//     buffer (gfx950's LDS-direct loads, the swizzle in the source address)
//   * the whole segment is ONE batch: its eight LDS reads, then its sixteen table lookups, are in flight together and
//     the double-precision chains run under them (with two waves per SIMD it is the LDS round trips per segment that
//     decide how long a wave is stalled)
//   * geometry and per-sample code are papr_sweep_kernel's: one persistent workgroup of eight waves per CU, the
//     one-edge-per-cell table, histogram and stash without a branch or an exec-masked region, 16-byte write-through
//     spills out of a per-wave slice — which here takes whatever the table leaves of the CU's LDS
//   * the segment's pair without an ordered tree (segment_pair, papr_sweep_dev.h).  A lane's run maps the parity of the
//     running sum onto itself: pi(0) = lsb of x0, pi(1) = lsb of x1 (the sums it formed from the even and the odd
//     canonical entry state), and its increment is d0 or d1 = d0 + delta * ulp, delta in {-1, 0, +1}.  The parity with
//     which the running sum ENTERS lane l, for segment entry parity p, follows from the 64 one-bit maps alone — ballots
//     and a dozen 64-bit scalar operations: a prefix XOR over the lanes that swap, restarted behind every lane whose map
//     is constant — and then D[p] = sum(d0) + ulp * (#{delta = +1, entered odd} - #{delta = -1, entered odd}): ONE
//     unordered wave sum (every term is a multiple of the ulp: exact in any order) and four popcounts.
// Its predecessor papr_sweep2_kernel<EXACT> (12 waves, compact table, ring stash, ordered 64-lane composition: 37 VALU +
// 18 SALU per sample against 26 + 12 here) and the forms of this kernel that lost (two batches per segment, the powers
// instead of the samples through LDS, 12 waves) are measure/papr_sweep_lab.hip's and DESIGN.md section 5's.
// FINE: the 0.1 dB table's form (a kernel of its own: as one wave-uniform branch the second path costs the first 0.05 ms)
template <bool FINE>
__global__ __launch_bounds__(PAPR_SWEEP_THREADS) void papr_sweep3_kernel(const papr_sweep2_params p)
{
    constexpr int U = 8, WAVES = PAPR_SWEEP_THREADS / kWave;
    constexpr int BLOCK = PAPR_SWEEP_THREADS;
    constexpr uint64_t SEG_F4 = 64ull * U;
    __shared__ unsigned long long seg_fill, seg_real_sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const papr_ccdf_params P = uniform_params(p.Pdev, p.P);
    const uint32_t nbins = P.nkeys + 2;  // + the NaN trash bin (as papr_sweep_kernel)
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    uint32_t *hist = tab + P.table_words;
    float *slices = reinterpret_cast<float *>(hist + ((P.copies * nbins + 3u) & ~3u));
    // the stash slices take what table and histogram leave of the launch's LDS (p.lds_bytes): a small table (the 1 dB
    // one) means rare, wide spills; at least PAPR_SWEEP3_SLICE_FLOATS (what the geometry function promises), at most 4096
    const uint32_t used_words = (uint32_t)(slices - reinterpret_cast<float *>(smem));
    constexpr uint32_t kXposeWords = WAVES * 2048u;
    const uint32_t free_words = p.lds_bytes / 4u > used_words + kXposeWords ? p.lds_bytes / 4u - used_words - kXposeWords : 0u;
    uint32_t slice_words = (free_words / WAVES) & ~63u;
    slice_words = slice_words < PAPR_SWEEP3_SLICE_FLOATS ? PAPR_SWEEP3_SLICE_FLOATS : (slice_words > 4096u ? 4096u : slice_words);
    const uint32_t SLICE = __builtin_amdgcn_readfirstlane(slice_words);
    float4 *xpose = reinterpret_cast<float4 *>(slices + WAVES * SLICE);  // WAVES x 8 KiB

    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t / kWave);
    for (uint32_t k = t; k < P.table_words; k += BLOCK)
        tab[k] = p.table[k];
    for (uint32_t k = t; k < P.copies * nbins; k += BLOCK)
        hist[k] = 0;
    if (t == 0) {
        seg_fill = p.seg_slots[blockIdx.x];  // segments keep filling over the launches of a chunked ingest
        seg_real_sh = p.seg_real[blockIdx.x];
    }
    __syncthreads();

    const uint2 *lut_biased = reinterpret_cast<const uint2 *>(tab) - ((int32_t)P.cell_lo - 1);
    uint32_t *my = hist + (wave % P.copies) * nbins;
    WaveStash ws;
    ws.init(slices + wave * SLICE, SLICE, p.stash + (uint64_t)blockIdx.x * p.seg_cap, &seg_fill, &seg_real_sh, p.seg_cap, tab,
            P.table_words, p.gave_up);
    const int32_t cell_last = (int32_t)(P.cell_lo + P.ncells);
    int32_t cell_first;  // pinned in a VGPR (v_med3 takes one scalar operand)
    asm volatile("v_mov_b32 %0, %1" : "=v"(cell_first) : "s"((int32_t)P.cell_lo - 1));
    const uint32_t shift = P.shift;
    auto bin_of = [&](float pw) -> uint32_t {
        const int32_t cell = __float_as_int(pw) >> shift;  // arithmetic shift: sign-bit patterns go below the table
        const uint2 e = lut_biased[clamp_cell(cell, cell_first, cell_last)];
        return e.x + (__float_as_uint(pw) >= e.y ? 1u : 0u);
    };
    auto count_bin = [&](uint32_t k) {
        // bin 0 (below every band: not counted) adds to this lane's trash word instead of being skipped
        const unsigned long long nz = __ballot(k != 0u);
        const uint32_t a_bin = (uint32_t)(uintptr_t)(lds_u32 *)&my[k];
        const uint32_t a_trash = (uint32_t)(uintptr_t)(lds_u32 *)&ws.buf[ws.trash];
        uint32_t a;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a) : "v"(a_trash), "v"(a_bin), "s"(nz));
        (void)__hip_atomic_fetch_add((lds_u32 *)(uintptr_t)a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto count_and_stash = [&](float pw, uint32_t k) {  // (one sample at a time: the launch's remainder)
        count_bin(k);
        ws.put(pw, (k & 1u) != 0u);
    };

    const float4 *data = reinterpret_cast<const float4 *>(p.data);
    // which segments: the workgroup's eight waves take the eight segments of a 64 KiB tile, wave w the w-th; the tiles grid
    // stride with the XCD skew (SkewWalk, papr_sweep_dev.h).  A wave's segments come in increasing order.
    SkewWalk walk;
    walk.init(blockIdx.x, gridDim.x, p.xcd_skew & 0xFFFFu, (p.xcd_skew >> 31) ^ 1u);
    uint64_t seg = walk.tile() * WAVES + wave;
    uint32_t segs_done = 0;
    auto load_seg = [&](float4(&x)[U], uint64_t seg) {
        const unsigned long long base = uniform_u64((unsigned long long)(data + seg * SEG_F4));  // wave-uniform: scalar base
#pragma unroll
        for (int u = 0; u < U; u++)
            x[u] = load16_nt_at(base, lane + u * kWave);
    };
    // The 1 dB table's form takes the segment from memory STRAIGHT into the transpose buffer (gfx950's 16-byte LDS-direct
    // loads, global_load_lds_dwordx4): lane l of row r lands in slot r * 64 + l whatever it asks for, so it asks for the
    // float4 that belongs there — the swizzle of xpose_slot sits in the source address, inside the same 128-byte line, and
    // the wave's eight 16-byte LDS writes, 32 VGPRs and the wake-up per arriving row are gone: 1.705 -> 1.652 ms ...
    constexpr bool DIRECT = true;
    // ... and the 0.1 dB form, whose stash logic reads LDS all along the fold, sends its powers' table lookups off IN FRONT
    // of those loads: the compiler cannot tell that an LDS-direct load does not touch the table or the slices (everything is
    // carved out of one dynamic array) and puts s_waitcnt vmcnt(0) in front of every LDS READ that follows one — with the
    // lookups behind the loads that form was 4 % slower than through registers, with them in front it is 2.7 % faster
    // (1.816 -> 1.767 ms).  The 1 dB form has one such wait and is better off with its loads out first.
    constexpr bool EARLY = DIRECT && FINE;
    typedef __attribute__((address_space(1))) const void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;
    // All eight rows off ONE scalar base and ONE M0: a row is 1024 bytes in memory and in the buffer alike, and the
    // instruction's 13-bit immediate offset moves both addresses, so with the bases in the MIDDLE of the segment the rows are
    // offsets -4096 ... +3072.  What depends on the lane is the swizzle, and that only has two values (even rows, odd rows):
    // two vector offsets for the whole kernel.  (Per row, the form with explicit addresses cost a 64-bit vector addition, two
    // scalar ones, a write of M0 and the wait states behind it.)
    const uint32_t voff_even = 16u * ((lane & ~7u) | ((lane & 7u) ^ ((lane >> 4) & 7u)));          // xpose_slot's inverse for rows 0, 2, 4, 6
    const uint32_t voff_odd = 16u * ((lane & ~7u) | ((lane & 7u) ^ (((lane >> 4) + 4u) & 7u)));   // ... and 1, 3, 5, 7
    auto load_seg_lds = [&](float4 *dst, uint64_t seg) {
        const char *mid = reinterpret_cast<const char *>(uniform_u64((unsigned long long)(data + seg * SEG_F4 + 4 * kWave)));
        lvoid *lmid = (lvoid *)(dst + 4 * kWave);
#define PAPR_LDS_ROW(r) \
        __builtin_amdgcn_global_load_lds((gvoid *)(mid + (((r) & 1) ? voff_odd : voff_even)), lmid, 16, ((r) - 4) * 1024, 2 /* nt */)
        PAPR_LDS_ROW(0);
        PAPR_LDS_ROW(1);
        PAPR_LDS_ROW(2);
        PAPR_LDS_ROW(3);
        PAPR_LDS_ROW(4);
        PAPR_LDS_ROW(5);
        PAPR_LDS_ROW(6);
        PAPR_LDS_ROW(7);
#undef PAPR_LDS_ROW
    };

    double sum = 0.0;
    TileTrack tr = {{0.f, 0.f, 0.f, 0.f, 0.f}, {0, 0, 0, 0, 0}};
    float4 *mine = xpose + wave * (kWave * 8);
    const int32_t *__restrict__ tile_E = p.tile_E_spec;
    double2 *__restrict__ seg_D = reinterpret_cast<double2 *>(p.seg_D);
    float4 x[U];
    // Everything this wave has in the vector-memory queue is IN ORDER (vmcnt), so the order of issue decides what a wait
    // costs: the segment's speculated binade is requested IN FRONT of the segment's samples (it is there when they are),
    // and the previous segment's pair is stored in front of both (a store behind the loads would have every wait for
    // the samples wait for the store's acknowledgement as well; a binade requested at the top of the loop would have the
    // wave stand still for a whole memory round trip with nothing of its own in flight).
    int E_next = PAPR_EXACT_AMBIG;
    // The segments' pairs leave the wave SIXTY-FOUR AT A TIME.  A 16-byte store per segment — one lane, 0.2 % of the bytes —
    // cost the kernel 5 % (tools/exact_form_probe.hip, profiles/r06_exact_form_probe.txt: the transposition, the two fp64
    // chains and the whole composition together are free beside a plain read, 0.875 of peak; with the store 0.835): on gfx9
    // a store counts in vmcnt like a load, so the wait for the next segment's samples at the top of every fold was also a
    // wait for the previous pair's write acknowledgement.  The pair (valid in lane 63) is handed to lane j of a per-lane
    // register instead (four v_readlane, one compare, five selects; j = the segment's number in its batch), and a full batch goes
    // out as ONE scattered store of 64 pairs where the single store used to sit.
    double2 D_keep = make_double2(0.0, 0.0);
    uint32_t seg_keep = 0, nkept = 0;  // (nkept: wave-uniform)
    auto keep_pair = [&](const double2 D, uint32_t seg_index) {
        const uint32_t j = __builtin_amdgcn_readfirstlane(nkept);
        const bool mine_now = lane == j;  // (one compare, five selects of a scalar: v_writelane takes one SGPR only on gfx9)
        const int xl = __builtin_amdgcn_readlane(__double2loint(D.x), kWave - 1), xh = __builtin_amdgcn_readlane(__double2hiint(D.x), kWave - 1);
        const int yl = __builtin_amdgcn_readlane(__double2loint(D.y), kWave - 1), yh = __builtin_amdgcn_readlane(__double2hiint(D.y), kWave - 1);
        D_keep.x = mine_now ? __hiloint2double(xh, xl) : D_keep.x;
        D_keep.y = mine_now ? __hiloint2double(yh, yl) : D_keep.y;
        seg_keep = mine_now ? seg_index : seg_keep;
        nkept = j + 1;
    };
    auto flush_pairs = [&]() {
        if (lane < nkept)
            seg_D[seg_keep] = D_keep;
        nkept = 0;
    };
    if (seg < p.nsegs) {
        E_next = tile_E[(p.seg_offset + seg) >> 1];
        asm volatile("" ::: "memory");
        if constexpr (DIRECT)
            load_seg_lds(mine, seg);
        else
            load_seg(x, seg);
    }
    while (seg < p.nsegs) {  // (wave-uniform; tiles only grow, so the first segment past the end is the end)
        walk.advance();
        const uint64_t nseg = walk.tile() * WAVES + wave;
        const uint64_t now = __builtin_amdgcn_s_memrealtime();  // (for the spill check)
        const int E = __builtin_amdgcn_readfirstlane(E_next);
        float4 y[U];
        if constexpr (DIRECT) {
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the segment is in LDS ...
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < U; j++)
                y[j] = mine[xpose_slot((int)lane, j)];
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): ... and in registers: the buffer may be written again
            asm volatile("" ::: "memory");
        } else {
#pragma unroll
            for (int r = 0; r < U; r++) {
                const int f = r * kWave + (int)lane;  // float4 slot within the segment, file order
                mine[xpose_slot(f >> 3, f & 7)] = x[r];
            }
        }
        float pw[2 * U];
        uint2 e_lut[2 * U];
        SegMax m = {0u, 0u, 0u, INT32_MIN, INT32_MIN};
        if constexpr (EARLY) {
            // (the powers and all sixteen table lookups, in front of the next segment's LDS-direct loads: see EARLY)
#pragma unroll
            for (int u = 0; u < U; u++) {
                typedef float f32x2v __attribute__((ext_vector_type(2)));
                const f32x2v a = {y[u].x, y[u].y}, b = {y[u].z, y[u].w};
                const f32x2v aa = a * a, bb = b * b;
                asm("v_add_f32 %0, %1, %2" : "=v"(pw[2 * u]) : "v"(aa.x), "v"(aa.y));
                asm("v_add_f32 %0, %1, %2" : "=v"(pw[2 * u + 1]) : "v"(bb.x), "v"(bb.y));
                segmax_fold(m, y[u], pw[2 * u], pw[2 * u + 1]);
            }
#pragma unroll
            for (int u = 0; u < 2 * U; u++)
                e_lut[u] = lut_biased[clamp_cell(__float_as_int(pw[u]) >> shift, cell_first, cell_last)];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (nkept == kWave)  // (wave-uniform: once in 64 segments)
            flush_pairs();
        asm volatile("" ::: "memory");
        // the registers are free again: the next segment's loads fly while this one is folded out of LDS — in front of
        // any spill store of this segment, so that a spill never stands between the wave and its next data
        // (same wave wrote and reads the buffer: LDS operations of one wave complete in order)
        if (nseg < p.nsegs) {
            E_next = tile_E[(p.seg_offset + nseg) >> 1];
            asm volatile("" ::: "memory");
            if constexpr (DIRECT)
                load_seg_lds(mine, nseg);
            else
                load_seg(x, nseg);
        }
        const bool valid = E != PAPR_EXACT_AMBIG;
        const double m0 = valid ? pow2_f64(E) : 0.0, ulp = valid ? pow2_f64(E - 52) : 0.0, m1 = m0 + ulp;
        double x0 = m0, x1 = m1;
        // room for the segment's 16 samples of every lane (and the trash words)?  Checked IN FRONT of the fold, behind the
        // next segment's loads: a spill's stores then have the fold's duration to drain before this wave waits for memory
        // again (vmcnt is in order and counts stores too)
        segs_done++;
        ws.spill_in_step(now, FINE ? SLICE - kWave : SLICE - (2 * U + 1) * kWave, segs_done * (uint32_t)(WAVES * 2 * SEG_F4));
        if constexpr (!DIRECT) {
#pragma unroll
            for (int j = 0; j < U; j++)
                y[j] = mine[xpose_slot((int)lane, j)];
        }
        if constexpr (!EARLY) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                // (power_of with the squares as one packed multiplication and the additions written out: see papr_sweep_kernel)
                typedef float f32x2v __attribute__((ext_vector_type(2)));
                const f32x2v a = {y[u].x, y[u].y}, b = {y[u].z, y[u].w};
                const f32x2v aa = a * a, bb = b * b;
                asm("v_add_f32 %0, %1, %2" : "=v"(pw[2 * u]) : "v"(aa.x), "v"(aa.y));
                asm("v_add_f32 %0, %1, %2" : "=v"(pw[2 * u + 1]) : "v"(bb.x), "v"(bb.y));
                segmax_fold(m, y[u], pw[2 * u], pw[2 * u + 1]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2 * U; u++) {
            const double v = (double)pw[u];
            x0 += v;  // the reference's additions themselves (papr.c:104), from the two canonical entry states
            x1 += v;
        }
        uint32_t k[2 * U];
#pragma unroll
        for (int u = 0; u < 2 * U; u++) {
            if constexpr (EARLY)
                k[u] = e_lut[u].x + (__float_as_uint(pw[u]) >= e_lut[u].y ? 1u : 0u);
            else
                k[u] = bin_of(pw[u]);  // the LUT reads in flight together
        }
        // Every sample is counted as its lookup comes back; the segment's in-band powers then go to the stash together, in
        // slots handed out per lane (WaveStash::put_tile) — which also makes room for what the segment WILL put instead of
        // its worst case (every sample of every lane in band: 4 KiB, most of a fine table's slice).
        if constexpr (FINE) {
            uint32_t flag[2 * U], cnt = 0;
#pragma unroll
            for (int u = 0; u < 2 * U; u++) {
                count_bin(k[u]);
                flag[u] = k[u] & 1u;
                cnt += flag[u];
            }
            ws.put_tile(pw, flag, cnt, SLICE, segs_done * (uint32_t)(WAVES * 2 * SEG_F4));
        } else {
            // (the coarse table puts five powers per segment: a slot per sample as it comes, nothing to wait for — with slots per
            // lane this form was 1.2 % slower on one box and even on two: profiles/r05_ab_exact_stash_slots.txt)
#pragma unroll
            for (int u = 0; u < 2 * U; u++)
                count_and_stash(pw[u], k[u]);
        }
        segmax_commit(tr, m, (uint32_t)seg);  // (the tracker remembers the SEGMENT)
        // ---- the segment's pair ----
        const double d0 = x0 - m0, d1 = x1 - m1;  // exact: multiples of the ulp inside the binade (plain sums when no binade was given)
        sum += d0;
        keep_pair(segment_pair(x0, x1, d0, d1, ulp), (uint32_t)(p.seg_offset + seg));
        seg = nseg;
    }
    flush_pairs();
    // remainder of the launch: binned here (its pass-1 part is folded in by papr_stats_finalize, its exact-sum part
    // travels raw in the sum program)
    if (blockIdx.x == gridDim.x - 1) {
        const float2 *tail = reinterpret_cast<const float2 *>(p.tail);
        for (uint32_t k0 = 0; k0 < p.tail_samples; k0 += BLOCK) {  // wave-uniform trip count
            const bool ok = k0 + t < p.tail_samples;
            const float2 v = ok ? tail[k0 + t] : make_float2(0.f, 0.f);
            const float pw = power_of(v.x, v.y);
            ws.spill_if_above(SLICE - 2 * kWave, ~0u);
            count_and_stash(pw, ok ? bin_of(pw) : 0u);
        }
    }
    ws.spill_if_above(0, ~0u);

    sweep2_record<WAVES, U, true>(sum, tr, 0, 1, data, p.base_index, p.out);  // (segment = 0 + the tracker's entry * 1)
    hist_flush<BLOCK>(hist, nbins, P.copies, p.ghist);  // (starts with a barrier: every wave has spilled)
    if (t == 0) {
        p.seg_slots[blockIdx.x] = seg_fill;
        p.seg_real[blockIdx.x] = seg_real_sh;
    }
}

// =============================================================================
// 4. pass 2 over the stash (float powers, not IQ)
// =============================================================================
// Whole-chip geometry rather than one workgroup per segment: 2 x 1024 threads per CU, every segment cut into `split`
// parts, jobs taken round robin.  (2048 workgroups of 256 threads took 31 us for the default table's 28 MB: every one of
// them stages the table and ends with one global atomic per non-empty bin, and 2048 atomics on the same 31 addresses
// are ~20 us of serialised memory-side read-modify-writes.)  Four 16-byte loads per lane are in flight.
template <bool LUT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void papr_ccdf_power_kernel(const float *__restrict__ stash,
                                                                 const unsigned long long *__restrict__ seg_counts,
                                                                 uint64_t seg_cap, uint32_t nsegs, uint32_t split,
                                                                 const uint32_t *__restrict__ table,
                                                                 papr_ccdf_params Parg,
                                                                 unsigned long long *__restrict__ ghist,
                                                                 const papr_ccdf_params *__restrict__ Pdev)
{
    const papr_ccdf_params P = uniform_params(Pdev, Parg);  // (the table may have been planned on the device: papr_true_table_kernel)
    if (P.nkeys == 0)
        return;  // (... which found nothing worth recounting)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t nbins = P.nkeys + 1;
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    uint32_t *hist = tab + P.table_words;
    for (uint32_t k = threadIdx.x; k < P.table_words; k += BLOCK)
        tab[k] = table[k];
    for (uint32_t k = threadIdx.x; k < P.copies * nbins; k += BLOCK)
        hist[k] = 0;
    __syncthreads();
    const uint2 *lut = reinterpret_cast<const uint2 *>(tab);
    uint32_t *my = hist + ((threadIdx.x / kWave) % P.copies) * nbins;
    auto count = [&](float v) {
        const uint32_t bits = __float_as_uint(v);
        const uint32_t k = LUT ? lut_bin(bits, lut, P) : search_bin(bits, tab, P);
        if (k)
            atomicAdd(&my[k], 1u);
    };
    constexpr int UNR = 4;
    for (uint32_t job = blockIdx.x; job < nsegs * split; job += gridDim.x) {
        const uint32_t seg = job / split, part = job % split;
        const float *pw = stash + (uint64_t)seg * seg_cap;  // seg_cap is a multiple of 4: 16-byte aligned
        const uint64_t n = min((uint64_t)seg_counts[seg], seg_cap);
        const uint64_t nquads = n / 4;
        const float4 *q = reinterpret_cast<const float4 *>(pw);
        const uint64_t step = (uint64_t)split * BLOCK;
        for (uint64_t i = (uint64_t)part * BLOCK + threadIdx.x; i < nquads; i += UNR * step) {
            float4 x[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++)
                if (i + u * step < nquads)
                    x[u] = load16<true>(q + i + u * step);
#pragma unroll
            for (int u = 0; u < UNR; u++)
                if (i + u * step < nquads) {
                    count(x[u].x);
                    count(x[u].y);
                    count(x[u].z);
                    count(x[u].w);
                }
        }
        if (part == 0 && threadIdx.x < (uint32_t)(n - 4 * nquads))
            count(pw[4 * nquads + threadIdx.x]);
    }
    hist_flush<BLOCK>(hist, nbins, P.copies, ghist);
    // (handing the histogram to the host from here — a ticket per workgroup, the last one copies — was measured: the
    // device-scope fence every workgroup needs in front of its ticket makes this kernel take 139 us instead of 14;
    // profiles/r02_step_timeline.txt.  The D2H copy behind the kernel stays.)
}

// ---- timed launches (papr_time_next_launch) ---------------------------------------------
static thread_local papr_launch_timer tl_timer;
static thread_local bool tl_timer_armed = false;

void papr_time_next_launch(const papr_launch_timer *t)
{
    tl_timer_armed = t != nullptr;
    if (t)
        tl_timer = *t;
}

bool papr_take_launch_timer(papr_launch_timer *out)  // (launch_maybe_timed, papr_sweep_dev.h)
{
    if (!tl_timer_armed)
        return false;
    tl_timer_armed = false;
    *out = tl_timer;
    return true;
}

// ---- launch wrappers -------------------------------------------------------------------
// Kernel forms carry ids (papr_hip_tuning.sweep_variant - 1): PAPR_SWEEP_VARIANT (111) is papr_sweep_kernel,
// PAPR_SWEEP3_VARIANT (131) papr_sweep3_kernel; every other id belongs to measure/papr_sweep_lab.hip and exists in a
// `make MEASURE=1` build only.

void papr_launch_estimate(hipStream_t st, int blocks, const void *data, uint64_t ngroups, uint32_t ratio,
                          papr_partial *out, double *group_sums, double *block_sq)
{
    launch_maybe_timed(papr_estimate_kernel, dim3(blocks), dim3(PAPR_BLOCK), 0, st, (const float4 *)data, ngroups, ratio,
                       out, group_sums, block_sq);
}

// Which XCD a queue's workgroup 0 lands on (the round-robin's start differs between queues: 6 in a plain process, 5 with RCCL's
// queues beside ours — profiles/r05_xcd_skew.txt), asked once per context on its own stream (papr_sweep_rt.cpp: xcd_even_slow).
__global__ void papr_xcd_probe_kernel(unsigned long long *out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *out = papr_xcc_id();
}
void papr_launch_xcd_probe(hipStream_t st, unsigned long long *out)
{
    hipLaunchKernelGGL(papr_xcd_probe_kernel, dim3(8), dim3(64), 0, st, out);
}

// The XCD skew of the product kernels' walk (SkewWalk): in every R rounds of tiles the workgroups on the odd XCDs sit the last
// one out (the odd workgroups, or with PAPR_MAP_EVEN_SLOW / bit 31 of papr_sweep2_params::xcd_skew the even ones).
// How much the odd XCDs lag depends on how much of a kernel's time is memory's: the tree-sum kernel (kind 0) wants R = 24 (its
// odd workgroups fold 4.2 % less: -1.0 ... -1.5 % of kernel time; R = 20 with the 0.1 dB table, kind 3: -1.6 ... -1.9 %), the exact-sum kernel's 0.1 dB
// form (kind 2) R = 48 (-1 %), its 1 dB form (kind 1) R = 96 (at 48 the EVEN workgroups finish last) — profiles/r05_xcd_skew.txt.
// PAPR_XCD_SKEW=R overrides all three (0: plain grid stride).  Launches of fewer than four periods (a chunked ingest's) are
// not skewed: nothing to even out.
uint32_t papr_sweep_xcd_skew_rounds(uint64_t ntiles, int blocks, int kind)
{
    static const long forced = [] {
        const char *e = getenv("PAPR_XCD_SKEW");
        return e && *e ? atol(e) : -1L;
    }();
    const long v = forced >= 0 ? forced : (kind == 0 ? 24 : kind == 3 ? 20 : kind == 2 ? 48 : 96);
    const uint32_t rounds = (uint32_t)(v < 2 ? 0 : (v > 4096 ? 4096 : v));
    if (!rounds || blocks <= 0 || (blocks & 7) != 0 || ntiles < 4ull * rounds * (uint64_t)blocks)
        return 0;
    return rounds;
}

int papr_sweep_variant(int variant)
{
    if (variant == PAPR_SWEEP_VARIANT)
        return variant;
#ifdef PAPR_MEASURE
    return papr_lab_sweep_variant(variant);
#else
    return -1;
#endif
}

int papr_sweep_geometry(int variant, int *threads, uint64_t *tile_samples, size_t *stash_lds)
{
    if (variant == PAPR_SWEEP_VARIANT) {
        *threads = PAPR_SWEEP_THREADS;
        *tile_samples = 2ull * PAPR_SWEEP_THREADS * PAPR_SWEEP_LOADS;
        *stash_lds = (size_t)(PAPR_SWEEP_THREADS / kWave) * PAPR_SWEEP_SLICE_FLOATS * sizeof(float) + 16;
        return 0;
    }
#ifdef PAPR_MEASURE
    return papr_lab_sweep_geometry(variant, threads, tile_samples, stash_lds);
#else
    return -1;
#endif
}

void papr_launch_sweep(hipStream_t st, int variant, int blocks, size_t lds_bytes, const void *data, uint64_t ntiles,
                       uint64_t base_index, int map, papr_partial *out, const void *tail, uint32_t tail_samples,
                       const uint32_t *table, const papr_ccdf_params &P, unsigned long long *ghist, float *stash,
                       unsigned long long *seg_counts, uint64_t seg_cap, unsigned long long *gave_up,
                       unsigned long long *seg_real, const papr_ccdf_params *Pdev)
{
    if (variant == PAPR_SWEEP_VARIANT) {
        // (the product kernel walks grid stride whatever `map` says, with the XCD skew: its period rides in map's upper bits)
        map = (map & (0x7F | PAPR_MAP_EVEN_SLOW)) | (int)(papr_sweep_xcd_skew_rounds((uint64_t)ntiles, blocks, (map & 0x80) || P.nkeys > 128u ? 3 : 0) << 8);
        launch_maybe_timed(papr_sweep_kernel, dim3(blocks), dim3(PAPR_SWEEP_THREADS), lds_bytes, st, (const float4 *)data, ntiles,
                           base_index, map, out, (const float2 *)tail, tail_samples, table, P, ghist, stash, seg_counts, seg_cap,
                           gave_up, seg_real, Pdev);
        return;
    }
#ifdef PAPR_MEASURE
    map &= 0x7F;  // (the laboratory's forms compare `map` with PAPR_MAP_*: the product kernel's 0.1 dB and XCD-parity bits are not theirs)
    papr_lab_launch_sweep(st, variant, blocks, lds_bytes, data, ntiles, base_index, map, out, tail, tail_samples, table, P, ghist,
                          stash, seg_counts, seg_cap, gave_up, seg_real, Pdev);
#endif
}

int papr_sweep2_geometry(int variant, int *threads, uint64_t *seg_samples, size_t *lds_fixed, int *exact)
{
#ifdef PAPR_MEASURE
    return papr_lab_sweep2_geometry(variant, threads, seg_samples, lds_fixed, exact);
#else
    (void)variant, (void)threads, (void)seg_samples, (void)lds_fixed, (void)exact;
    return -1;
#endif
}

void papr_launch_sweep2(hipStream_t st, int variant, int blocks, size_t lds_bytes, const papr_sweep2_params &p)
{
#ifdef PAPR_MEASURE
    papr_lab_launch_sweep2(st, variant, blocks, lds_bytes, p);
#else
    (void)st, (void)variant, (void)blocks, (void)lds_bytes, (void)p;
#endif
}

int papr_sweep3_geometry(int variant, int *threads, size_t *lds_fixed, int *exact)
{
    if (variant != PAPR_SWEEP3_VARIANT)
        return -1;
    *threads = PAPR_SWEEP_THREADS;
    *lds_fixed = (size_t)(PAPR_SWEEP_THREADS / kWave) * (PAPR_SWEEP3_SLICE_FLOATS * sizeof(float) + 8192u) + 16;
    if (exact)
        *exact = 1;
    return 0;
}

void papr_launch_sweep3(hipStream_t st, int variant, int blocks, size_t lds_bytes, const papr_sweep2_params &p)
{
    if (variant != PAPR_SWEEP3_VARIANT)
        return;
    papr_sweep2_params q = p;
    q.lds_bytes = (uint32_t)lds_bytes;
    q.xcd_skew = (p.xcd_skew & 0x80000000u) | papr_sweep_xcd_skew_rounds((p.nsegs + 7) / 8, blocks, q.fine_table ? 2 : 1);
    // (which form: the table's size is known to whoever planned it — the host, or papr_guess_bands_kernel through p.fine_hint)
    if (q.fine_table)
        launch_maybe_timed(papr_sweep3_kernel<true>, dim3(blocks), dim3(PAPR_SWEEP_THREADS), lds_bytes, st, q);
    else
        launch_maybe_timed(papr_sweep3_kernel<false>, dim3(blocks), dim3(PAPR_SWEEP_THREADS), lds_bytes, st, q);
}

void papr_launch_ccdf_power(hipStream_t st, int num_cus, bool lut, size_t lds_bytes, const float *stash,
                            const unsigned long long *seg_counts, uint64_t seg_cap, uint32_t nsegs, const uint32_t *table,
                            const papr_ccdf_params &P, unsigned long long *ghist, const papr_ccdf_params *Pdev)
{
    constexpr int kBlock = 1024;
    const uint32_t chip = 2u * (uint32_t)(num_cus > 0 ? num_cus : 1);  // workgroups that are resident together
    const uint32_t split = nsegs && nsegs < chip ? chip / nsegs : 1u;
    const uint32_t jobs = nsegs * split;
    const dim3 grid(jobs < chip ? (jobs ? jobs : 1u) : chip);
    if (lut)
        launch_maybe_timed((papr_ccdf_power_kernel<true, kBlock>), grid, dim3(kBlock), lds_bytes, st, stash, seg_counts,
                           seg_cap, nsegs, split, table, P, ghist, Pdev);
    else
        launch_maybe_timed((papr_ccdf_power_kernel<false, kBlock>), grid, dim3(kBlock), lds_bytes, st, stash, seg_counts,
                           seg_cap, nsegs, split, table, P, ghist, Pdev);
}

void papr_sweep_prepare_device(void)
{
    const int want = papr_ccdf_max_dynamic_lds();
    (void)hipFuncSetAttribute((const void *)papr_sweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, want);
    (void)hipFuncSetAttribute((const void *)papr_sweep3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, want);
    (void)hipFuncSetAttribute((const void *)papr_sweep3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, want);
    (void)hipFuncSetAttribute((const void *)papr_ccdf_power_kernel<true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, want);
    (void)hipFuncSetAttribute((const void *)papr_ccdf_power_kernel<false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, want);
#ifdef PAPR_MEASURE
    papr_lab_prepare_device();
#endif
}
