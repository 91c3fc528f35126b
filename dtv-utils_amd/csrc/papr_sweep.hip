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
// The default form (variant 40) is papr_sweep_kernel<512, 8, ..., SMODE 43> launched as ONE persistent workgroup
// per CU: eight waves per CU with eight 16-byte loads each in flight is what reads fastest, and with two waves
// per SIMD the per-sample code has to be free of branches and exec-masked regions.

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "papr_kernels.h"
#include "papr_device.h"
#include "papr_stream.h"

namespace {


typedef __attribute__((address_space(3))) uint32_t lds_u32;  // an LDS word addressed as LDS (ds_read/ds_write, not flat)

// min(max(cell, first), last) in one instruction (the compiler will not form med3 from min/max when it
// cannot prove last >= 0)
__device__ __forceinline__ int32_t clamp_cell(int32_t cell, int32_t first, int32_t last)
{
    int32_t r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(cell), "v"(first), "s"(last));  // one SGPR operand at most on gfx9
    return r;
}

// which tile of its group an estimate row is read from
__device__ __forceinline__ uint32_t papr_estimate_pick(uint64_t g, uint32_t ratio)
{
    return (uint32_t)(((g + 1) * 0x9E3779B97F4A7C15ull) >> 40) % ratio;
}

// Per-wave append buffer in LDS + its spill to the workgroup's segment of the HBM stash.  Lanes that
// hold an in-band power reserve a slot with a returning LDS atomic on the wave's own counter (three
// VALU instructions per sample; a ballot/mbcnt compaction costs seven).
// A sweep whose bands catch most of the stream (a constant-envelope capture: every power sits next to the mean)
// cannot be answered from the stash, and must not cost more than the pass it replaces: each spill compares what the
// workgroup stashed in this launch with what it folded, and once more than half of it was in band — or the segment is
// full — the wave GIVES UP for the whole workgroup: it overwrites the LUT in LDS with "bin 0 everywhere" (no counter,
// no stash: the rest of the launch runs at pass-1 speed) and pushes the segment's length past its capacity, which the
// host reads as `stash full` and answers with the plain pass 2.  Pass-1 results do not depend on the LUT.
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v)  // the first active lane's value
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// the table's geometry: the kernel argument, or — when a kernel on the same stream has just built the table — what that
// kernel left in device memory; either way in scalar registers (every lane reads the same words)
__device__ __forceinline__ papr_ccdf_params uniform_params(const papr_ccdf_params *dev, const papr_ccdf_params &arg)
{
    static_assert(sizeof(papr_ccdf_params) == 9 * sizeof(uint32_t), "nine words");
    papr_ccdf_params P = arg;
    if (dev) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(dev);
        uint32_t w[9];
#pragma unroll
        for (int k = 0; k < 9; k++)
            w[k] = __builtin_amdgcn_readfirstlane(q[k]);
        P.shift = w[0];
        P.cell_lo = w[1];
        P.ncells = w[2];
        P.nkeys = w[3];
        P.above_lo = w[4];
        P.above_count = w[5];
        P.table_words = w[6];
        P.copies = w[7];
        P.search_step = w[8];
    }
    return P;
}

__device__ __forceinline__ void sweep_give_up(uint32_t *tab, uint32_t table_words, uint32_t neutral_x,
                                              unsigned long long *seg_fill, uint64_t seg_cap,
                                              unsigned long long *gave_up)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    for (uint32_t k = lane; 2 * k + 1 < table_words; k += kWave) {
        tab[2 * k] = neutral_x;
        tab[2 * k + 1] = 0xFFFFFFFFu;
    }
    if (lane == 0) {
        atomicAdd(seg_fill, (unsigned long long)seg_cap + 1ull);
        atomicAdd(gave_up, 1ull);  // (for papr_hip_sweep_info: how often the rule fired)
    }
}
constexpr uint32_t kGiveUpMin = 16384;  // in-band samples of a workgroup before the ratio test means anything

typedef float f32x4s __attribute__((ext_vector_type(4)));

// store policies for the stash: 0 plain, 1 nontemporal, 2 write-through (sc0 sc1)
template <int WT>
__device__ __forceinline__ void store16(float *p, f32x4s v)
{
    if constexpr (WT == 2) {
        // The s_nop belongs to the store: hipcc's hazard recogniser does not look inside inline asm, and gfx940+ needs
        // two wait states between a VMEM store of more than 8 bytes and a VALU write to its data registers — without
        // them the next instruction can overwrite the powers before the store has read them (seen as wrong, run-to-run
        // different stash contents whenever the scheduler happened to put a VALU write right behind this store).
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" : : "v"(p), "v"(v) : "memory");
    } else if constexpr (WT == 1) {
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4s *>(p));
    } else {
        *reinterpret_cast<f32x4s *>(p) = v;
    }
}

// MODE bit 0 (SP16): spill in 16-byte stores (a partial quad padded with quiet NaNs, which the recount ignores)
// instead of dwords.  MODE bit 1 (BALLOT): the slice is this wave's alone, so its fill count can live in a scalar
// register and slots be handed out by ballot + mbcnt — no returning LDS atomic (and no wait for it) per in-band sample.
// MODE bit 3 (NOBR, with BALLOT): no branch and no exec-masked region per sample — every lane writes, its power to
// its slot or to a trash word of its own at the end of the slice.  MODE bit 4: plain instead of write-through spill
// stores (measurement).  (Bits 2 and 5 — histogram sets, double slice — belong to the kernel, not to this struct.)
template <int MODE = 0>
struct WaveStashT {
    static constexpr bool SP16 = (MODE & 1) != 0, BALLOT = (MODE & 2) != 0, NOBR = (MODE & 8) != 0;
    uint32_t nfill = 0;                 // BALLOT: entries in buf (wave-uniform)
    float *buf;                         // this wave's slice of LDS
    uint32_t *fill;                     // LDS: entries in buf (this wave's counter)
    float *__restrict__ seg;            // this workgroup's stash segment
    unsigned long long *seg_fill;       // LDS: floats reserved in the segment so far (may run past seg_cap)
    uint64_t seg_cap;
    uint32_t *tab;                      // LDS: the LUT (sweep_give_up)
    uint32_t table_words, neutral_x;
    unsigned long long seg_start;       // the segment's length when this launch began
    unsigned long long *gave_up;        // device counter of give-ups
    unsigned long long *seg_real;       // LDS: powers stashed without padding (SP16)
    uint32_t trash = 0;                 // NOBR: this lane's own word at the end of the slice, where what is not in band goes
    uint32_t sbase = 0, sbytes = 0;     // NOBR: LDS byte address of the slice, and of its next free slot (wave-uniform)

    __device__ __forceinline__ void put(float pw, bool take)
    {
        if constexpr (BALLOT && NOBR) {
            // no branch at all: every lane writes — its power to its slot, or to its own trash word (a wave with only two
            // waves per SIMD beside it cannot hide a v_cmp -> s_cbranch round per sample)
            // (the select is written out: from `take ? at : trash` hipcc makes an exec-masked region per sample)
            // Addresses in bytes: slot = rank among the takers * 4 + (slice base + nfill * 4), the bracket wave-uniform
            // (one SALU op, one scalar operand of the v_lshl_add) — no copy of nfill into a vector register per sample.
            const unsigned long long m = __ballot(take);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            // (sbytes: the LDS address of the next free slot, wave-uniform.  Both steps are written out: hipcc turns the
            // byte address back into base + 4 * (count + rank), one more vector addition per sample)
            uint32_t a_slot;
            asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a_slot) : "v"(rank), "s"(__builtin_amdgcn_readfirstlane(sbytes)));
            const uint32_t a_trash = (uint32_t)(uintptr_t)(lds_u32 *)&buf[trash];
            uint32_t a;
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a) : "v"(a_trash), "v"(a_slot), "s"(m));
            *(__attribute__((address_space(3))) float *)(uintptr_t)a = pw;
            // (s_lshl2_add_u32 writes SCC: said, so that the compiler never schedules it between a compare and its consumer)
            asm("s_lshl2_add_u32 %0, %1, %2" : "=s"(sbytes) : "s"((uint32_t)__popcll(m)), "s"(__builtin_amdgcn_readfirstlane(sbytes)) : "scc");
        } else if constexpr (BALLOT) {
            const unsigned long long m = __ballot(take);
            if (m) {  // (wave-uniform)
                const uint32_t at = nfill + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (take)
                    buf[at] = pw;
                nfill += (uint32_t)__popcll(m);
            }
        } else {
            if (take)
                buf[atomicAdd(fill, 1u)] = pw;
        }
    }
    // spill if more than `limit` entries are waiting (wave-uniform decision); `folded` = samples this workgroup
    // has folded in this launch, about
    __device__ __forceinline__ void spill_if_above(uint32_t limit, uint32_t folded)
    {
        uint32_t n;
        if constexpr (BALLOT && NOBR) {
            n = (sbytes - sbase) >> 2;
            if (n <= limit)
                return;
            sbytes = sbase;
        } else if constexpr (BALLOT) {
            n = nfill;
            if (n <= limit)
                return;
            nfill = 0;
        }
        __builtin_amdgcn_wave_barrier();  // LDS is in-order per wave; this pins the compiler's order too
        if constexpr (!BALLOT) {
            // other lanes' atomics: never cached.  The cast matters: through a generic pointer the volatile read is a
            // flat_load sc0 sc1 followed by s_waitcnt vmcnt(0) — it drains the prefetched tile's loads every iteration
            n = __builtin_amdgcn_readfirstlane(*(volatile lds_u32 *)(lds_u32 *)fill);
            if (n <= limit)
                return;
        }
        const uint32_t lane = threadIdx.x & (kWave - 1);
        const uint32_t nres = SP16 ? ((n + 3u) & ~3u) : n;  // floats reserved in the segment
        unsigned long long pos = 0;
        if (lane == 0) {
            pos = atomicAdd(seg_fill, (unsigned long long)nres);  // counts even what no longer fits: the host sees the overflow
            if constexpr (SP16)
                atomicAdd(seg_real, (unsigned long long)n);
            if constexpr (!BALLOT)
                *(volatile lds_u32 *)(lds_u32 *)fill = 0;
        }
        pos = uniform_u64(pos);  // lane 0's value, in scalar registers
        if constexpr (SP16) {
            for (uint32_t i = 4 * lane; i < nres; i += 4 * kWave) {  // (the slice and the segment are 16-byte aligned)
                f32x4s v = *reinterpret_cast<const f32x4s *>(buf + i);
                const float pad = __uint_as_float(PAPR_STASH_PAD_BITS);
                v.y = i + 1 < n ? v.y : pad;
                v.z = i + 2 < n ? v.z : pad;
                v.w = i + 3 < n ? v.w : pad;
                if (pos + i + 4 <= seg_cap)
                    store16<(MODE & 16) ? 0 : 2>(seg + pos + i, v);
            }
        } else {
            for (uint32_t i = lane; i < n; i += kWave)
                if (pos + i < seg_cap) {  // write-through (sc0 sc1): 1-3 % faster than leaving these lines dirty in L2 for a later eviction
                    if constexpr ((MODE & 16) != 0)
                        seg[pos + i] = buf[i];  // (MODE bit 4: plain stores — measurement)
                    else
                        __hip_atomic_store(&seg[pos + i], buf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t got = (uint32_t)(pos - seg_start) + nres;  // (a workgroup folds < 2^32 samples per launch)
        if (pos <= seg_cap && (pos + nres > seg_cap || (got >= kGiveUpMin && got > folded / 2)))
            sweep_give_up(tab, table_words, neutral_x, seg_fill, seg_cap, gave_up);  // (pos > seg_cap: someone already did)
    }
};
typedef WaveStashT<0> WaveStash;

}  // namespace

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
            x[u] = load16<false>(data + tile * TILE_F4 + (uint64_t)u * PAPR_BLOCK + threadIdx.x);
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
    *n_total_dev = n;
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
    uint32_t zero_words, const papr_est_record *__restrict__ recs, uint32_t nrecs, uint32_t my_rank)
{
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
        const float db = graph ? (float)t * 0.1f : (float)t;
        const float lv = (float)(pow(10.0, (double)(db / 10.0f)) * mean) * spoil;
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
                             uint32_t my_rank)
{
    hipLaunchKernelGGL(papr_guess_bands_kernel, dim3(1), dim3(1024), 0, st, est_partials, est_sq, est_blocks, ngroups, sampled,
                       nsamples, ratio, graph, max_db, spoil, band_override, copies, compact, soft_lds, table, table_cap_words,
                       out_dev, out_host, zero, zero_words, recs, nrecs, my_rank);
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
                                                                const unsigned long long *__restrict__ nsamples_dev)
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
        if (graph) {
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
                            const unsigned long long *gave_up, const unsigned long long *nsamples_dev)
{
    hipLaunchKernelGGL(papr_true_table_kernel, dim3(1), dim3(1024), 0, st, result, nsamples, graph, copies, soft_lds, table,
                       table_cap_words, out_dev, out_host, zero, zero_words, gave_up, nsamples_dev);
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

namespace {

struct TileTrack {
    float best[5];     // peak power, re_pos, re_neg, im_pos, im_neg
    uint32_t iter[5];  // loop iteration in which `best` first appeared
};

__device__ __forceinline__ int32_t imax3(int32_t a, int32_t b, int32_t c) { return max(max(a, b), c); }
__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) { return max(max(a, b), c); }

template <int U>
__device__ __forceinline__ void track_tile(TileTrack &tr, const float4 (&x)[U], const float (&pw)[2 * U], uint32_t it)
{
    uint32_t m_pk = 0, m_rn = 0, m_in = 0;             // unsigned max: most negative float, or largest power
    int32_t m_rp = INT32_MIN, m_ip = INT32_MIN;        // signed max: largest positive float
#pragma unroll
    for (int u = 0; u < U; u++) {
        m_pk = umax3(m_pk, __float_as_uint(pw[2 * u]), __float_as_uint(pw[2 * u + 1]));
        m_rp = imax3(m_rp, __float_as_int(x[u].x), __float_as_int(x[u].z));
        m_rn = umax3(m_rn, __float_as_uint(x[u].x), __float_as_uint(x[u].z));
        m_ip = imax3(m_ip, __float_as_int(x[u].y), __float_as_int(x[u].w));
        m_in = umax3(m_in, __float_as_uint(x[u].y), __float_as_uint(x[u].w));
    }
    const float c[5] = {__uint_as_float(m_pk), __int_as_float(m_rp), __uint_as_float(m_rn), __int_as_float(m_ip),
                        __uint_as_float(m_in)};
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const bool win = (k == 2 || k == 4) ? (c[k] < tr.best[k]) : (c[k] > tr.best[k]);  // strict: first tile wins
        tr.best[k] = win ? c[k] : tr.best[k];
        tr.iter[k] = win ? it : tr.iter[k];
    }
}

// Workgroup record of the sweep kernel.  THREADS = workgroup size, ROW = lanes that share a tile row, tl = this
// lane's position in the row.
template <int THREADS, int ROW, int U>
__device__ __forceinline__ void sweep_record(double sum, const TileTrack &tr, const TileWalk &w, const float4 *__restrict__ data,
                                             uint64_t base_index, uint32_t tl, papr_partial *__restrict__ out)
{
    // Workgroup record.  The lanes only know in WHICH tile their extreme first appeared; finding the slot means
    // re-reading that tile, which is uncoalesced (every lane another tile: 64-128 B fetched per 16 B used), so it is
    // done by the workgroup's winners only: reduce the VALUES first, then just the lanes that hold the winning value
    // (normally one) look up their slot, then the smallest index among them wins — the reference's first occurrence.
    constexpr int kWaves = THREADS / kWave;
    constexpr uint64_t TILE_F4 = (uint64_t)ROW * U;
    const uint32_t t = threadIdx.x;
    __shared__ double sh_sum[kWaves];
    __shared__ float sh_val[kWaves][5];
    __shared__ unsigned long long sh_idx[kWaves][5];
    const int lane = t & (kWave - 1), wave = t / kWave;
    float wv[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        float v = tr.best[k];
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
            const float o = __shfl_down(v, off, kWave);
            v = (k == 2 || k == 4) ? (o < v ? o : v) : (o > v ? o : v);
        }
        wv[k] = v;
    }
    const double wsum = wave_reduce_sum(sum);
    if (lane == 0) {
        sh_sum[wave] = wsum;
#pragma unroll
        for (int k = 0; k < 5; k++)
            sh_val[wave][k] = wv[k];
    }
    __syncthreads();
    float win[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        float v = sh_val[0][k];
        for (int wq = 1; wq < kWaves; wq++) {
            const float o = sh_val[wq][k];
            v = (k == 2 || k == 4) ? (o < v ? o : v) : (o > v ? o : v);
        }
        win[k] = v;
    }
    unsigned long long idx[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        idx[k] = ~0ull;
        if (win[k] != 0.f && tr.best[k] == win[k]) {  // a tracker that never fired keeps value 0 and reports index 0
            const uint64_t tile = w.first + (uint64_t)tr.iter[k] * w.stride;
            const float4 *q = data + tile * TILE_F4 + tl;
            for (int u = U - 1; u >= 0; u--) {  // last match written last = first slot wins
                const float4 x = q[(uint64_t)u * ROW];
                const float a = k == 0 ? power_of(x.x, x.y) : (k <= 2 ? x.x : x.y);
                const float b = k == 0 ? power_of(x.z, x.w) : (k <= 2 ? x.z : x.w);
                const uint64_t i0 = base_index + 2 * (tile * TILE_F4 + (uint64_t)u * ROW + tl);
                if (b == win[k])
                    idx[k] = i0 + 1;
                if (a == win[k])
                    idx[k] = i0;
            }
        }
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_down(idx[k], off, kWave);
            idx[k] = o < idx[k] ? o : idx[k];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; k++)
            sh_idx[wave][k] = idx[k];
    }
    __syncthreads();
    if (t == 0) {
        papr_partial q;
        q.sum = sh_sum[0];
        for (int wq = 1; wq < kWaves; wq++)  // fixed order => deterministic sum (as block_reduce_stats)
            q.sum += sh_sum[wq];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            unsigned long long best_idx = sh_idx[0][k];
            for (int wq = 1; wq < kWaves; wq++)
                best_idx = sh_idx[wq][k] < best_idx ? sh_idx[wq][k] : best_idx;
            q.val[k] = win[k];
            q.idx[k] = win[k] != 0.f ? best_idx : 0;
        }
        q.pad = 0;
        out[blockIdx.x] = q;
    }
}

}  // namespace

// ABL (measurement only): leave out one ingredient to see what it costs — 1 stash, 2 histogram,
// 4 LUT lookup, 8 trackers, 16 sum, 32 spill check.  The results of such a launch are meaningless.
// LUT2: the compact band-edge table of papr_kernels.h (two edges per cell: 1-8 KiB instead of 32-40), which lets small
// workgroups — the geometry papr_stats_kernel runs best in — afford a table of their own.
template <int BLOCK, int U, bool NT, int PIPE, int ABL = 0, bool LUT2 = false, int SMODE = 0>
__global__ __launch_bounds__(BLOCK) void papr_sweep_kernel(const float4 *__restrict__ data, uint64_t ntiles,
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
    // the table's geometry: an argument, or — when papr_guess_bands_kernel built the table just before this launch,
    // without the host in between — read from where that kernel left it (wave-uniform loads: scalar registers)
    const papr_ccdf_params P = uniform_params(Pdev, Parg);
    constexpr uint64_t TILE_F4 = (uint64_t)BLOCK * U;
    constexpr uint32_t SLICE = papr_sweep_slice_floats(U) * ((SMODE & 32) ? 2u : 1u);  // (bit 5: twice the slice — measurement)
    __shared__ unsigned long long seg_fill, seg_real_sh;
    __shared__ uint32_t wave_fill[BLOCK / kWave];
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
    if (t < BLOCK / kWave)
        wave_fill[t] = 0;
    __syncthreads();

    const uint2 *lut_biased = reinterpret_cast<const uint2 *>(tab) - ((int32_t)P.cell_lo - 1);
    // SMODE bit 2 (HSETS): histogram laid out [bin][copy] with a power-of-two number of copies, and the lanes of a wave
    // spread over eight of them: the samples pile up in a handful of bins (63 % below the first band, 8 % in the next
    // bin, ...), and 64 lanes adding to five addresses is what the LDS spends its time on (profiles/r02_work_probe.txt)
    constexpr bool HSETS = (SMODE & 4) != 0;
    const uint32_t csh = HSETS ? 31u - (uint32_t)__clz((int)P.copies) : 0u;
    const uint32_t mycopy = HSETS ? (((t / kWave) * 8u + (t & 7u)) & ((1u << csh) - 1u)) : 0u;
    uint32_t *my = HSETS ? hist + mycopy : hist + ((t / kWave) % P.copies) * nbins;
    WaveStashT<SMODE> ws{0u, slices + (t / kWave) * SLICE, &wave_fill[t / kWave], stash + (uint64_t)blockIdx.x * seg_cap,
                        &seg_fill, seg_cap, tab, P.table_words, LUT2 ? PAPR_LUT2_NEVER : 0u, seg_fill, gave_up,
                        &seg_real_sh};
    ws.trash = SLICE - kWave + (t & (kWave - 1));
    ws.sbase = ws.sbytes = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u32 *)ws.buf);
    // cell index straight from the bit pattern: lut_biased[cell] with cell clamped to [cell_lo - 1, cell_lo + ncells]
    const int32_t cell_last = (int32_t)(P.cell_lo + P.ncells);
    int32_t cell_first;  // pinned in a VGPR for the whole kernel (v_med3 takes one scalar operand)
    asm volatile("v_mov_b32 %0, %1" : "=v"(cell_first) : "s"((int32_t)P.cell_lo - 1));
    const uint32_t shift = P.shift;

    const uint32_t offmask = (1u << shift) - 1u;
    auto bin_of = [&](float pw) -> uint32_t {
        const int32_t cell = __float_as_int(pw) >> shift;   // arithmetic shift: sign-bit patterns go below
        const uint2 e = lut_biased[clamp_cell(cell, cell_first, cell_last)];
        if constexpr (LUT2) {
            const uint32_t off = __float_as_uint(pw) & offmask;
            return (e.x >> PAPR_LUT2_OFF_BITS) + (off >= (e.x & PAPR_LUT2_NEVER) ? 1u : 0u) + (off >= e.y ? 1u : 0u);
        } else {
            return e.x + (__float_as_uint(pw) >= e.y ? 1u : 0u);
        }
    };
    auto count_and_stash = [&](float pw, uint32_t k) {
        if constexpr (!(ABL & 2)) {
            if constexpr ((SMODE & 8) != 0) {
                // branch-free: bin 0 (below every band: not counted) adds to this lane's trash word instead
                const unsigned long long nz = __ballot(k != 0u);
                const uint32_t a_bin = (uint32_t)(uintptr_t)(lds_u32 *)&my[HSETS ? (k << csh) : k];
                const uint32_t a_trash = (uint32_t)(uintptr_t)(lds_u32 *)&ws.buf[ws.trash];
                uint32_t a;
                asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a) : "v"(a_trash), "v"(a_bin), "s"(nz));
                (void)__hip_atomic_fetch_add((lds_u32 *)(uintptr_t)a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (k) {
                atomicAdd(&my[HSETS ? (k << csh) : k], 1u);
            }
        }
        if constexpr (!(ABL & 1))
            ws.put(pw, (k & 1u) != 0u);
    };

    double sum = 0.0;
    TileTrack tr = {{0.f, 0.f, 0.f, 0.f, 0.f}, {0, 0, 0, 0, 0}};
    const TileWalk w = tile_walk(blockIdx.x, gridDim.x, ntiles, map);
    auto fold = [&](const float4(&x)[U], uint32_t it) {
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
        if constexpr (!(ABL & 16)) {
#pragma unroll
            for (int u = 0; u < 2 * U; u++)
                sum += (double)pw[u];  // same order as papr_stats_kernel
        } else {
            sum += (double)(pw[0] + pw[2 * U - 1]);
        }
        if constexpr (!(ABL & 8))
            track_tile<U>(tr, x, pw, it);
        else
            tr.best[0] = fmaxf(tr.best[0], pw[1]);
        uint32_t k[2 * U];
#pragma unroll
        for (int u = 0; u < 2 * U; u++)
            k[u] = (ABL & 4) ? (__float_as_uint(pw[u]) >> 30) : bin_of(pw[u]);  // all LUT reads of the tile in flight together
#pragma unroll
        for (int u = 0; u < 2 * U; u++)
            count_and_stash(pw[u], k[u]);
        if constexpr (!(ABL & 32) && !(SMODE & 64))
            ws.spill_if_above(SLICE - 2 * U * kWave - ((SMODE & 8) ? kWave : 0), (it + 1) * (uint32_t)(2 * TILE_F4));  // the next tile might not fit
    };

    const float4 *p = data + w.first * TILE_F4 + t;
    const uint64_t step = w.stride * TILE_F4;
    if constexpr (PIPE == 2) {
        // true double buffering (two register sets, loop unrolled by two): no cur = nxt copies
        static_assert(!(SMODE & 64), "the early spill check belongs to the prefetching loop");
        float4 a[U], b[U];
        const float4 *plast = data + (w.first + (uint64_t)(w.count ? w.count - 1 : 0) * w.stride) * TILE_F4 + t;
        if (w.count)
            load_tile<BLOCK, U, NT>(a, p);
        uint32_t it = 0;
        for (; it + 1 < w.count; it += 2) {
            load_tile<BLOCK, U, NT>(b, p + step);
            fold(a, it);
            p += 2 * step;
            load_tile<BLOCK, U, NT>(a, it + 2 < w.count ? p : plast);  // past the end: harmless re-read
            fold(b, it + 1);
        }
        if (it < w.count)
            fold(a, it);
    } else if constexpr (PIPE == 1) {
        float4 cur[U], nxt[U];
        if (w.count)
            load_tile<BLOCK, U, NT>(cur, p);
        for (uint32_t it = 0; it < w.count; it++) {
            p += step;
            if (it + 1 < w.count)
                load_tile<BLOCK, U, NT>(nxt, p);
            // SMODE bit 6: the spill check IN FRONT of the fold, behind the next tile's loads — a spill's stores then have
            // the fold's duration to drain before this wave waits for memory again (vmcnt is in order and counts stores)
            if constexpr ((SMODE & 64) != 0)
                ws.spill_if_above(SLICE - 2 * U * kWave - ((SMODE & 8) ? kWave : 0), (it + 1) * (uint32_t)(2 * TILE_F4));
            fold(cur, it);
#pragma unroll
            for (int u = 0; u < U; u++)
                cur[u] = nxt[u];
        }
    } else {
        static_assert(!(SMODE & 64), "the early spill check belongs to the prefetching loop");
        for (uint32_t it = 0; it < w.count; it++, p += step) {
            float4 x[U];
            load_tile<BLOCK, U, NT>(x, p);
            fold(x, it);
        }
    }
    // sub-tile remainder of the shard: binned here (its pass-1 part is folded in by papr_stats_finalize)
    if (blockIdx.x == gridDim.x - 1) {
        if constexpr ((SMODE & 64) != 0)
            ws.spill_if_above(SLICE - 2 * kWave, ~0u);
        for (uint32_t k0 = 0; k0 < tail_samples; k0 += BLOCK) {  // wave-uniform trip count
            const bool valid = k0 + t < tail_samples;
            const float2 x = valid ? tail[k0 + t] : make_float2(0.f, 0.f);
            const float pw = power_of(x.x, x.y);
            count_and_stash(pw, valid ? bin_of(pw) : 0u);
            ws.spill_if_above(SLICE - kWave - ((SMODE & 8) ? kWave : 0), ~0u);
        }
    }
    ws.spill_if_above(0, ~0u);

    sweep_record<BLOCK, BLOCK, U>(sum, tr, w, data, base_index, t, out);
    if constexpr (HSETS) {
        __syncthreads();  // every wave has spilled and counted
        for (uint32_t b = t; b < nbins; b += BLOCK) {
            unsigned long long sb = 0;
            for (uint32_t c = 0; c < (1u << csh); c++)
                sb += hist[(b << csh) + c];
            if (sb)
                atomicAdd(&ghist[b], sb);
        }
    } else {
        hist_flush<BLOCK>(hist, nbins, P.copies, ghist);  // (starts with a barrier: every wave has spilled)
    }
    if (t == 0) {
        seg_counts[blockIdx.x] = seg_fill;
        seg_real[blockIdx.x] = (SMODE & 1) ? seg_real_sh : seg_fill;  // (dword spills: no padding, the two are the same)
    }
}

// =============================================================================
// 3a'. the sweep with the two passes on different waves of one workgroup
// =============================================================================
// papr_sweep_kernel makes every wave do everything: loads, pass 1, LUT lookups, LDS atomics, stash — 21.8 VALU per
// sample and a wait on the LDS between issuing a tile's loads and folding it.  Here a workgroup's first PW waves are
// LOADERS: they run pass 1's loop (loads one tile ahead, power, sum, per-tile trackers) and leave the tile's POWERS
// (4 bytes per sample: half the input) in a small LDS ring, in slots of 8 powers per lane (2 KiB); every loader feeds
// NB BINNER waves, which take its slots in turn and do the band lookup, the histogram and the stash.  No barrier in
// the loop: a loader and its binners talk through two LDS words per slot (filled / consumed sequence numbers), and LDS
// operations of one wave execute in order, so a slot's data is there when its sequence number is.  Compact LUT (two
// edges per cell) always: the ring takes the LDS the wide table would need.
template <int PW, int NB, int LU, int DEPTH>
__global__ __launch_bounds__((PW + PW * NB) * kWave) void papr_sweep_split_kernel(
    const float4 *__restrict__ data, uint64_t ntiles, uint64_t base_index, int map, papr_partial *__restrict__ out,
    const float2 *__restrict__ tail, uint32_t tail_samples, const uint32_t *__restrict__ table, papr_ccdf_params P,
    unsigned long long *__restrict__ ghist, float *__restrict__ stash, unsigned long long *__restrict__ seg_counts,
    uint64_t seg_cap, unsigned long long *__restrict__ gave_up, unsigned long long *__restrict__ seg_real)
{
    constexpr int ROW = PW * kWave;                      // loader lanes: one tile row
    constexpr int BW = PW * NB;                          // binner waves
    constexpr int BLOCK = (PW + BW) * kWave;
    constexpr uint64_t TILE_F4 = (uint64_t)ROW * LU;
    constexpr uint32_t SLICE = papr_sweep_slice_floats(4);
    constexpr uint32_t SLOT = 8 * kWave;                 // floats per slot: 8 powers per lane
    constexpr uint32_t SPT = LU / 4;                     // slots a loader fills per tile
    static_assert(LU % 4 == 0 && DEPTH % NB == 0 && DEPTH >= NB, "slot bookkeeping");
    __shared__ unsigned long long seg_fill;
    __shared__ uint32_t wave_fill[BW];
    __shared__ uint32_t slot_filled[PW][DEPTH], slot_consumed[PW][DEPTH];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t nbins = P.nkeys + 2;  // + the NaN trash bin
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    uint32_t *hist = tab + P.table_words;
    float *slices = reinterpret_cast<float *>(hist + ((P.copies * nbins + 3u) & ~3u));
    float *ring = slices + BW * SLICE;                   // PW x DEPTH slots (16-byte aligned: everything above is)

    const uint32_t t = threadIdx.x;
    for (uint32_t k = t; k < P.table_words; k += BLOCK)
        tab[k] = table[k];
    for (uint32_t k = t; k < P.copies * nbins; k += BLOCK)
        hist[k] = 0;
    if (t == 0)
        seg_fill = seg_counts[blockIdx.x];
    if (t < BW)
        wave_fill[t] = 0;
    if (t < PW * DEPTH) {
        (&slot_filled[0][0])[t] = 0;
        (&slot_consumed[0][0])[t] = 0;
    }
    __syncthreads();

    const uint32_t lane = t & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t / kWave);
    const bool loader = wave < PW;
    const uint32_t bidx = loader ? 0u : wave - PW;       // binner number
    const uint32_t feed = loader ? wave : bidx / NB;     // the loader this wave is, or is fed by
    float *my_ring = ring + feed * (DEPTH * SLOT);
    const TileWalk w = tile_walk(blockIdx.x, gridDim.x, ntiles, map);

    double sum = 0.0;
    TileTrack tr = {{0.f, 0.f, 0.f, 0.f, 0.f}, {0, 0, 0, 0, 0}};
    if (loader) {
        __builtin_amdgcn_s_setprio(2);  // the load stream first
        const float4 *p = data + w.first * TILE_F4 + t;
        const uint64_t step = w.stride * TILE_F4;
        float4 cur[LU], nxt[LU];
        if (w.count)
            load_tile<ROW, LU, true>(cur, p);
        for (uint32_t it = 0; it < w.count; it++) {
            p += step;
            if (it + 1 < w.count)
                load_tile<ROW, LU, true>(nxt, p);
            float pw[2 * LU];
#pragma unroll
            for (int u = 0; u < LU; u++) {
                pw[2 * u] = power_of(cur[u].x, cur[u].y);
                pw[2 * u + 1] = power_of(cur[u].z, cur[u].w);
            }
#pragma unroll
            for (int u = 0; u < 2 * LU; u++)
                sum += (double)pw[u];  // same order as papr_stats_kernel
            track_tile<LU>(tr, cur, pw, it);
#pragma unroll
            for (uint32_t h = 0; h < SPT; h++) {
                const uint32_t q = it * SPT + h, d = q % DEPTH;
                // the slot's previous content (sequence number q - DEPTH) must have been taken
                lds_u32 *taken = (lds_u32 *)&slot_consumed[feed][d];
                while ((int32_t)(__builtin_amdgcn_readfirstlane(*(volatile lds_u32 *)taken) + DEPTH - (q + 1)) < 0)
                    __builtin_amdgcn_s_sleep(1);
                float *slot = my_ring + d * SLOT;
                *reinterpret_cast<f32x4s *>(slot + 4 * lane) = f32x4s{pw[8 * h], pw[8 * h + 1], pw[8 * h + 2], pw[8 * h + 3]};
                *reinterpret_cast<f32x4s *>(slot + 4 * kWave + 4 * lane) =
                    f32x4s{pw[8 * h + 4], pw[8 * h + 5], pw[8 * h + 6], pw[8 * h + 7]};
                __builtin_amdgcn_wave_barrier();
                if (lane == 0)  // (behind the data: LDS operations of a wave execute in order)
                    *(volatile lds_u32 *)(lds_u32 *)&slot_filled[feed][d] = q + 1;
            }
#pragma unroll
            for (int u = 0; u < LU; u++)
                cur[u] = nxt[u];
        }
    } else {
        const uint2 *lut_biased = reinterpret_cast<const uint2 *>(tab) - ((int32_t)P.cell_lo - 1);
        uint32_t *my = hist + (bidx % P.copies) * nbins;
        WaveStash ws{0u, slices + bidx * SLICE, &wave_fill[bidx], stash + (uint64_t)blockIdx.x * seg_cap, &seg_fill,
                     seg_cap, tab, P.table_words, PAPR_LUT2_NEVER, seg_fill, gave_up, nullptr};
        const int32_t cell_last = (int32_t)(P.cell_lo + P.ncells);
        int32_t cell_first;  // pinned in a VGPR for the whole kernel (v_med3 takes one scalar operand)
        asm volatile("v_mov_b32 %0, %1" : "=v"(cell_first) : "s"((int32_t)P.cell_lo - 1));
        const uint32_t shift = P.shift, offmask = (1u << shift) - 1u;
        auto bin_of = [&](float v) -> uint32_t {
            const int32_t cell = __float_as_int(v) >> shift;  // arithmetic shift: sign-bit patterns go below
            const uint2 e = lut_biased[clamp_cell(cell, cell_first, cell_last)];
            const uint32_t off = __float_as_uint(v) & offmask;
            return (e.x >> PAPR_LUT2_OFF_BITS) + (off >= (e.x & PAPR_LUT2_NEVER) ? 1u : 0u) + (off >= e.y ? 1u : 0u);
        };
        auto count_and_stash = [&](float v, uint32_t k) {
            if (k)
                atomicAdd(&my[k], 1u);
            ws.put(v, (k & 1u) != 0u);
        };
        const uint32_t nslots = w.count * SPT;
        for (uint32_t q = bidx % NB; q < nslots; q += NB) {
            const uint32_t d = q % DEPTH;
            lds_u32 *filled = (lds_u32 *)&slot_filled[feed][d];
            while ((int32_t)(__builtin_amdgcn_readfirstlane(*(volatile lds_u32 *)filled) - (q + 1)) < 0)
                __builtin_amdgcn_s_sleep(1);
            const float *slot = my_ring + d * SLOT;
            const f32x4s a = *reinterpret_cast<const f32x4s *>(slot + 4 * lane);
            const f32x4s b = *reinterpret_cast<const f32x4s *>(slot + 4 * kWave + 4 * lane);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0)  // (behind the reads: they have executed when this does)
                *(volatile lds_u32 *)(lds_u32 *)&slot_consumed[feed][d] = q + 1;
            const float pw[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            uint32_t k[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
                k[u] = bin_of(pw[u]);
#pragma unroll
            for (int u = 0; u < 8; u++)
                count_and_stash(pw[u], k[u]);
            ws.spill_if_above(SLICE - 8 * kWave, (q / SPT + 1) * (uint32_t)(2 * TILE_F4));
        }
        // sub-tile remainder of the shard: binned here (its pass-1 part is folded in by papr_stats_finalize)
        if (blockIdx.x == gridDim.x - 1) {
            const uint32_t tc = t - ROW;
            for (uint32_t k0 = 0; k0 < tail_samples; k0 += BW * kWave) {  // wave-uniform trip count
                const bool valid = k0 + tc < tail_samples;
                const float2 x = valid ? tail[k0 + tc] : make_float2(0.f, 0.f);
                const float v = power_of(x.x, x.y);
                count_and_stash(v, valid ? bin_of(v) : 0u);
                ws.spill_if_above(SLICE - kWave, ~0u);
            }
        }
        ws.spill_if_above(0, ~0u);
    }

    sweep_record<BLOCK, ROW, LU>(sum, tr, w, data, base_index, loader ? t : 0u, out);
    hist_flush<BLOCK>(hist, nbins, P.copies, ghist);  // (starts with a barrier: every binner has spilled)
    if (t == 0) {
        seg_counts[blockIdx.x] = seg_fill;
        seg_real[blockIdx.x] = seg_fill;
    }
}

// =============================================================================
// 3b. the sweep, second generation: wave-private segments, compact LUT, ring stash, optional exact-sum pairs
// =============================================================================
// What changed against papr_sweep_kernel, and why (measurements: DESIGN.md section 7):
//  * the unit of work is a WAVE-private segment of 64 * U float4 (U = 8: 1024 samples, 8 KiB), not a workgroup-wide
//    tile: one persistent workgroup per CU stages the LUT and zeroes / flushes its histogram ONCE, waves never meet
//    at a barrier inside the loop, and a wave has 8 KiB in flight instead of 4
//  * compact LUT (papr_kernels.h): up to two band edges per cell, so cells are as wide as the spacing of the
//    THRESHOLDS allows (2^17 patterns for the 0.1 dB table: 8 KiB of LDS instead of 32-40 KiB) whatever the band
//    width — more lanes hit the same entry (broadcast instead of bank conflict), and the band can shrink with the
//    quality of the estimate
//  * the stash leaves a wave through a RING in LDS in fixed spills of 256 floats written as one 16-byte store per
//    lane at a 1 KiB-aligned position (a write-through dword store is one fabric write each: ~6x the time per byte
//    of a 16-byte one); a partial spill (end of the launch) is padded to 16 bytes with quiet NaNs, which the
//    recount ignores
//  * EXACT: the same read also produces the per-segment rounding functions (D0, D1) of papr_exact.hip for a
//    SPECULATED binade of the running sum (from the estimate's per-group sums), and D0 — exact or not — is the
//    segment's sum, from which the true prefix sums are formed afterwards; segments whose speculated binade turns
//    out wrong are redone by papr_exact_redo_kernel (a fraction of a per cent), so that the bit-exact sequential
//    sum costs one read as well.  Lanes own 16 CONSECUTIVE samples there (XOR-swizzled LDS transpose), which is
//    also the order everything else is then computed in.

namespace {


// BALLOT: the ring is this wave's alone, so its head can live in a scalar register and slots be handed out by
// ballot + mbcnt — no returning LDS atomic (which hipcc expands into a dozen instructions) per in-band sample
template <uint32_t RING, int WT, bool BALLOT = false, bool NOBR = false>
struct StashRing {
    uint32_t nhead = 0;               // BALLOT: slots handed out so far (wave-uniform)
    float *ring;                      // this wave's ring in LDS (RING floats, 16-byte aligned)
    uint32_t *head;                   // LDS: slots reserved by this wave's lanes so far
    uint32_t tail;                    // wave-uniform: slots already written out (multiple of the spill size)
    float *__restrict__ seg;          // this workgroup's stash segment in HBM
    unsigned long long *seg_fill;     // LDS: floats reserved in the segment (may run past seg_cap: overflow)
    unsigned long long *seg_real;     // LDS: powers stashed (without padding)
    uint64_t seg_cap;
    uint32_t *tab;                    // LDS: the LUT (sweep_give_up)
    uint32_t table_words;
    unsigned long long seg_start;     // the segment's length when this launch began
    unsigned long long *gave_up;      // device counter of give-ups
    uint32_t folded;                  // samples this workgroup has folded in this launch, about (kept by the kernel)
    bool give_up;                     // wave-uniform: this wave found the bands too full (see sweep_give_up)
    uint32_t ring_addr = 0, trash_addr = 0;  // NOBR: LDS byte address of the ring (wave-uniform) / of this lane's trash word

    __device__ __forceinline__ void give_up_if_asked()
    {
        if (give_up) {
            sweep_give_up(tab, table_words, PAPR_LUT2_NEVER, seg_fill, seg_cap, gave_up);
            give_up = false;
        }
    }

    // One reservation per lane for ALL its in-band powers of a batch (one LDS round trip per batch; a returning
    // atomic per sample serialises up to 2 * BATCH of them behind s_waitcnt lgkmcnt(0)), then plain LDS writes.
    // Returns the ring's head as this lane saw it issued AFTER its own reservation (see pending_from).
    template <int N>
    __device__ __forceinline__ uint32_t put_batch(const float (&pw)[N], const uint32_t (&k)[N])
    {
        uint32_t cnt = 0;
#pragma unroll
        for (int u = 0; u < N; u++)
            cnt += k[u] & 1u;
        uint32_t slot = 0;
        if (cnt)
            slot = atomicAdd(head, cnt);
        // the wave's head after every lane's reservation: LDS operations of one wave complete in issue order, so
        // this read (issued behind the atomics, consumed only after the ring writes) sees them all
        const uint32_t seen = *(volatile lds_u32 *)(lds_u32 *)head;
#pragma unroll
        for (int u = 0; u < N; u++) {
            if (k[u] & 1u) {
                ring[slot & (RING - 1)] = pw[u];
                slot++;
            }
        }
        return seen;
    }
    __device__ __forceinline__ void put(float pw, bool take)
    {
        if constexpr (BALLOT && NOBR) {
            // no branch, no exec-masked region (as WaveStashT's): what is not in band goes to the lane's trash word
            const unsigned long long m = __ballot(take);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const uint32_t at = (nhead + rank) & (RING - 1);
            uint32_t a_slot, a;
            asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a_slot) : "v"(at), "s"(__builtin_amdgcn_readfirstlane(ring_addr)));
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a) : "v"(trash_addr), "v"(a_slot), "s"(m));
            *(__attribute__((address_space(3))) float *)(uintptr_t)a = pw;
            nhead += (uint32_t)__popcll(m);
        } else if constexpr (BALLOT) {
            const unsigned long long m = __ballot(take);
            if (m) {  // (wave-uniform)
                const uint32_t at = nhead + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (take)
                    ring[at & (RING - 1)] = pw;
                nhead += (uint32_t)__popcll(m);
            }
        } else {
            if (take)
                ring[atomicAdd(head, 1u) & (RING - 1)] = pw;
        }
    }
    // write ring[tail, tail + n) to the segment; n <= 256, tail is a multiple of 256
    __device__ __forceinline__ void chunk(uint32_t n)
    {
        const uint32_t lane = threadIdx.x & (kWave - 1);
        const uint32_t n4 = (n + 3u) & ~3u;
        unsigned long long pos = 0;
        if (lane == 0) {
            pos = atomicAdd(seg_fill, (unsigned long long)n4);
            atomicAdd(seg_real, (unsigned long long)n);
        }
        pos = uniform_u64(pos);  // lane 0's value, in scalar registers
        if (4 * lane < n4) {
            f32x4s v = *reinterpret_cast<const f32x4s *>(ring + (tail & (RING - 1)) + 4 * lane);
            const float pad = __uint_as_float(PAPR_STASH_PAD_BITS);
            v.y = 4 * lane + 1 < n ? v.y : pad;
            v.z = 4 * lane + 2 < n ? v.z : pad;
            v.w = 4 * lane + 3 < n ? v.w : pad;
            if (pos + 4 * lane + 4 <= seg_cap)
                store16<WT>(seg + pos + 4 * lane, v);
        }
        tail += n;
        // (32-bit on purpose: a workgroup folds < 2^32 samples per launch, and the 64-bit compare-with-literal forms
        // cost registers this kernel does not have)
        const uint32_t got = (uint32_t)(pos - seg_start) + n4;
        if (pos <= seg_cap && (pos + n4 > seg_cap || (got >= kGiveUpMin && got > folded / 2)))
            give_up = true;  // (pos > seg_cap: someone already did); acted on between two segments
    }
    __device__ __forceinline__ uint32_t pending()
    {
        __builtin_amdgcn_wave_barrier();  // LDS is in-order per wave; this pins the compiler's order too
        if constexpr (BALLOT)
            return nhead - tail;
        else
            return __builtin_amdgcn_readfirstlane(*(volatile lds_u32 *)(lds_u32 *)head) - tail;
    }
    // spill whole 256-float chunks; `seen` = what put_batch returned (any lane's value is the wave's head)
    __device__ __forceinline__ void spill_from(uint32_t seen)
    {
        uint32_t n = __builtin_amdgcn_readfirstlane(seen) - tail;
        if (n >= PAPR_SWEEP2_SPILL) {
            __builtin_amdgcn_wave_barrier();
            do {
                chunk(PAPR_SWEEP2_SPILL);
                n -= PAPR_SWEEP2_SPILL;
            } while (n >= PAPR_SWEEP2_SPILL);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __device__ __forceinline__ void spill_full()
    {
        uint32_t n = pending();
        while (n >= PAPR_SWEEP2_SPILL) {
            chunk(PAPR_SWEEP2_SPILL);
            n -= PAPR_SWEEP2_SPILL;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void flush()
    {
        spill_full();
        const uint32_t n = pending();
        if (n)
            chunk(n);
        tail = (tail + PAPR_SWEEP2_SPILL - 1) & ~(PAPR_SWEEP2_SPILL - 1);  // (keeps the ring reads 16-byte aligned)
        __builtin_amdgcn_wave_barrier();
        if constexpr (BALLOT)
            nhead = tail;
        else if ((threadIdx.x & (kWave - 1)) == 0)
            *(volatile lds_u32 *)(lds_u32 *)head = tail;
        __builtin_amdgcn_wave_barrier();
    }
};

// running per-segment extremes as integer bit patterns (see track_tile)
struct SegMax {
    uint32_t pk, rn, in;
    int32_t rp, ip;
};

__device__ __forceinline__ void segmax_fold(SegMax &m, const float4 &x, float p0, float p1)
{
    m.pk = umax3(m.pk, __float_as_uint(p0), __float_as_uint(p1));
    m.rp = imax3(m.rp, __float_as_int(x.x), __float_as_int(x.z));
    m.rn = umax3(m.rn, __float_as_uint(x.x), __float_as_uint(x.z));
    m.ip = imax3(m.ip, __float_as_int(x.y), __float_as_int(x.w));
    m.in = umax3(m.in, __float_as_uint(x.y), __float_as_uint(x.w));
}

__device__ __forceinline__ void segmax_commit(TileTrack &tr, const SegMax &m, uint32_t it)
{
    const float c[5] = {__uint_as_float(m.pk), __int_as_float(m.rp), __uint_as_float(m.rn), __int_as_float(m.ip),
                        __uint_as_float(m.in)};
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const bool win = (k == 2 || k == 4) ? (c[k] < tr.best[k]) : (c[k] > tr.best[k]);  // strict: first segment wins
        tr.best[k] = win ? c[k] : tr.best[k];
        tr.iter[k] = win ? it : tr.iter[k];
    }
}

// ---- exact-sum pairs (the algebra is papr_exact.hip's; restated here because both files keep their helpers
// in anonymous namespaces) ----
struct Pair2 {
    double d0, d1;
};
__device__ __forceinline__ double pow2_f64(int e) { return __longlong_as_double((long long)(e + 1023) << 52); }
__device__ __forceinline__ Pair2 compose2(Pair2 f, Pair2 g, double m0)
{
    const int q0 = __double2loint(m0 + f.d0) & 1;
    const int q1 = (__double2loint(m0 + f.d1) & 1) ^ 1;
    Pair2 h;
    h.d0 = f.d0 + (q0 ? g.d1 : g.d0);
    h.d1 = f.d1 + (q1 ? g.d1 : g.d0);
    return h;
}
template <int SHIFT>
__device__ __forceinline__ double row_shl2(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x100 + SHIFT, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x100 + SHIFT, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int SHIFT>
__device__ __forceinline__ Pair2 compose2_row(Pair2 f, double m0)
{
    Pair2 g;
    g.d0 = row_shl2<SHIFT>(f.d0);
    g.d1 = row_shl2<SHIFT>(f.d1);
    return compose2(f, g, m0);
}
__device__ __forceinline__ Pair2 wave_compose2(Pair2 f, double m0)  // ordered merge, lane order = file order; result in lane 0
{
    f = compose2_row<1>(f, m0);
    f = compose2_row<2>(f, m0);
    f = compose2_row<4>(f, m0);
    f = compose2_row<8>(f, m0);
#pragma unroll
    for (int off = 16; off < kWave; off <<= 1) {
        Pair2 g;
        g.d0 = __shfl_down(f.d0, off, kWave);
        g.d1 = __shfl_down(f.d1, off, kWave);
        f = compose2(f, g, m0);
    }
    return f;
}
// slot of float4 w of run r in the wave's transpose buffer (conflict-free both ways; papr_exact.hip)
__device__ __forceinline__ int xpose_slot(int run, int w) { return run * 8 + (w ^ ((run >> 1) & 7)); }

// Workgroup record of the v2 kernel: as sweep_record, for wave-private segments.  LANE_MAJOR: lane l owns float4
// l*U .. l*U+U-1 of its segment (exact mode); otherwise float4 u*64 + l.
template <int WAVES, int U, bool LANE_MAJOR>
__device__ __forceinline__ void sweep2_record(double sum, const TileTrack &tr, uint64_t seg0, uint64_t seg_stride,
                                              const float4 *__restrict__ data, uint64_t base_index,
                                              papr_partial *__restrict__ out)
{
    constexpr uint64_t SEG_F4 = 64ull * U;
    const uint32_t t = threadIdx.x;
    __shared__ double sh_sum[WAVES];
    __shared__ float sh_val[WAVES][5];
    __shared__ unsigned long long sh_idx[WAVES][5];
    const int lane = t & (kWave - 1), wave = t / kWave;
    float wv[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        float v = tr.best[k];
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
            const float o = __shfl_down(v, off, kWave);
            v = (k == 2 || k == 4) ? (o < v ? o : v) : (o > v ? o : v);
        }
        wv[k] = v;
    }
    const double wsum = wave_reduce_sum(sum);
    if (lane == 0) {
        sh_sum[wave] = wsum;
#pragma unroll
        for (int k = 0; k < 5; k++)
            sh_val[wave][k] = wv[k];
    }
    __syncthreads();
    float win[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        float v = sh_val[0][k];
        for (int wq = 1; wq < WAVES; wq++) {
            const float o = sh_val[wq][k];
            v = (k == 2 || k == 4) ? (o < v ? o : v) : (o > v ? o : v);
        }
        win[k] = v;
    }
    unsigned long long idx[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        idx[k] = ~0ull;
        if (win[k] != 0.f && tr.best[k] == win[k]) {  // a tracker that never fired keeps value 0 and reports index 0
            const uint64_t f4_0 = (seg0 + (uint64_t)tr.iter[k] * seg_stride) * SEG_F4;
            for (int u = U - 1; u >= 0; u--) {  // last match written last = first slot wins
                const uint64_t f4 = f4_0 + (LANE_MAJOR ? (uint64_t)lane * U + u : (uint64_t)u * kWave + lane);
                const float4 x = data[f4];
                const float a = k == 0 ? power_of(x.x, x.y) : (k <= 2 ? x.x : x.y);
                const float b = k == 0 ? power_of(x.z, x.w) : (k <= 2 ? x.z : x.w);
                if (b == win[k])
                    idx[k] = base_index + 2 * f4 + 1;
                if (a == win[k])
                    idx[k] = base_index + 2 * f4;
            }
        }
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_down(idx[k], off, kWave);
            idx[k] = o < idx[k] ? o : idx[k];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; k++)
            sh_idx[wave][k] = idx[k];
    }
    __syncthreads();
    if (t == 0) {
        papr_partial q;
        q.sum = sh_sum[0];
        for (int wq = 1; wq < WAVES; wq++)  // fixed order => deterministic sum
            q.sum += sh_sum[wq];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            unsigned long long best_idx = sh_idx[0][k];
            for (int wq = 1; wq < WAVES; wq++)
                best_idx = sh_idx[wq][k] < best_idx ? sh_idx[wq][k] : best_idx;
            q.val[k] = win[k];
            q.idx[k] = win[k] != 0.f ? best_idx : 0;
        }
        q.pad = 0;
        out[blockIdx.x] = q;
    }
}

}  // namespace

template <int WAVES, int U, int PIPE, bool EXACT, int WTB>
__global__ __launch_bounds__(WAVES *kWave) void papr_sweep2_kernel(const papr_sweep2_params p)
{
    static_assert(!EXACT || U == 8, "exact-sum segments are 1024 samples");
    constexpr bool BATCHED = (WTB & 4) != 0;  // one stash reservation per lane per batch instead of one per in-band sample
    constexpr bool LEAN_SUM = (WTB & 8) != 0; // exact mode: the lane's sum is x0 - m0 (no separate accurate accumulation)
    constexpr int BLOCK = WAVES * kWave;
    constexpr uint64_t SEG_F4 = 64ull * U;
    constexpr bool WIDE = (WTB & 32) != 0;           // exact mode: batches of 4 float4 (ring of 1024) like the plain form
    constexpr uint32_t RING = (EXACT && !WIDE) ? 512u : 1024u;  // >= 255 + 64 * (samples per lane between two ring checks)
    constexpr int BATCH = (EXACT && !WIDE) ? 2 : 4;  // float4 per lane folded between two ring checks
    __shared__ unsigned long long seg_fill, seg_real_sh;
    __shared__ uint32_t ring_head[WAVES];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const papr_ccdf_params P = uniform_params(p.Pdev, p.P);
    const uint32_t nbins = P.nkeys + 1;
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    uint32_t *hist = tab + P.table_words;                                         // table_words is a multiple of 4
    float *rings = reinterpret_cast<float *>(hist + ((P.copies * nbins + 3u) & ~3u));
    float4 *xpose = reinterpret_cast<float4 *>(rings + WAVES * RING);             // EXACT: WAVES x 8 KiB

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
    if (t < WAVES)
        ring_head[t] = 0;
    __syncthreads();

    const uint2 *lut_biased = reinterpret_cast<const uint2 *>(tab) - ((int32_t)P.cell_lo - 1);
    uint32_t *my = hist + (wave % P.copies) * nbins;
    constexpr bool BALLOT = (WTB & 16) != 0;  // stash slots by ballot + mbcnt instead of a returning LDS atomic
    // ... and no branch / exec-masked region per sample (bin 0 and out-of-band powers go to a trash word).  Measured
    // (variants 28 / 29 / 31, `make MEASURE=1`): the exact kernel is VALU-bound — 38 VALU per sample, the two double-precision chains at
    // half rate — and executing the stash code for every sample costs more than its branches: 1.916 against 1.851 ms
    // (-g: 2.057 against 2.090)
    constexpr bool NOBR_HIST = (WTB & 64) != 0, NOBR = (WTB & 128) != 0;  // (histogram / stash without a branch, separately)
    static_assert(!(BALLOT && BATCHED), "the batched reservation is an LDS atomic");
    static_assert(!(NOBR || NOBR_HIST) || (BALLOT && EXACT), "the trash word is the lane's first transposition slot");
    StashRing<RING, (WTB & 3), BALLOT, NOBR> ws{0u, rings + wave * RING,
                           &ring_head[wave],
                           0u,
                           p.stash + (uint64_t)blockIdx.x * p.seg_cap,
                           &seg_fill,
                           &seg_real_sh,
                           p.seg_cap,
                           tab,
                           P.table_words,
                           seg_fill,
                           p.gave_up,
                           0,
                           false};
    const int32_t cell_last = (int32_t)(P.cell_lo + P.ncells);
    int32_t cell_first;  // pinned in a VGPR (v_med3 takes one scalar operand)
    asm volatile("v_mov_b32 %0, %1" : "=v"(cell_first) : "s"((int32_t)P.cell_lo - 1));
    const uint32_t shift = P.shift;
    const uint32_t offmask = (1u << shift) - 1u;

    auto bin_of = [&](float pw) -> uint32_t {
        const int32_t cell = __float_as_int(pw) >> shift;  // arithmetic shift: sign-bit patterns go below the table
        const uint2 e = lut_biased[clamp_cell(cell, cell_first, cell_last)];
        const uint32_t off = __float_as_uint(pw) & offmask;
        return (e.x >> PAPR_LUT2_OFF_BITS) + (off >= (e.x & PAPR_LUT2_NEVER) ? 1u : 0u) + (off >= e.y ? 1u : 0u);
    };
    if constexpr (NOBR || NOBR_HIST) {
        // the lane's own first transposition slot: read (by this lane only) before anything of the segment is folded
        ws.ring_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u32 *)ws.ring);
        ws.trash_addr = (uint32_t)(uintptr_t)(lds_u32 *)(xpose + wave * (kWave * 8) + xpose_slot((int)lane, 0));
    }
    auto count_and_stash = [&](float pw, uint32_t k) {
        if constexpr (NOBR_HIST) {
            const unsigned long long nz = __ballot(k != 0u);
            const uint32_t a_bin = (uint32_t)(uintptr_t)(lds_u32 *)&my[k];
            uint32_t a;
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a) : "v"(ws.trash_addr), "v"(a_bin), "s"(nz));
            (void)__hip_atomic_fetch_add((lds_u32 *)(uintptr_t)a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (k) {
            atomicAdd(&my[k], 1u);
        }
        ws.put(pw, (k & 1u) != 0u);
    };

    const float4 *data = reinterpret_cast<const float4 *>(p.data);
    const uint64_t seg_stride = (uint64_t)gridDim.x * WAVES;
    const uint64_t seg0 = (uint64_t)blockIdx.x * WAVES + wave;
    const uint32_t count = p.nsegs > seg0 ? (uint32_t)((p.nsegs - seg0 + seg_stride - 1) / seg_stride) : 0u;

    double sum = 0.0;
    TileTrack tr = {{0.f, 0.f, 0.f, 0.f, 0.f}, {0, 0, 0, 0, 0}};

    // fold BATCH float4 (2 * BATCH samples) of this lane: powers, extremes, bins, stash
    auto fold_batch = [&](const float4(&x)[BATCH], SegMax &m, double &x0, double &x1) {
        float pw[2 * BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            // (power_of with the squares as one packed multiplication and the additions written out: see papr_sweep_kernel)
            typedef float f32x2v __attribute__((ext_vector_type(2)));
            const f32x2v a = {x[u].x, x[u].y}, b = {x[u].z, x[u].w};
            const f32x2v aa = a * a, bb = b * b;
            asm("v_add_f32 %0, %1, %2" : "=v"(pw[2 * u]) : "v"(aa.x), "v"(aa.y));
            asm("v_add_f32 %0, %1, %2" : "=v"(pw[2 * u + 1]) : "v"(bb.x), "v"(bb.y));
            segmax_fold(m, x[u], pw[2 * u], pw[2 * u + 1]);
        }
#pragma unroll
        for (int u = 0; u < 2 * BATCH; u++) {
            const double v = (double)pw[u];
            if constexpr (!(EXACT && LEAN_SUM))
                sum += v;  // the accurate per-lane sum (as papr_stats_kernel), also in exact mode
            if constexpr (EXACT) {
                x0 += v;  // the reference's additions themselves, from the two canonical entry states
                x1 += v;
            }
        }
        uint32_t k[2 * BATCH];
#pragma unroll
        for (int u = 0; u < 2 * BATCH; u++)
            k[u] = bin_of(pw[u]);  // the batch's LUT reads in flight together
        if constexpr (BATCHED) {
            // one stash reservation per lane per batch (one LDS round trip instead of up to 2 * BATCH)
#pragma unroll
            for (int u = 0; u < 2 * BATCH; u++)
                if (k[u])
                    atomicAdd(&my[k[u]], 1u);
            ws.spill_from(ws.put_batch(pw, k));
        } else {
#pragma unroll
            for (int u = 0; u < 2 * BATCH; u++)
                count_and_stash(pw[u], k[u]);
            ws.spill_full();
        }
    };

    auto load_seg = [&](float4(&x)[U], uint64_t seg) {
        // the segment's base is wave-uniform: kept in scalar registers (scalar base + 32-bit lane offset addressing),
        // not as a 64-bit pointer per lane
        const unsigned long long base = uniform_u64((unsigned long long)(data + seg * SEG_F4));
#pragma unroll
        for (int u = 0; u < U; u++)
            x[u] = load16_nt_at(base, lane + u * kWave);
    };

    if constexpr (EXACT) {
        float4 *mine = xpose + wave * (kWave * 8);
        const int32_t *__restrict__ tile_E = p.tile_E_spec;
        double2 *__restrict__ seg_D = reinterpret_cast<double2 *>(p.seg_D);
        float4 x[U];
        if (count)
            load_seg(x, seg0);
        for (uint32_t it = 0; it < count; it++) {
            const uint64_t seg = seg0 + (uint64_t)it * seg_stride;
            ws.folded = (it + 1) * (uint32_t)(WAVES * 2 * SEG_F4);
            const int E = tile_E[(p.seg_offset + seg) >> 1];
#pragma unroll
            for (int r = 0; r < U; r++) {
                const int f = r * kWave + (int)lane;  // float4 slot within the segment, file order
                mine[xpose_slot(f >> 3, f & 7)] = x[r];
            }
            // the registers are free again: the next segment's loads fly while this one is folded out of LDS
            // (same wave wrote and reads the buffer: LDS operations of one wave complete in order)
            if (it + 1 < count)
                load_seg(x, seg + seg_stride);
            const bool valid = E != PAPR_EXACT_AMBIG;
            const double m0 = valid ? pow2_f64(E) : 0.0, m1 = valid ? m0 + pow2_f64(E - 52) : 0.0;
            double x0 = m0, x1 = m1;
            SegMax m = {0u, 0u, 0u, INT32_MIN, INT32_MIN};
#pragma unroll
            for (int b = 0; b < U / BATCH; b++) {
                float4 y[BATCH];
#pragma unroll
                for (int j = 0; j < BATCH; j++)
                    y[j] = mine[xpose_slot((int)lane, b * BATCH + j)];
                fold_batch(y, m, x0, x1);
            }
            segmax_commit(tr, m, it);
            Pair2 f;
            f.d0 = x0 - m0;  // exact: multiples of the ulp inside the binade (a plain sum when no binade was given)
            f.d1 = x1 - m1;
            if constexpr (LEAN_SUM)
                sum += f.d0;
            f = wave_compose2(f, m0);
            if (lane == 0)
                seg_D[p.seg_offset + seg] = make_double2(f.d0, f.d1);
            ws.give_up_if_asked();
        }
    } else {
        double none0 = 0.0, none1 = 0.0;
        auto fold_seg = [&](const float4(&x)[U], uint32_t it) {
            ws.folded = (it + 1) * (uint32_t)(WAVES * 2 * SEG_F4);
            SegMax m = {0u, 0u, 0u, INT32_MIN, INT32_MIN};
#pragma unroll
            for (int b = 0; b < U / BATCH; b++) {
                float4 y[BATCH];
#pragma unroll
                for (int j = 0; j < BATCH; j++)
                    y[j] = x[b * BATCH + j];
                fold_batch(y, m, none0, none1);
            }
            segmax_commit(tr, m, it);
            ws.give_up_if_asked();
        };
        if constexpr (PIPE == 1) {
            float4 cur[U], nxt[U];
            if (count)
                load_seg(cur, seg0);
            for (uint32_t it = 0; it < count; it++) {
                if (it + 1 < count)
                    load_seg(nxt, seg0 + (uint64_t)(it + 1) * seg_stride);
                fold_seg(cur, it);
#pragma unroll
                for (int u = 0; u < U; u++)
                    cur[u] = nxt[u];
            }
        } else {
            for (uint32_t it = 0; it < count; it++) {
                float4 x[U];
                load_seg(x, seg0 + (uint64_t)it * seg_stride);
                fold_seg(x, it);
            }
        }
    }
    // remainder of the launch: binned here (its pass-1 part is folded in by papr_stats_finalize, its exact-sum
    // part travels raw in the sum program)
    ws.folded = ~0u;
    if (blockIdx.x == gridDim.x - 1) {
        const float2 *tail = reinterpret_cast<const float2 *>(p.tail);
        for (uint32_t k0 = 0; k0 < p.tail_samples; k0 += BLOCK) {  // wave-uniform trip count
            const bool valid = k0 + t < p.tail_samples;
            const float2 x = valid ? tail[k0 + t] : make_float2(0.f, 0.f);
            const float pw = power_of(x.x, x.y);
            count_and_stash(pw, valid ? bin_of(pw) : 0u);
            ws.spill_full();
        }
    }
    ws.flush();

    sweep2_record<WAVES, U, EXACT>(sum, tr, seg0, seg_stride, data, p.base_index, p.out);
    hist_flush<BLOCK>(hist, nbins, P.copies, p.ghist);  // (starts with a barrier: every wave has flushed)
    if (t == 0) {
        p.seg_slots[blockIdx.x] = seg_fill;
        p.seg_real[blockIdx.x] = seg_real_sh;
    }
}

// =============================================================================
// 3c. the exact-sum sweep, third form: papr_sweep_kernel's per-sample code on wave-private segments
// =============================================================================
// papr_sweep2_kernel<EXACT> is instruction-bound: 37 VALU + 18 SALU per sample at three waves per SIMD (compact-table
// lookup 9, one exec-masked region each for the histogram and the stash, an ordered 64-lane composition of the
// segment's pairs worth 6).  This form keeps its data flow — a wave owns 1024-sample segments, lanes own 16 consecutive
// samples through the XOR-swizzled LDS transposition, the next segment's loads fly while this one is folded out of
// LDS — and replaces the rest:
//   * geometry and per-sample code of the default kernel (papr_sweep_kernel<512, 8, SMODE 43>): one persistent
//     workgroup of eight waves per CU, the ONE-edge-per-cell table (five VALU per lookup), histogram and stash without
//     a branch or an exec-masked region, 16-byte write-through spills out of a per-wave slice;
//   * the segment's pair without an ordered tree.  A lane's run maps the parity of the running sum onto itself:
//     pi(0) = lsb of x0, pi(1) = lsb of x1 (the sums it formed from the even and the odd canonical entry state), and its
//     increment is d0 or d1 = d0 + delta * ulp, delta in {-1, 0, +1}.  The parity with which the running sum ENTERS
//     lane l, for segment entry parity p, follows from the 64 one-bit maps alone — ballots and a dozen 64-bit scalar
//     operations: a prefix XOR over the lanes that swap, restarted behind every lane whose map is constant — and then
//     D[p] = sum(d0) + ulp * (#{delta = +1, entered odd} - #{delta = -1, entered odd}): ONE unordered wave sum (every
//     term is a multiple of the ulp: exact in any order) and four popcounts.
template <int WAVES, int SMODE, int HALF, bool EXACT>
__global__ __launch_bounds__(WAVES *kWave) void papr_sweep3_kernel(const papr_sweep2_params p)
{
    constexpr int U = 8;
    constexpr int BLOCK = WAVES * kWave;
    constexpr uint64_t SEG_F4 = 64ull * U;
    // HALF = float4 per lane between two spill checks: 4 (two batches per segment), or 8 — the whole segment as ONE batch:
    // its eight LDS reads, then its sixteen table lookups, are in flight together and the double-precision chains run
    // under them; with two waves per SIMD it is the LDS round trips per segment that decide how long a wave is stalled
    static_assert(HALF == 4 || HALF == 8, "batches per segment");
    static_assert((SMODE & 11) == 11, "branch-free ballot stash with 16-byte spills");
    __shared__ unsigned long long seg_fill, seg_real_sh;
    __shared__ uint32_t wave_fill[WAVES];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const papr_ccdf_params P = uniform_params(p.Pdev, p.P);
    const uint32_t nbins = P.nkeys + 2;  // + the NaN trash bin (as papr_sweep_kernel)
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    uint32_t *hist = tab + P.table_words;
    float *slices = reinterpret_cast<float *>(hist + ((P.copies * nbins + 3u) & ~3u));
    // the stash slices take what table and histogram leave of the launch's LDS (p.lds_bytes): a small table (the 1 dB
    // one) means rare, wide spills; at least PAPR_SWEEP3_SLICE_FLOATS (what the geometry function promises), at most 4096
    const uint32_t used_words = (uint32_t)(slices - reinterpret_cast<float *>(smem));
    constexpr uint32_t kXposeWords = EXACT ? WAVES * 2048u : 0u;
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
    if (t < WAVES)
        wave_fill[t] = 0;
    __syncthreads();

    const uint2 *lut_biased = reinterpret_cast<const uint2 *>(tab) - ((int32_t)P.cell_lo - 1);
    uint32_t *my = hist + (wave % P.copies) * nbins;
    WaveStashT<SMODE> ws{0u, slices + wave * SLICE, &wave_fill[wave], p.stash + (uint64_t)blockIdx.x * p.seg_cap,
                        &seg_fill, p.seg_cap, tab, P.table_words, 0u, seg_fill, p.gave_up, &seg_real_sh};
    ws.trash = SLICE - kWave + lane;
    ws.sbase = ws.sbytes = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u32 *)ws.buf);
    const int32_t cell_last = (int32_t)(P.cell_lo + P.ncells);
    int32_t cell_first;  // pinned in a VGPR (v_med3 takes one scalar operand)
    asm volatile("v_mov_b32 %0, %1" : "=v"(cell_first) : "s"((int32_t)P.cell_lo - 1));
    const uint32_t shift = P.shift;
    auto bin_of = [&](float pw) -> uint32_t {
        const int32_t cell = __float_as_int(pw) >> shift;  // arithmetic shift: sign-bit patterns go below the table
        const uint2 e = lut_biased[clamp_cell(cell, cell_first, cell_last)];
        return e.x + (__float_as_uint(pw) >= e.y ? 1u : 0u);
    };
    auto count_and_stash = [&](float pw, uint32_t k) {
        // bin 0 (below every band: not counted) adds to this lane's trash word instead of being skipped
        const unsigned long long nz = __ballot(k != 0u);
        const uint32_t a_bin = (uint32_t)(uintptr_t)(lds_u32 *)&my[k];
        const uint32_t a_trash = (uint32_t)(uintptr_t)(lds_u32 *)&ws.buf[ws.trash];
        uint32_t a;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a) : "v"(a_trash), "v"(a_bin), "s"(nz));
        (void)__hip_atomic_fetch_add((lds_u32 *)(uintptr_t)a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ws.put(pw, (k & 1u) != 0u);
    };

    const float4 *data = reinterpret_cast<const float4 *>(p.data);
    const uint64_t seg_stride = (uint64_t)gridDim.x * WAVES;
    const uint64_t seg0 = (uint64_t)blockIdx.x * WAVES + wave;
    const uint32_t count = p.nsegs > seg0 ? (uint32_t)((p.nsegs - seg0 + seg_stride - 1) / seg_stride) : 0u;
    auto load_seg = [&](float4(&x)[U], uint64_t seg) {
        const unsigned long long base = uniform_u64((unsigned long long)(data + seg * SEG_F4));  // wave-uniform: scalar base
#pragma unroll
        for (int u = 0; u < U; u++)
            x[u] = load16_nt_at(base, lane + u * kWave);
    };

    double sum = 0.0;
    TileTrack tr = {{0.f, 0.f, 0.f, 0.f, 0.f}, {0, 0, 0, 0, 0}};
    // one batch of HALF float4 per lane: powers, extremes, (exact mode: the two canonical sums), bins, stash
    auto fold_batch = [&](const float4(&y)[HALF], SegMax &m, double &x0, double &x1) {
        float pw[2 * HALF];
#pragma unroll
        for (int u = 0; u < HALF; u++) {
            // (power_of with the squares as one packed multiplication and the additions written out: see papr_sweep_kernel)
            typedef float f32x2v __attribute__((ext_vector_type(2)));
            const f32x2v a = {y[u].x, y[u].y}, b = {y[u].z, y[u].w};
            const f32x2v aa = a * a, bb = b * b;
            asm("v_add_f32 %0, %1, %2" : "=v"(pw[2 * u]) : "v"(aa.x), "v"(aa.y));
            asm("v_add_f32 %0, %1, %2" : "=v"(pw[2 * u + 1]) : "v"(bb.x), "v"(bb.y));
            segmax_fold(m, y[u], pw[2 * u], pw[2 * u + 1]);
        }
#pragma unroll
        for (int u = 0; u < 2 * HALF; u++) {
            const double v = (double)pw[u];
            x0 += v;  // exact mode: the reference's additions themselves (papr.c:104), from the two canonical entry states;
            if constexpr (EXACT)
                x1 += v;  // otherwise x0 is the lane's accurate sum (as papr_stats_kernel)
        }
        uint32_t k[2 * HALF];
#pragma unroll
        for (int u = 0; u < 2 * HALF; u++)
            k[u] = bin_of(pw[u]);  // the LUT reads in flight together
#pragma unroll
        for (int u = 0; u < 2 * HALF; u++)
            count_and_stash(pw[u], k[u]);
    };
    if constexpr (EXACT) {
    float4 *mine = xpose + wave * (kWave * 8);
    const int32_t *__restrict__ tile_E = p.tile_E_spec;
    double2 *__restrict__ seg_D = reinterpret_cast<double2 *>(p.seg_D);
    float4 x[U];
    if (count)
        load_seg(x, seg0);
    for (uint32_t it = 0; it < count; it++) {
        const uint64_t seg = seg0 + (uint64_t)it * seg_stride;
        const int E = __builtin_amdgcn_readfirstlane(tile_E[(p.seg_offset + seg) >> 1]);
#pragma unroll
        for (int r = 0; r < U; r++) {
            const int f = r * kWave + (int)lane;  // float4 slot within the segment, file order
            mine[xpose_slot(f >> 3, f & 7)] = x[r];
        }
        // the registers are free again: the next segment's loads fly while this one is folded out of LDS — in front of
        // any spill store of this segment, so that a spill never stands between the wave and its next data
        // (same wave wrote and reads the buffer: LDS operations of one wave complete in order)
        if (it + 1 < count)
            load_seg(x, seg + seg_stride);
        const bool valid = E != PAPR_EXACT_AMBIG;
        const double m0 = valid ? pow2_f64(E) : 0.0, ulp = valid ? pow2_f64(E - 52) : 0.0, m1 = m0 + ulp;
        double x0 = m0, x1 = m1;
        SegMax m = {0u, 0u, 0u, INT32_MIN, INT32_MIN};
#pragma unroll
        for (int h = 0; h < U / HALF; h++) {
            // room for the next 2 * HALF samples of every lane (and the trash words)?  Checked IN FRONT of the fold, behind
            // the next segment's loads: a spill's stores then have the fold's duration to drain before this wave waits
            // for memory again (vmcnt is in order and counts stores too)
            ws.spill_if_above(SLICE - (2 * HALF + 1) * kWave, (it + 1) * (uint32_t)(WAVES * 2 * SEG_F4));
            float4 y[HALF];
#pragma unroll
            for (int j = 0; j < HALF; j++)
                y[j] = mine[xpose_slot((int)lane, h * HALF + j)];
            fold_batch(y, m, x0, x1);
        }
        segmax_commit(tr, m, it);
        // ---- the segment's pair ----
        const double d0 = x0 - m0, d1 = x1 - m1;  // exact: multiples of the ulp inside the binade (plain sums when no binade was given)
        sum += d0;
        const unsigned long long A = __ballot((__double2loint(x0) & 1) != 0), B = __ballot((__double2loint(x1) & 1) != 0);
        const unsigned long long up = __ballot(d1 > d0), dn = __ballot(d1 < d0);
        const unsigned long long C = ~(A ^ B), N = A & ~B;  // lanes whose map is constant / swaps the parity
        unsigned long long px = N;                            // prefix XOR (inclusive), then exclusive
        px ^= px << 1;
        px ^= px << 2;
        px ^= px << 4;
        px ^= px << 8;
        px ^= px << 16;
        px ^= px << 32;
        px <<= 1;
        // fill forward from the constant lanes: the carry of an addition runs through the ones of ~C up to the next marker
        const unsigned long long Z = ~C, Y = (A ^ px) & C;
        const unsigned long long fwd = (Z + (Y << 1)) ^ Z;    // bit l: (A ^ px) of the last constant lane below l, 0 if none
        const unsigned long long has = (Z + (C << 1)) ^ Z;    // bit l: there is a constant lane below l
        const unsigned long long odd0 = px ^ fwd, odd1 = odd0 ^ ~has;  // lanes the sum enters with odd parity, for p = 0 / 1
        const int k0 = __popcll(up & odd0) - __popcll(dn & odd0), k1 = __popcll(up & odd1) - __popcll(dn & odd1);
        const double S = wave_sum_to_lane63(d0);
        if (lane == kWave - 1)
            seg_D[p.seg_offset + seg] = make_double2(S + (double)k0 * ulp, S + (double)k1 * ulp);
    }
    } else {
        // plain form: folded straight out of the registers (lane l owns float4 u * 64 + l), the next segment's loads
        // issued — and the spill check made — in front of the fold
        float4 cur[U], nxt[U];
        if (count)
            load_seg(cur, seg0);
        double none = 0.0;
        for (uint32_t it = 0; it < count; it++) {
            if (it + 1 < count)
                load_seg(nxt, seg0 + (uint64_t)(it + 1) * seg_stride);
            SegMax m = {0u, 0u, 0u, INT32_MIN, INT32_MIN};
#pragma unroll
            for (int h = 0; h < U / HALF; h++) {
                ws.spill_if_above(SLICE - (2 * HALF + 1) * kWave, (it + 1) * (uint32_t)(WAVES * 2 * SEG_F4));
                float4 y[HALF];
#pragma unroll
                for (int j = 0; j < HALF; j++)
                    y[j] = cur[h * HALF + j];
                fold_batch(y, m, sum, none);
            }
            segmax_commit(tr, m, it);
#pragma unroll
            for (int u = 0; u < U; u++)
                cur[u] = nxt[u];
        }
    }
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

    sweep2_record<WAVES, U, EXACT>(sum, tr, seg0, seg_stride, data, p.base_index, p.out);
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

// hipLaunchKernelGGL, or — when a timer is armed — the same dispatch with the timer's events bound to it
template <typename... Args, typename F = void (*)(Args...)>
static inline void launch_maybe_timed(F kernel, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, Args... args)
{
    if (tl_timer_armed) {
        tl_timer_armed = false;
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds_bytes, st, tl_timer.start, tl_timer.stop, 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds_bytes, st, args...);
    }
}

// ---- launch wrappers -------------------------------------------------------------------

void papr_launch_estimate(hipStream_t st, int blocks, const void *data, uint64_t ngroups, uint32_t ratio,
                          papr_partial *out, double *group_sums, double *block_sq)
{
    launch_maybe_timed(papr_estimate_kernel, dim3(blocks), dim3(PAPR_BLOCK), 0, st, (const float4 *)data, ngroups, ratio,
                       out, group_sums, block_sq);
}

// Geometry variants of the sweep (ids as in papr_kernels.hip's table).
#ifdef PAPR_MEASURE
#define PAPR_FOR_EACH_SWEEP_VARIANT(X) \
    X(0, 256, 8, 0) X(1, 256, 4, 1) X(2, 256, 8, 1) X(3, 512, 8, 0) X(4, 1024, 4, 0) X(6, 512, 4, 1) X(7, 256, 4, 0) \
    X(8, 1024, 4, 1) X(9, 1024, 2, 1) X(10, 512, 2, 1) X(11, 256, 2, 1) X(12, 1024, 2, 0) X(13, 512, 4, 0)              \
    X(14, 256, 4, 2) X(15, 512, 4, 2) X(16, 256, 8, 2) X(17, 1024, 4, 2)
#else  // the default (4) and one of every loop shape / workgroup size for the tests
#define PAPR_FOR_EACH_SWEEP_VARIANT(X) X(1, 256, 4, 1) X(4, 1024, 4, 0) X(8, 1024, 4, 1) X(13, 512, 4, 0) X(14, 256, 4, 2)
#endif

// the same kernel with the compact two-edges-per-cell table (small workgroups can afford a table of their own)
#ifdef PAPR_MEASURE
#define PAPR_FOR_EACH_SWEEP_LUT2_VARIANT(X)                                                                  \
    X(20, 256, 4, 1) X(21, 256, 4, 0) X(22, 512, 4, 1) X(23, 512, 4, 0) X(24, 1024, 4, 0) X(25, 256, 8, 1)    \
    X(26, 256, 2, 1) X(27, 512, 2, 1) X(28, 1024, 4, 1) X(29, 256, 8, 0)
#else
#define PAPR_FOR_EACH_SWEEP_LUT2_VARIANT(X) X(20, 256, 4, 1) X(24, 1024, 4, 0)
#endif

// papr_sweep_kernel with other stash forms: id, workgroup size, loads per lane, loop form, compact table,
// stash mode (bit 0: 16-byte spills, bit 1: ballot compaction, bit 2: [bin][copy] histogram sets, bit 3: no branch and
// no exec-masked region in the per-sample code, bit 4: plain instead of write-through spill stores, bit 5: twice the
// LDS slice per wave).  40 — 512 threads x 8 loads per lane, ONE persistent workgroup per CU, branch-free ballot stash,
// 16-byte spills out of a double slice — is the default: eight waves per CU with eight 16-byte loads each in flight is
// the shape a stripped kernel reads fastest in (tools/work_probe.hip); with only two waves per SIMD nothing hides a
// v_cmp -> s_and_saveexec round per sample or a spill's store latency, hence the branch-free code and the rarer, wider
// spills.  The others did not pay and are built by `make MEASURE=1` only.
#ifdef PAPR_MEASURE
#define PAPR_FOR_EACH_SWEEP_SP16_VARIANT(X) \
    X(40, 512, 8, 0, false, 43) X(100, 512, 8, 0, false, 10)                                                              \
    X(5, 1024, 4, 0, false, 1) X(18, 1024, 4, 0, true, 1) X(19, 512, 4, 0, false, 1) X(36, 1024, 4, 0, false, 2)          \
    X(37, 1024, 4, 0, false, 3) X(38, 1024, 4, 0, true, 2) X(39, 512, 4, 0, false, 2)                                  \
    X(80, 256, 8, 0, false, 2) X(81, 256, 8, 0, false, 3) X(82, 256, 8, 1, false, 2) X(83, 512, 8, 0, false, 2)       \
    X(84, 256, 8, 0, false, 6) X(85, 256, 8, 0, false, 14) X(86, 256, 8, 0, false, 10) X(87, 512, 8, 0, false, 14)         \
    X(88, 1024, 4, 0, false, 14) X(89, 256, 8, 1, false, 14) X(101, 512, 8, 1, false, 10)                                  \
    X(102, 512, 4, 0, false, 10) X(103, 1024, 8, 0, false, 10) X(104, 512, 8, 0, false, 11) X(105, 512, 8, 0, false, 26)   \
    X(106, 512, 8, 0, false, 27) X(107, 512, 8, 0, false, 42) X(109, 512, 8, 0, false, 59)   \
    X(110, 512, 8, 1, false, 11) X(111, 512, 8, 1, false, 43) X(112, 256, 8, 0, false, 43) X(113, 1024, 4, 0, false, 43) \
    X(114, 512, 8, 1, false, 107)
#else
#define PAPR_FOR_EACH_SWEEP_SP16_VARIANT(X) X(40, 512, 8, 0, false, 43) X(111, 512, 8, 1, false, 43) X(114, 512, 8, 1, false, 107)
#endif

// loader / binner split (papr_sweep_split_kernel): id, loader waves, binners per loader, loads per lane per tile, ring depth
// (measured slower than papr_sweep_kernel in every shape — DESIGN.md section 4b — so only `make MEASURE=1` builds it)
#ifdef PAPR_MEASURE
#define PAPR_FOR_EACH_SWEEP_SPLIT_VARIANT(X) X(70, 8, 1, 4, 4) X(71, 4, 3, 8, 6) X(72, 4, 3, 4, 6) X(73, 4, 2, 8, 4) X(74, 5, 2, 8, 4) X(75, 2, 7, 8, 14)
#else
#define PAPR_FOR_EACH_SWEEP_SPLIT_VARIANT(X)
#endif

int papr_sweep_variant(int variant)
{
#ifdef PAPR_MEASURE
    if ((variant >= 60 && variant <= 69) || (variant >= 90 && variant <= 99 && variant != 93 && variant != 96) ||
        (variant >= 120 && variant <= 129 && variant != 123 && variant != 126))
        return variant;  // ablations of <1024, 4> / <256, 8> (measurement only)
#endif
    switch (variant) {
#define X(V, PW, NB, LU, D) case V: return V;
        PAPR_FOR_EACH_SWEEP_SPLIT_VARIANT(X)
#undef X
#define X(V, B, U, P, L2, SM) case V: return V;
        PAPR_FOR_EACH_SWEEP_SP16_VARIANT(X)
#undef X
#define X(V, B, U, P) case V: return V;
        PAPR_FOR_EACH_SWEEP_LUT2_VARIANT(X)
#undef X
#define X(V, B, U, P) case V: return V;
        PAPR_FOR_EACH_SWEEP_VARIANT(X)
#undef X
    default: return -1;
    }
}

int papr_sweep_geometry(int variant, int *threads, uint64_t *tile_samples, size_t *stash_lds)
{
    if (variant >= 60 && variant <= 69)
        variant = 4;
    if (variant >= 90 && variant <= 99)
        variant = 0;
    if (variant >= 120 && variant <= 129)
        variant = 40;
    switch (variant) {
#define X(V, PW, NB, LU, D)                                                                                      \
    case V:                                                                                                       \
        *threads = (PW + PW * NB) * kWave;                                                                        \
        *tile_samples = 2ull * PW * kWave * LU;                                                                   \
        *stash_lds = (size_t)(PW * NB) * papr_sweep_slice_floats(4) * sizeof(float) +                             \
                     (size_t)PW * D * 512 * sizeof(float) + 16;                                                   \
        return 0;
        PAPR_FOR_EACH_SWEEP_SPLIT_VARIANT(X)
#undef X
#define X(V, B, U, P, L2, SM)                                                             \
    case V:                                                                                \
        *threads = B;                                                                      \
        *tile_samples = 2ull * B * U;                                                      \
        *stash_lds = (size_t)(B / kWave) * papr_sweep_slice_floats(U) * ((SM & 32) ? 2 : 1) * sizeof(float) + 16; \
        return 0;
        PAPR_FOR_EACH_SWEEP_SP16_VARIANT(X)
#undef X
#define X(V, B, U, P)                                                                     \
    case V:                                                                                \
        *threads = B;                                                                      \
        *tile_samples = 2ull * B * U;                                                      \
        *stash_lds = (size_t)(B / kWave) * papr_sweep_slice_floats(U) * sizeof(float) + 16; \
        return 0;
        PAPR_FOR_EACH_SWEEP_LUT2_VARIANT(X)
#undef X
#define X(V, B, U, P)                                                                     \
    case V:                                                                                \
        *threads = B;                                                                      \
        *tile_samples = 2ull * B * U;                                                      \
        *stash_lds = (size_t)(B / kWave) * papr_sweep_slice_floats(U) * sizeof(float) + 16; \
        return 0;
        PAPR_FOR_EACH_SWEEP_VARIANT(X)
#undef X
    default: return -1;
    }
}

#ifdef PAPR_MEASURE  // (tools/ablation_probe.py)
#define PAPR_FOR_EACH_ABLATION(X) X(60, 1) X(61, 2) X(62, 3) X(63, 4) X(64, 8) X(65, 16) X(66, 63) X(67, 32) X(68, 7) X(69, 24)
// (the same of <256, 8>: eight loads in flight per lane, two workgroups per CU — the geometry a stripped kernel reads fastest in)
#define PAPR_FOR_EACH_ABLATION2(X) X(90, 1) X(91, 2) X(92, 3) X(94, 8) X(95, 16) X(97, 32) X(98, 7) X(99, 24)
// (and of the default, <512, 8> with the branch-free stash)
#define PAPR_FOR_EACH_ABLATION3(X) X(120, 1) X(121, 2) X(122, 3) X(124, 8) X(125, 16) X(127, 32) X(128, 7) X(129, 24)
#else
#define PAPR_FOR_EACH_ABLATION(X)
#define PAPR_FOR_EACH_ABLATION2(X)
#define PAPR_FOR_EACH_ABLATION3(X)
#endif

void papr_launch_sweep(hipStream_t st, int variant, int blocks, size_t lds_bytes, const void *data, uint64_t ntiles,
                       uint64_t base_index, int map, papr_partial *out, const void *tail, uint32_t tail_samples,
                       const uint32_t *table, const papr_ccdf_params &P, unsigned long long *ghist, float *stash,
                       unsigned long long *seg_counts, uint64_t seg_cap, unsigned long long *gave_up,
                       unsigned long long *seg_real, const papr_ccdf_params *Pdev)
{
    switch (variant) {
#define X(V, PW, NB, LU, D)                                                                                          \
    case V:                                                                                                           \
        launch_maybe_timed((papr_sweep_split_kernel<PW, NB, LU, D>), dim3(blocks), dim3((PW + PW * NB) * kWave),      \
                           lds_bytes, st, (const float4 *)data, ntiles, base_index, map, out, (const float2 *)tail,   \
                           tail_samples, table, P, ghist, stash, seg_counts, seg_cap, gave_up, seg_real);                       \
        break;
        PAPR_FOR_EACH_SWEEP_SPLIT_VARIANT(X)
#undef X
#define X(V, B, U, PP, L2, SM)                                                                                       \
    case V:                                                                                                           \
        launch_maybe_timed((papr_sweep_kernel<B, U, true, PP, 0, L2, SM>), dim3(blocks), dim3(B), lds_bytes, st,    \
                           (const float4 *)data, ntiles, base_index, map, out, (const float2 *)tail, tail_samples,    \
                           table, P, ghist, stash, seg_counts, seg_cap, gave_up, seg_real, Pdev);                           \
        break;
        PAPR_FOR_EACH_SWEEP_SP16_VARIANT(X)
#undef X
#define X(V, A)                                                                                                      \
    case V:                                                                                                           \
        launch_maybe_timed((papr_sweep_kernel<1024, 4, true, 0, A>), dim3(blocks), dim3(1024), lds_bytes, st,         \
                           (const float4 *)data, ntiles, base_index, map, out, (const float2 *)tail, tail_samples,    \
                           table, P, ghist, stash, seg_counts, seg_cap, gave_up, seg_real, Pdev);                                            \
        break;
        PAPR_FOR_EACH_ABLATION(X)
#undef X
#define X(V, A)                                                                                                      \
    case V:                                                                                                           \
        launch_maybe_timed((papr_sweep_kernel<256, 8, true, 0, A>), dim3(blocks), dim3(256), lds_bytes, st,           \
                           (const float4 *)data, ntiles, base_index, map, out, (const float2 *)tail, tail_samples,    \
                           table, P, ghist, stash, seg_counts, seg_cap, gave_up, seg_real, Pdev);                     \
        break;
        PAPR_FOR_EACH_ABLATION2(X)
#undef X
#define X(V, A)                                                                                                      \
    case V:                                                                                                           \
        launch_maybe_timed((papr_sweep_kernel<512, 8, true, 0, A, false, 43>), dim3(blocks), dim3(512), lds_bytes, st, \
                           (const float4 *)data, ntiles, base_index, map, out, (const float2 *)tail, tail_samples,    \
                           table, P, ghist, stash, seg_counts, seg_cap, gave_up, seg_real, Pdev);                     \
        break;
        PAPR_FOR_EACH_ABLATION3(X)
#undef X
#define X(V, B, U, PP)                                                                                               \
    case V:                                                                                                           \
        launch_maybe_timed((papr_sweep_kernel<B, U, true, PP, 0, true>), dim3(blocks), dim3(B), lds_bytes, st,        \
                           (const float4 *)data, ntiles, base_index, map, out, (const float2 *)tail, tail_samples,    \
                           table, P, ghist, stash, seg_counts, seg_cap, gave_up, seg_real, Pdev);                                            \
        break;
        PAPR_FOR_EACH_SWEEP_LUT2_VARIANT(X)
#undef X
#define X(V, B, U, PP)                                                                                               \
    case V:                                                                                                           \
        launch_maybe_timed((papr_sweep_kernel<B, U, true, PP>), dim3(blocks), dim3(B), lds_bytes, st,                 \
                           (const float4 *)data, ntiles, base_index, map, out, (const float2 *)tail, tail_samples,    \
                           table, P, ghist, stash, seg_counts, seg_cap, gave_up, seg_real, Pdev);                                            \
        break;
        PAPR_FOR_EACH_SWEEP_VARIANT(X)
#undef X
    }
}

// Geometry variants of the second-generation sweep: id, waves per workgroup, 16-byte loads per lane per segment,
// next-segment prefetch, exact-sum pairs, stash store policy (0 plain, 1 nontemporal, 2 write-through).
#ifdef PAPR_MEASURE
#define PAPR_FOR_EACH_SWEEP2_VARIANT(X)                                                                         \
    X(32, 16, 8, 0, false, 2) X(34, 16, 4, 0, false, 2) X(35, 16, 4, 1, false, 2) X(30, 12, 8, 0, false, 2)      \
    X(41, 12, 8, 1, false, 2) X(42, 8, 4, 1, false, 2) X(44, 16, 8, 0, false, 6) X(45, 12, 8, 1, false, 6)       \
    X(48, 12, 8, 0, true, 2) X(49, 12, 8, 0, true, 6) X(50, 12, 8, 0, true, 10) X(51, 12, 8, 0, true, 0)         \
    X(52, 12, 8, 0, true, 14) X(54, 11, 8, 0, true, 10) X(55, 12, 8, 0, true, 18) X(56, 12, 8, 0, true, 26)                \
    X(57, 16, 8, 0, false, 18) X(58, 12, 8, 1, false, 18) X(59, 12, 8, 0, true, 50) X(53, 12, 8, 0, true, 16)        \
    X(46, 12, 8, 0, true, 17) X(47, 12, 8, 0, true, 58) X(43, 14, 8, 0, true, 26) X(33, 13, 8, 0, true, 26)       \
    X(31, 12, 8, 0, true, 218) X(29, 12, 8, 0, true, 90) X(28, 12, 8, 0, true, 154)
#else  // 56: the exact-sum default (ballot ring, lean sum); 48: its first form (returning-atomic ring, separate sum);
       // 32 / 41: the kernel without the pairs (tests)
#define PAPR_FOR_EACH_SWEEP2_VARIANT(X) \
    X(32, 16, 8, 0, false, 2) X(41, 12, 8, 1, false, 2) X(48, 12, 8, 0, true, 2) X(56, 12, 8, 0, true, 26)
#endif

int papr_sweep2_geometry(int variant, int *threads, uint64_t *seg_samples, size_t *lds_fixed, int *exact)
{
    switch (variant) {
#define X(V, W, U, PP, EX, WT)                                                                                  \
    case V:                                                                                                      \
        *threads = W * kWave;                                                                                    \
        *seg_samples = 2ull * kWave * U;                                                                         \
        *lds_fixed = (size_t)W * ((EX && !((WT) & 32)) ? 512u : 1024u) * sizeof(float) + (EX ? (size_t)W * 8192u : 0u); \
        *exact = EX ? 1 : 0;                                                                                     \
        return 0;
        PAPR_FOR_EACH_SWEEP2_VARIANT(X)
#undef X
    default: return -1;
    }
}

void papr_launch_sweep2(hipStream_t st, int variant, int blocks, size_t lds_bytes, const papr_sweep2_params &p)
{
    switch (variant) {
#define X(V, W, U, PP, EX, WT)                                                                                  \
    case V:                                                                                                      \
        launch_maybe_timed((papr_sweep2_kernel<W, U, PP, EX, WT>), dim3(blocks), dim3(W * kWave), lds_bytes, st, p); \
        break;
        PAPR_FOR_EACH_SWEEP2_VARIANT(X)
#undef X
    }
}

// Third form of the exact-sum sweep: id, waves per workgroup, stash mode (as papr_sweep_kernel's SMODE)
#define PAPR_FOR_EACH_SWEEP3_VARIANT(X) X(130, 8, 43, 4, true) X(131, 8, 43, 8, true) X(132, 8, 43, 8, false) X(133, 12, 43, 8, false) X(134, 12, 43, 4, false)

int papr_sweep3_geometry(int variant, int *threads, size_t *lds_fixed, int *exact)
{
    switch (variant) {
#define X(V, W, SM, H, EX)                                                                           \
    case V:                                                                                           \
        *threads = W * kWave;                                                                         \
        *lds_fixed = (size_t)W * PAPR_SWEEP3_SLICE_FLOATS * sizeof(float) + (EX ? (size_t)W * 8192u : 0u) + 16; \
        if (exact)                                                                                    \
            *exact = EX ? 1 : 0;                                                                      \
        return 0;
        PAPR_FOR_EACH_SWEEP3_VARIANT(X)
#undef X
    default: return -1;
    }
}

void papr_launch_sweep3(hipStream_t st, int variant, int blocks, size_t lds_bytes, const papr_sweep2_params &p)
{
    switch (variant) {
#define X(V, W, SM, H, EX)                                                                                       \
    case V: {                                                                                                     \
        papr_sweep2_params q = p;                                                                                 \
        q.lds_bytes = (uint32_t)lds_bytes;                                                                        \
        launch_maybe_timed((papr_sweep3_kernel<W, SM, H, EX>), dim3(blocks), dim3(W * kWave), lds_bytes, st, q);  \
        break;                                                                                                    \
    }
        PAPR_FOR_EACH_SWEEP3_VARIANT(X)
#undef X
    }
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
#define X(V, B, U, PP)                                                                                               \
    (void)hipFuncSetAttribute((const void *)papr_sweep_kernel<B, U, true, PP>,                                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    PAPR_FOR_EACH_SWEEP_VARIANT(X)
#undef X
#define X(V, A)                                                                                                      \
    (void)hipFuncSetAttribute((const void *)papr_sweep_kernel<1024, 4, true, 0, A>,                                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    PAPR_FOR_EACH_ABLATION(X)
#undef X
#define X(V, A)                                                                                                      \
    (void)hipFuncSetAttribute((const void *)papr_sweep_kernel<256, 8, true, 0, A>,                                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    PAPR_FOR_EACH_ABLATION2(X)
#undef X
#define X(V, A)                                                                                                      \
    (void)hipFuncSetAttribute((const void *)papr_sweep_kernel<512, 8, true, 0, A, false, 43>,                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    PAPR_FOR_EACH_ABLATION3(X)
#undef X
#define X(V, B, U, PP)                                                                                               \
    (void)hipFuncSetAttribute((const void *)papr_sweep_kernel<B, U, true, PP, 0, true>,                              \
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    PAPR_FOR_EACH_SWEEP_LUT2_VARIANT(X)
#undef X
#define X(V, W, U, PP, EX, WT)                                                                                  \
    (void)hipFuncSetAttribute((const void *)papr_sweep2_kernel<W, U, PP, EX, WT>,                                \
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    PAPR_FOR_EACH_SWEEP2_VARIANT(X)
#undef X
#define X(V, PW, NB, LU, D)                                                                                          \
    (void)hipFuncSetAttribute((const void *)papr_sweep_split_kernel<PW, NB, LU, D>,                                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    PAPR_FOR_EACH_SWEEP_SPLIT_VARIANT(X)
#undef X
#define X(V, B, U, PP, L2, SM)                                                                                       \
    (void)hipFuncSetAttribute((const void *)papr_sweep_kernel<B, U, true, PP, 0, L2, SM>,                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    PAPR_FOR_EACH_SWEEP_SP16_VARIANT(X)
#undef X
#define X(V, W, SM, H, EX) \
    (void)hipFuncSetAttribute((const void *)papr_sweep3_kernel<W, SM, H, EX>, hipFuncAttributeMaxDynamicSharedMemorySize, want);
    PAPR_FOR_EACH_SWEEP3_VARIANT(X)
#undef X
    (void)hipFuncSetAttribute((const void *)papr_ccdf_power_kernel<true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, want);
    (void)hipFuncSetAttribute((const void *)papr_ccdf_power_kernel<false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, want);
}
