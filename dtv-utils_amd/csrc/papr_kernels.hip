// papr_kernels.hip — gfx950 (MI355X / CDNA4) kernels for the papr hot path.
//
// Both passes are pure HBM streaming reductions (8 B per IQ sample per pass,
// 15-23 VALU ops per sample, no data reuse): no MFMA, no tiling through LDS for
// the samples themselves.  What matters is 16-byte-per-lane coalesced
// nontemporal loads with the next ones already in flight, the right number of
// resident wave64s per CU (measured: 8 for pass 1, 16 for pass 2), and keeping
// the reduction state (trackers, LDS histograms) off the global-memory path.
//
//   papr_stats_kernel<B,U,NT,PIPE,TSUM>  pass 1  (reference papr.c:102-128)
//   papr_stats_finalize    tail samples + fixed-order merge of workgroup partials
//   papr_first_nan_kernel  only launched when the sum came out NaN
//   papr_ccdf_kernel<B,U,NT,PIPE,LUT>    pass 2  (reference papr.c:145-152 / 177-184):
//                          LUT = exact bit-pattern lookup table, else a binary search for
//                          level tables the LUT cannot hold; LDS-privatised histograms
//   papr_generate_kernel   synthetic IQ (include/papr_synth.h) straight into HBM
// (the bit-exact sequential-sum kernels live in papr_exact.hip)
//
// Arithmetic contract (SURVEY.md appendix A rule 3): power = fl(fl(I*I) +
// fl(Q*Q)) in float with NO fused multiply-add; this file is compiled with
// -ffp-contract=off and also spells the roundings out with __fmul_rn/__fadd_rn.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "papr_kernels.h"
#include "papr_device.h"
#include "papr_stream.h"

// =============================================================================
// pass 1 — power, double sum, first-index peak and component extrema
// =============================================================================
// One launch covers `ntiles` full tiles (2 * BLOCK * U samples each) of
// `data`; the < 1 tile remainder of the shard is folded in by
// papr_stats_finalize.  Per lane: U independent 16-byte loads are issued
// before any arithmetic (U KiB in flight per wave; with PIPE the next tile's
// loads are issued before the current tile is reduced), then 2U samples are
// folded into a double partial sum and five (value, 32-bit sample code)
// trackers.  The code (iteration * 2U + slot) is expanded to a 64-bit global
// sample index once, after the loop.

// TSUM (exact-sum mode): every wave also stores the sum of its lanes' samples of
// each tile, tile_sums[(tile_offset + tile) * (BLOCK/64) + wave]; the waves of a
// workgroup together cover the tile, so those BLOCK/64 numbers add up to the
// tile's sum (papr_exact.hip turns them into per-tile prefix sums).
template <int BLOCK, int U, bool NT, int PIPE, bool TSUM = false>
__global__ __launch_bounds__(BLOCK) void papr_stats_kernel(const float4 *__restrict__ data, uint64_t ntiles,
                                                            uint64_t base_index, int map,
                                                            papr_partial *__restrict__ out,
                                                            double *__restrict__ tile_sums, uint64_t tile_offset)
{
    constexpr uint64_t TILE_F4 = (uint64_t)BLOCK * U;
    const uint32_t t = threadIdx.x;
    const TileWalk w = tile_walk(blockIdx.x, gridDim.x, ntiles, map);
    auto emit_tile_sum = [&](uint32_t it, double lane_sum) {
        if constexpr (TSUM) {
            const double ws = wave_sum_to_lane63(lane_sum);
            if ((t & (kWave - 1)) == kWave - 1)
                tile_sums[(tile_offset + w.first + (uint64_t)it * w.stride) * (BLOCK / kWave) + t / kWave] = ws;
        }
    };

    StatsRegs r = {0.0, 0.f, 0.f, 0.f, 0.f, 0.f, 0, 0, 0, 0, 0};
    const float4 *p = data + w.first * TILE_F4 + t;
    const uint64_t step = w.stride * TILE_F4;
    uint32_t code = 0;
    if constexpr (PIPE == 2) {
        // true double buffering: two register sets and a loop unrolled by two, so that the compiler's
        // wait counts are exact ("the other set may still be in flight") instead of the conservative
        // vmcnt(0) it derives for the copy-based form below
        float4 a[U], b[U];
        const float4 *plast = data + (w.first + (uint64_t)(w.count ? w.count - 1 : 0) * w.stride) * TILE_F4 + t;
        if (w.count)
            load_tile<BLOCK, U, NT>(a, p);
        uint32_t it = 0;
        for (; it + 1 < w.count; it += 2, code += 4 * U) {
            load_tile<BLOCK, U, NT>(b, p + step);
            emit_tile_sum(it, stats_fold<U, TSUM>(r, a, code));
            p += 2 * step;
            load_tile<BLOCK, U, NT>(a, it + 2 < w.count ? p : plast);  // past the end: harmless re-read
            emit_tile_sum(it + 1, stats_fold<U, TSUM>(r, b, code + 2 * U));
        }
        if (it < w.count)
            emit_tile_sum(it, stats_fold<U, TSUM>(r, a, code));
    } else if constexpr (PIPE == 1) {
        float4 cur[U], nxt[U];
        if (w.count)
            load_tile<BLOCK, U, NT>(cur, p);
        for (uint32_t it = 0; it < w.count; it++, code += 2 * U) {
            p += step;
            if (it + 1 < w.count)
                load_tile<BLOCK, U, NT>(nxt, p);
            emit_tile_sum(it, stats_fold<U, TSUM>(r, cur, code));
#pragma unroll
            for (int u = 0; u < U; u++)
                cur[u] = nxt[u];
        }
    } else {
        for (uint32_t it = 0; it < w.count; it++, p += step, code += 2 * U) {
            float4 x[U];
            load_tile<BLOCK, U, NT>(x, p);
            emit_tile_sum(it, stats_fold<U, TSUM>(r, x, code));
        }
    }

    stats_finish<BLOCK, U>(r, w, base_index, out);
}

// Tail samples + merge of all workgroup partials, one workgroup, fixed order.
// `tail` points at the first sample not covered by full tiles (may hold an odd
// count; samples are read as scalar float pairs).
__global__ __launch_bounds__(PAPR_BLOCK) void papr_stats_finalize(const float2 *__restrict__ tail, uint32_t tail_samples,
                                                                   uint64_t tail_base_index,
                                                                   const papr_partial *__restrict__ partials,
                                                                   uint32_t npartials, papr_partial *__restrict__ result,
                                                                   papr_partial *__restrict__ result_dev,
                                                                   const unsigned long long *__restrict__ copy_src,
                                                                   unsigned long long *__restrict__ copy_dst,
                                                                   uint32_t copy_words)
{
    // (single-wait step: the sweep's bins and segment counters go to mapped host memory from here — a D2H copy of
    // their own would be one more packet pair in the stream)
    for (uint32_t w = threadIdx.x; w < copy_words; w += PAPR_BLOCK)
        copy_dst[w] = copy_src[w];
    LaneStats s;
    s.sum = 0.0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        s.val[k] = 0.f;
        s.idx[k] = 0;
    }
    // partials first (they precede the tail in file order), each thread a fixed subset
    for (uint32_t r = threadIdx.x; r < npartials; r += PAPR_BLOCK) {
        const papr_partial q = partials[r];
        s.sum += q.sum;
        if (beats<false>(q.val[0], q.idx[0], s.val[0], s.idx[0])) { s.val[0] = q.val[0]; s.idx[0] = q.idx[0]; }
        if (beats<false>(q.val[1], q.idx[1], s.val[1], s.idx[1])) { s.val[1] = q.val[1]; s.idx[1] = q.idx[1]; }
        if (beats<true>(q.val[2], q.idx[2], s.val[2], s.idx[2])) { s.val[2] = q.val[2]; s.idx[2] = q.idx[2]; }
        if (beats<false>(q.val[3], q.idx[3], s.val[3], s.idx[3])) { s.val[3] = q.val[3]; s.idx[3] = q.idx[3]; }
        if (beats<true>(q.val[4], q.idx[4], s.val[4], s.idx[4])) { s.val[4] = q.val[4]; s.idx[4] = q.idx[4]; }
    }
    for (uint32_t k = threadIdx.x; k < tail_samples; k += PAPR_BLOCK) {
        const float2 x = tail[k];
        lane_stats_sample(s, x.x, x.y, tail_base_index + k);
    }
    block_reduce_stats<PAPR_BLOCK / kWave>(s);
    if (threadIdx.x == 0) {
        papr_partial r;
        r.sum = s.sum;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            r.idx[k] = s.idx[k];
            r.val[k] = s.val[k];
        }
        r.pad = npartials ? partials[0].pad : 0;  // (a sweep's workgroup 0: when it was done, and on which XCD)
        *result = r;
        if (result_dev)
            *result_dev = r;  // (for papr_true_table_kernel: `result` is host memory)
    }
}

// Rare path: the double sum came out NaN, so some power value is NaN.  Find the
// first such sample and the sign x86 gives its NaN (sign of I if I is NaN,
// else sign of Q): key = index << 1 | sign, minimised.
__global__ __launch_bounds__(PAPR_BLOCK) void papr_first_nan_kernel(const float2 *__restrict__ data, uint64_t nsamples,
                                                                     uint64_t base_index,
                                                                     unsigned long long *__restrict__ key)
{
    const uint64_t stride = (uint64_t)gridDim.x * PAPR_BLOCK;
    for (uint64_t k = (uint64_t)blockIdx.x * PAPR_BLOCK + threadIdx.x; k < nsamples; k += stride) {
        const float2 x = data[k];
        const float pw = power_of(x.x, x.y);
        if (pw != pw) {
            const uint32_t sign = (x.x != x.x ? __float_as_uint(x.x) : __float_as_uint(x.y)) >> 31;
            atomicMin(key, (unsigned long long)(((base_index + k) << 1) | sign));
        }
    }
}

// =============================================================================
// pass 2 — CCDF counting
// =============================================================================
// counts_above[j] = #{v > level[j]} is the suffix sum of a histogram over the
// intervals between sorted thresholds, so each sample needs ONE bin index
// k(v) = #{j : level[j] < v} instead of L compares.  For v >= 0 the IEEE bit
// pattern is monotone in v, so with key_j = "smallest bit pattern whose float
// is > level[j]", k(v) = #{j : key_j <= bits(v)} exactly, in integers.
//
// LUT form: cut the bit-pattern axis into cells of 2^shift patterns; the host
// guarantees at most one key per cell.  lut[cell] = {keys in lower cells, the
// key in this cell or 0xFFFFFFFF}, so k = lut.x + (bits >= lut.y): one 8-byte
// LDS read and one compare per sample.  Bin 0 (below the lowest level, ~63 %
// of Gaussian-like IQ) is never counted.  Counters are LDS-privatised per wave
// (ds_add_u32) and flushed once per workgroup with 64-bit global atomics.

template <int BLOCK, int U, bool NT, int PIPE, bool LUT>
__global__ __launch_bounds__(BLOCK) void papr_ccdf_kernel(const float4 *__restrict__ data, uint64_t ntiles, int map,
                                                           const float2 *__restrict__ tail, uint32_t tail_samples,
                                                           const uint32_t *__restrict__ table, papr_ccdf_params P,
                                                           unsigned long long *__restrict__ ghist)
{
    constexpr uint64_t TILE_F4 = (uint64_t)BLOCK * U;
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

    auto count = [&](float pw) {
        const uint32_t bits = __float_as_uint(pw);
        const uint32_t k = LUT ? lut_bin(bits, lut, P) : search_bin(bits, tab, P);
        if (k)
            atomicAdd(&my[k], 1u);
    };
    auto fold = [&](const float4(&x)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            count(power_of(x[u].x, x[u].y));
            count(power_of(x[u].z, x[u].w));
        }
    };

    const uint32_t t = threadIdx.x;
    const TileWalk w = tile_walk(blockIdx.x, gridDim.x, ntiles, map);
    const float4 *p = data + w.first * TILE_F4 + t;
    const uint64_t step = w.stride * TILE_F4;
    if constexpr (PIPE == 2) {  // true double buffering, see papr_stats_kernel
        float4 a[U], b[U];
        const float4 *plast = data + (w.first + (uint64_t)(w.count ? w.count - 1 : 0) * w.stride) * TILE_F4 + t;
        if (w.count)
            load_tile<BLOCK, U, NT>(a, p);
        uint32_t it = 0;
        for (; it + 1 < w.count; it += 2) {
            load_tile<BLOCK, U, NT>(b, p + step);
            fold(a);
            p += 2 * step;
            load_tile<BLOCK, U, NT>(a, it + 2 < w.count ? p : plast);
            fold(b);
        }
        if (it < w.count)
            fold(a);
    } else if constexpr (PIPE == 1) {
        float4 cur[U], nxt[U];
        if (w.count)
            load_tile<BLOCK, U, NT>(cur, p);
        for (uint32_t it = 0; it < w.count; it++) {
            p += step;
            if (it + 1 < w.count)
                load_tile<BLOCK, U, NT>(nxt, p);
            fold(cur);
#pragma unroll
            for (int u = 0; u < U; u++)
                cur[u] = nxt[u];
        }
    } else {
        for (uint32_t it = 0; it < w.count; it++, p += step) {
            float4 x[U];
            load_tile<BLOCK, U, NT>(x, p);
            fold(x);
        }
    }
    if (blockIdx.x == gridDim.x - 1) {
        for (uint32_t k = t; k < tail_samples; k += BLOCK) {
            const float2 x = tail[k];
            count(power_of(x.x, x.y));
        }
    }
    hist_flush<BLOCK>(hist, nbins, P.copies, ghist);
}

// =============================================================================
// synthetic IQ generator (include/papr_synth.h), two samples per lane per store
// =============================================================================
__global__ __launch_bounds__(PAPR_BLOCK) void papr_generate_kernel(float2 *__restrict__ out, uint64_t nsamples,
                                                                    uint64_t first_index, papr_synth_spec spec)
{
    const uint64_t stride = (uint64_t)gridDim.x * PAPR_BLOCK;
    for (uint64_t k = (uint64_t)blockIdx.x * PAPR_BLOCK + threadIdx.x; k < nsamples; k += stride) {
        float i, q;
        papr_synth_sample(&spec, first_index + k, &i, &q);
        out[k] = make_float2(i, q);
    }
}

// ---- launch wrappers (called from the host runtime, papr_runtime.cpp & co., through plain C++) ------
//
// Kernel geometry variants; tile = 2 * block * unroll samples (<= 8192).
//   id: block x loads-per-lane, P = software-pipelined (next tile's loads issued
//   before the current tile is reduced)
//    0: 256x8      1: 256x4 P    2: 256x8 P    3: 512x8      4: 1024x4
//    5: 256x16     6: 512x4 P    7: 256x4      8: 1024x4 P   9: 1024x2 P
//   10: 512x2 P   11: 256x2 P   12: 1024x2    13: 512x4
//   14: 256x4 D   15: 512x4 D   16: 256x8 D   17: 1024x4 D      (D = true double buffering, unrolled by two)
// Plain (non-"nt") loads exist for variant 0 only (A/B of the nontemporal hint).

// The product carries the geometries that are defaults or cover a distinct code shape in the tests (plain, prefetch
// and double-buffered loops; 4-, 8- and 16-wave workgroups); `make MEASURE=1` adds the rest of the sweep space that
// tools/sweep.py explored.
#ifdef PAPR_MEASURE
#define PAPR_FOR_EACH_VARIANT(X) \
    X(0, 256, 8, 0) X(1, 256, 4, 1) X(2, 256, 8, 1) X(3, 512, 8, 0) X(4, 1024, 4, 0) \
    X(5, 256, 16, 0) X(6, 512, 4, 1) X(7, 256, 4, 0) X(8, 1024, 4, 1) X(9, 1024, 2, 1) \
    X(10, 512, 2, 1) X(11, 256, 2, 1) X(12, 1024, 2, 0) X(13, 512, 4, 0) \
    X(14, 256, 4, 2) X(15, 512, 4, 2) X(16, 256, 8, 2) X(17, 1024, 4, 2)
#else
#define PAPR_FOR_EACH_VARIANT(X) \
    X(0, 256, 8, 0) X(1, 256, 4, 1) X(4, 1024, 4, 0) X(6, 512, 4, 1) X(13, 512, 4, 0) X(14, 256, 4, 2)
#endif

int papr_variant_geometry(int variant, int *block, int *unroll)
{
    switch (variant) {
#define X(V, B, U, P) case V: *block = B; *unroll = U; return 0;
        PAPR_FOR_EACH_VARIANT(X)
#undef X
    default: return -1;
    }
}

void papr_launch_stats(hipStream_t st, int variant, int blocks, bool nt, const void *data, uint64_t ntiles,
                       uint64_t base_index, int map, papr_partial *out)
{
    if (variant == 0 && !nt) {
        hipLaunchKernelGGL((papr_stats_kernel<256, 8, false, 0>), dim3(blocks), dim3(256), 0, st,
                           (const float4 *)data, ntiles, base_index, map, out, (double *)nullptr, (uint64_t)0);
        return;
    }
    switch (variant) {
#define X(V, B, U, P)                                                                                               \
    case V:                                                                                                          \
        hipLaunchKernelGGL((papr_stats_kernel<B, U, true, P>), dim3(blocks), dim3(B), 0, st, (const float4 *)data,   \
                           ntiles, base_index, map, out, (double *)nullptr, (uint64_t)0);                            \
        break;
        PAPR_FOR_EACH_VARIANT(X)
#undef X
    }
}

// exact-sum mode: fixed geometry (PAPR_EXACT_TILE_SAMPLES per tile, 4 waves), per-wave tile sums stored
void papr_launch_stats_tilesums(hipStream_t st, int blocks, const void *data, uint64_t ntiles, uint64_t base_index,
                                int map, papr_partial *out, double *tile_sums, uint64_t tile_offset)
{
    static_assert(2 * 256 * 4 == PAPR_EXACT_TILE_SAMPLES, "exact-sum tile geometry");
    hipLaunchKernelGGL((papr_stats_kernel<256, 4, true, 1, true>), dim3(blocks), dim3(256), 0, st,
                       (const float4 *)data, ntiles, base_index, map, out, tile_sums, tile_offset);
}

void papr_launch_stats_finalize(hipStream_t st, const void *tail, uint32_t tail_samples, uint64_t tail_base_index,
                                const papr_partial *partials, uint32_t npartials, papr_partial *result,
                                papr_partial *result_dev, const unsigned long long *copy_src, unsigned long long *copy_dst,
                                uint32_t copy_words)
{
    hipLaunchKernelGGL(papr_stats_finalize, dim3(1), dim3(PAPR_BLOCK), 0, st, (const float2 *)tail, tail_samples,
                       tail_base_index, partials, npartials, result, result_dev, copy_src, copy_dst, copy_words);
}

// The ingest's H2D leg as a kernel: a few workgroups pull a pinned staging buffer (mapped host memory) over the link with
// 16-byte nontemporal loads, four in flight per lane, and store it where the copy engine would have put it.  48 workgroups
// move 16 MiB chunks at 55-56 GB/s where hipMemcpyAsync gives 52-54 (17 us between two of its copies, 7 ms for its queue
// at the first one: tools/zero_copy_probe.hip); more workgroups are SLOWER (256: 50 GB/s) — the link wants few, long streams.
typedef unsigned int papr_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void papr_pull_kernel(const papr_u32x4 *__restrict__ src, papr_u32x4 *__restrict__ dst, uint64_t n16,
                                                        const unsigned long long *__restrict__ src8, unsigned long long *__restrict__ dst8,
                                                        uint32_t has_word)
{
    constexpr int U = 4;
    const uint64_t stride = (uint64_t)gridDim.x * 256 * U;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 * U + threadIdx.x; i < n16; i += stride) {
        papr_u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            v[u] = i + u * 256 < n16 ? __builtin_nontemporal_load(src + i + u * 256) : papr_u32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; u++)
            if (i + u * 256 < n16)
                __builtin_nontemporal_store(v[u], dst + i + u * 256);
    }
    if (has_word && blockIdx.x == 0 && threadIdx.x == 0)
        *dst8 = *src8;  // (an odd number of samples: the last 8 bytes)
}

void papr_launch_pull(hipStream_t st, const void *src_mapped, void *dst, uint64_t bytes)
{
    const uint64_t n16 = bytes / 16;
    const uint32_t has_word = (bytes & 8) ? 1u : 0u;
    hipLaunchKernelGGL(papr_pull_kernel, dim3(48), dim3(256), 0, st, (const papr_u32x4 *)src_mapped, (papr_u32x4 *)dst, n16,
                       (const unsigned long long *)((const char *)src_mapped + n16 * 16),
                       (unsigned long long *)((char *)dst + n16 * 16), has_word);
}

void papr_launch_first_nan(hipStream_t st, int blocks, const void *data, uint64_t nsamples, uint64_t base_index,
                           unsigned long long *key)
{
    hipLaunchKernelGGL(papr_first_nan_kernel, dim3(blocks), dim3(PAPR_BLOCK), 0, st, (const float2 *)data, nsamples,
                       base_index, key);
}

template <int B, int U, bool NT, int P>
static void launch_ccdf_variant(hipStream_t st, int blocks, bool lut, size_t lds_bytes, const void *data,
                                uint64_t ntiles, int map, const void *tail, uint32_t tail_samples,
                                const uint32_t *table, const papr_ccdf_params &Pm, unsigned long long *ghist)
{
    if (lut)
        hipLaunchKernelGGL((papr_ccdf_kernel<B, U, NT, P, true>), dim3(blocks), dim3(B), lds_bytes, st,
                           (const float4 *)data, ntiles, map, (const float2 *)tail, tail_samples, table, Pm, ghist);
    else
        hipLaunchKernelGGL((papr_ccdf_kernel<B, U, NT, P, false>), dim3(blocks), dim3(B), lds_bytes, st,
                           (const float4 *)data, ntiles, map, (const float2 *)tail, tail_samples, table, Pm, ghist);
}

void papr_launch_ccdf(hipStream_t st, int variant, int blocks, bool nt, bool lut, size_t lds_bytes, const void *data,
                      uint64_t ntiles, int map, const void *tail, uint32_t tail_samples, const uint32_t *table,
                      const papr_ccdf_params &P, unsigned long long *ghist)
{
    if (variant == 0 && !nt) {
        launch_ccdf_variant<256, 8, false, 0>(st, blocks, lut, lds_bytes, data, ntiles, map, tail, tail_samples,
                                                  table, P, ghist);
        return;
    }
    switch (variant) {
#define X(V, B, U, PP)                                                                                              \
    case V:                                                                                                          \
        launch_ccdf_variant<B, U, true, PP>(st, blocks, lut, lds_bytes, data, ntiles, map, tail, tail_samples,       \
                                            table, P, ghist);                                                        \
        break;
        PAPR_FOR_EACH_VARIANT(X)
#undef X
    }
}

void papr_launch_generate(hipStream_t st, int blocks, void *out, uint64_t nsamples, uint64_t first_index,
                          const papr_synth_spec &spec)
{
    hipLaunchKernelGGL(papr_generate_kernel, dim3(blocks), dim3(PAPR_BLOCK), 0, st, (float2 *)out, nsamples, first_index,
                       spec);
}

// Largest dynamic-LDS request a pass-2 workgroup may make (160 KiB LDS per CU on gfx950, default
// limit 64 KiB).  Function attributes belong to the CURRENT device, so every context calls
// papr_kernels_prepare_device() once after hipSetDevice (a multi-GPU process has several).
int papr_ccdf_max_dynamic_lds(void)
{
    return 160 * 1024 - 2048;
}

void papr_kernels_prepare_device(void)
{
    const int want = papr_ccdf_max_dynamic_lds();
#define X(V, B, U, P)                                                                                               \
    (void)hipFuncSetAttribute((const void *)papr_ccdf_kernel<B, U, true, P, true>,                                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);                                     \
    (void)hipFuncSetAttribute((const void *)papr_ccdf_kernel<B, U, true, P, false>,                                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    PAPR_FOR_EACH_VARIANT(X)
#undef X
    (void)hipFuncSetAttribute((const void *)papr_ccdf_kernel<256, 8, false, 0, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    (void)hipFuncSetAttribute((const void *)papr_ccdf_kernel<256, 8, false, 0, false>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    papr_exact_prepare_device();
    papr_sweep_prepare_device();
}
