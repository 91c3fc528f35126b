// papr_exact.hip — bit-exact emulation of the reference's SEQUENTIAL double
// accumulator (`sum += value`, papr.c:104) on the GPU.
//
// Why it is parallelisable.  All terms are >= 0, so the running sum S only
// grows.  While S stays inside one binade [2^E, 2^(E+1)) its ulp is the fixed
// u = 2^(E-52), and fl(S + v) = S + u * round_half_even(v / u) where the
// half-way case is resolved by the parity of S/u ALONE.  So "add v" acts on S as
// a function that depends on S only through one bit, and a whole run of
// additions inside a binade collapses to a pair (D0, D1): the total increment
// for even / odd entry parity.  Pairs compose associatively
//     (f then g)(p) = f.D[p] + g.D[p ^ lsb(f.D[p] / u)],
// so they can be built per lane and merged in file order by an ordered tree.
//
// How a lane gets its pair without any integer rounding logic: it runs the
// additions themselves, in double, from the two canonical entry states
// M0 = 2^E (even) and M1 = 2^E + u (odd).  The hardware's round-to-nearest-even
// then does exactly what it would do to the real S, as long as the lane's own
// total stays far below 2^E — which the tile classification guarantees.
//
// What cannot be collapsed: the additions during which S changes binade.  Pass 1
// (papr_stats_kernel<..., TSUM>) leaves a sum per 2048-sample tile; a scan gives
// each tile an accurate prefix P, and a tile is "safe" for binade E only if
// [P(1-d), (P+s)(1+d)] lies inside [2^E, 2^(E+1)) with margin d >= the worst-case
// drift of a sequential sum (and s <= 2^(E-2)).  The few tiles that are not
// provably safe (one or two per binade crossing, plus the very first tiles) are
// shipped raw and added one by one on the host, which also chains everything
// (papr_exact_chain in papr_host.c) — a few thousand dependent operations.
//
// Kernels:  papr_exact_block_sums -> papr_exact_classify (each block adds up the block sums in front of it)
//           papr_exact_seg_kernel   (8 B/sample HBM read; LDS transpose so that each
//                                    lane owns 16 CONSECUTIVE samples)
//           papr_exact_group_kernel (pre-composes 128-tile groups)

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>

#include "papr_kernels.h"
#include "papr_device.h"
#include "papr_exact_format.h"
#include "papr_stream.h"

namespace {

constexpr int kTilesPerBlock = 1024;  // classification workgroup: 256 threads x 4 tiles
constexpr int kSegF4 = PAPR_EXACT_SEG_SAMPLES / 2;  // float4 slots per segment (512)
constexpr int kRows = kSegF4 / kWave;               // 16-byte loads per lane per segment (8)
constexpr int kRunSamples = 16, kTileRuns = PAPR_EXACT_TILE_SAMPLES / kRunSamples;  // (papr_exact_format.h: PAPR_XF_RUN_SAMPLES)
constexpr size_t kRawRecBytes = 8 + 4 * (size_t)PAPR_EXACT_TILE_SAMPLES + 4 * kTileRuns + 16 * kTileRuns;  // sizeof(papr_exact_raw_rec)
static_assert(kRawRecBytes == sizeof(papr_exact_raw_rec) && kRunSamples == PAPR_XF_RUN_SAMPLES && PAPR_EXACT_TILE_SAMPLES == PAPR_XF_TILE_SAMPLES,
              "papr_exact_format.h");

// LDS transpose without padding: the lane that owns run r (8 consecutive float4 = 16 samples)
// finds its w-th float4 at slot r*8 + (w ^ ((r >> 1) & 7)).  Writers (8 consecutive lanes fill one
// run) still cover one contiguous 128 bytes; readers (lane l reads run l, same w) land on 16
// different 16-byte bank groups per 16-lane group: conflict-free both ways, 8 KiB per wave.
__device__ __forceinline__ int transpose_slot(int run, int w)
{
    return run * kRows + (w ^ ((run >> 1) & 7));
}

__device__ __forceinline__ double two_pow(int e)  // 2^e for normal results
{
    return __longlong_as_double((long long)(e + 1023) << 52);
}

// a tile's sum: from pass 1's four per-wave sums, or (SEGD: the one-read sweep) from D0 of its two segments' pairs
template <bool SEGD>
__device__ __forceinline__ double tile_sum_of(const double *tws, uint64_t tile)
{
    if constexpr (SEGD) {
        const double *p = tws + tile * 4;  // two double2 per tile: (D0, D1) of either segment
        return p[0] + p[2];
    } else {
        const double *p = tws + tile * PAPR_EXACT_TILE_WAVES;
        return ((p[0] + p[1]) + p[2]) + p[3];
    }
}

// (f then g) for entry parity 0 / 1; m0 = 2^E
struct Pair {
    double d0, d1;
};

__device__ __forceinline__ Pair compose(Pair f, Pair g, double m0)
{
    const int q0 = __double2loint(m0 + f.d0) & 1;        // parity after f from an even entry
    const int q1 = (__double2loint(m0 + f.d1) & 1) ^ 1;  // ... from an odd entry
    Pair h;
    h.d0 = f.d0 + (q0 ? g.d1 : g.d0);
    h.d1 = f.d1 + (q1 ? g.d1 : g.d0);
    return h;
}

// value of lane (l + SHIFT) of the same 16-lane row (garbage-free: 0 where there is none), via DPP
template <int SHIFT>
__device__ __forceinline__ double row_shl_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x100 + SHIFT, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x100 + SHIFT, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int SHIFT>
__device__ __forceinline__ Pair compose_row_step(Pair f, double m0)
{
    Pair g;
    g.d0 = row_shl_f64<SHIFT>(f.d0);
    g.d1 = row_shl_f64<SHIFT>(f.d1);
    return compose(f, g, m0);
}

// ordered merge of the 64 lanes' pairs (lane order = file order); result in lane 0.
// Levels 1,2,4,8 stay inside 16-lane rows (DPP row_shl: VALU latency only); the two
// cross-row levels go through ds_bpermute.  Only lanes that are multiples of twice
// the step stay meaningful at each level.
__device__ __forceinline__ Pair wave_compose(Pair f, double m0)
{
    f = compose_row_step<1>(f, m0);
    f = compose_row_step<2>(f, m0);
    f = compose_row_step<4>(f, m0);
    f = compose_row_step<8>(f, m0);
#pragma unroll
    for (int off = 16; off < kWave; off <<= 1) {
        Pair g;
        g.d0 = __shfl_down(f.d0, off, kWave);
        g.d1 = __shfl_down(f.d1, off, kWave);
        f = compose(f, g, m0);
    }
    return f;
}

}  // namespace

// ---- tile prefix sums and classification ------------------------------------------

template <bool SEGD>
__global__ __launch_bounds__(256) void papr_exact_block_sums(const double *__restrict__ tws, uint64_t ntiles,
                                                              double *__restrict__ block_sums, uint32_t *__restrict__ zero_word)
{
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0)
        *zero_word = 0;  // (the redo list's counter, for the classification that follows: saves a memset in the stream)
    __shared__ double sh[256];
    const uint64_t t0 = (uint64_t)blockIdx.x * kTilesPerBlock + (uint64_t)threadIdx.x * 4;
    double s = 0.0;
    for (int k = 0; k < 4; k++)
        if (t0 + k < ntiles)
            s += tile_sum_of<SEGD>(tws, t0 + k);
    // (any order will do: the prefixes only have to be good to `delta`)
    const double w = wave_reduce_sum(s);
    if ((threadIdx.x & (kWave - 1)) == 0)
        sh[threadIdx.x / kWave] = w;
    __syncthreads();
    if (threadIdx.x == 0)
        block_sums[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// `spec` (one-read sweep): the binade each tile's pairs were built for; a tile that is provably inside ANOTHER binade
// (or had none) goes on the redo list — its two segments are recomputed by papr_exact_seg_kernel's list form
template <bool SEGD>
__global__ __launch_bounds__(256) void papr_exact_classify(const double *__restrict__ tws, uint64_t ntiles,
                                                            const double *__restrict__ block_prefix, double delta,
                                                            int32_t *__restrict__ tile_E,
                                                            uint32_t *__restrict__ ambig_list, uint32_t ambig_cap,
                                                            uint32_t *__restrict__ ambig_count,
                                                            const int32_t *__restrict__ spec,
                                                            uint32_t *__restrict__ redo_list, uint32_t redo_cap,
                                                            uint32_t *__restrict__ redo_count,
                                                            const unsigned long long *__restrict__ n_total_dev,
                                                            unsigned long long n_shard, double before,
                                                            const double *__restrict__ before_dev)
{
    if (n_total_dev) {  // (peers: the file's length is known on the device only — the margin as the host computes it)
        const unsigned long long nt = *n_total_dev;
        const double d = 4.0 * (double)(nt > n_shard ? nt : n_shard) * 1.1102230246251565e-16;  // (papr_exact_rt.cpp: kDeltaPerSample)
        delta = d > 1.0e-6 ? d : 1.0e-6;
    }
    __shared__ double sh[256];
    const uint64_t t0 = (uint64_t)blockIdx.x * kTilesPerBlock + (uint64_t)threadIdx.x * 4;
    double s[4], tot = 0.0;
    for (int k = 0; k < 4; k++) {
        s[k] = t0 + k < ntiles ? tile_sum_of<SEGD>(tws, t0 + k) : 0.0;
        tot += s[k];
    }
    // this block's prefix: `before` + the sums of the blocks in front of it, added up here (a few hundred values out of the
    // L2 — a scan kernel of its own between the block sums and this one was a launch and 5 us of one workgroup).  Any
    // order will do: the prefix only has to be good to `delta` — wave reductions and a wave-level scan (one thread
    // walking 512 LDS words was 8 of this kernel's 13 us).
    __shared__ double sh_pre[256 / kWave];
    {
        double a = 0.0;
        for (uint32_t k = threadIdx.x; k < blockIdx.x; k += 256)
            a += block_prefix[k];
        a = wave_reduce_sum(a);
        if ((threadIdx.x & (kWave - 1)) == 0)
            sh_pre[threadIdx.x / kWave] = a;
    }
    const double ex = block_exclusive_scan<double, 256>(tot, sh, (double *)nullptr);  // (its barriers also publish sh_pre)
    double P = ((before_dev ? *before_dev : before) + ((sh_pre[0] + sh_pre[1]) + (sh_pre[2] + sh_pre[3]))) + ex;
    for (int k = 0; k < 4; k++) {
        if (t0 + k >= ntiles)
            break;
        int32_t cls = PAPR_EXACT_AMBIG;
        const double sk = s[k];
        if (sk == 0.0) {
            cls = PAPR_EXACT_ZERO;  // sum of non-negative terms is +0 only if every term is +0
        } else if (sk > 0.0 && sk < 1.0e300 && P > 0.0 && P < 1.0e300) {
            const int biased = (int)((__double_as_longlong(P) >> 52) & 0x7ff);
            const int E = biased - 1023;
            if (biased != 0 && E >= -960) {
                const double m0 = two_pow(E);
                const double lo = P * (1.0 - delta), hi = (P + sk) * (1.0 + delta);
                if (lo >= m0 && hi < 2.0 * m0 && sk <= 0.25 * m0)
                    cls = E;
            }
        }
        tile_E[t0 + k] = cls;
        if (spec && cls != PAPR_EXACT_AMBIG && cls != PAPR_EXACT_ZERO && cls != spec[t0 + k]) {
            const uint32_t pos = atomicAdd(redo_count, 1u);
            if (pos < redo_cap)
                redo_list[pos] = (uint32_t)(t0 + k);
        }
        if (cls == PAPR_EXACT_AMBIG && ambig_list) {  // re-streamed shards: remember which tiles to capture raw
            const uint32_t pos = atomicAdd(ambig_count, 1u);
            if (pos < ambig_cap)
                ambig_list[pos] = (uint32_t)(t0 + k);
        }
        P += sk;
    }
}

// ---- per-segment rounding functions --------------------------------------------------
// One wave per 1024-sample segment.  Coalesced 16-byte loads, then a padded LDS
// transpose so that lane l holds samples 16 l .. 16 l + 15 of the segment, in order.
//
// CCDF = true fuses pass 2 into the same sweep (one 8 B/sample read for both): every
// power is also binned against the level table staged in LDS, exactly as
// papr_ccdf_kernel does it.  The caller runs this with the thresholds derived from the
// tree sum and re-runs plain pass 2 only in the rare case the exact sum moves a
// threshold (papr_hip_ccdf_exact).
template <bool CCDF, int BLOCK>
__global__ __launch_bounds__(BLOCK) void papr_exact_seg_kernel(const float4 *__restrict__ data, uint64_t nsegs,
                                                              const int32_t *__restrict__ tile_E,
                                                              double2 *__restrict__ seg_D,
                                                              const float2 *__restrict__ tail, uint32_t tail_samples,
                                                              const uint32_t *__restrict__ table, papr_ccdf_params P,
                                                              unsigned long long *__restrict__ ghist,
                                                              const uint32_t *__restrict__ tile_list,
                                                              const uint32_t *__restrict__ tile_count, uint32_t list_cap,
                                                              uint32_t compact)
{
    constexpr int kWaves = BLOCK / kWave;
    // list form (the one-read sweep's redo pass): only the two segments of each listed tile; `compact`: the samples
    // of the k-th listed tile are the k-th tile of `data` (a streamed shard: the tiles were read back from the file)
    if (tile_list)
        nsegs = 2ull * min(*tile_count, list_cap);
    auto seg_of = [&](uint64_t i) -> uint64_t { return tile_list ? 2ull * tile_list[i >> 1] + (i & 1) : i; };
    auto src_of = [&](uint64_t i, uint64_t seg) -> uint64_t { return compact ? i : seg; };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *lds = reinterpret_cast<float4 *>(smem);                    // kWaves x (64 runs x 8 float4)
    uint32_t *tab = reinterpret_cast<uint32_t *>(lds + kWaves * kSegF4);
    uint32_t *hist = tab + P.table_words;
    const uint32_t nbins = P.nkeys + 1;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    float4 *mine = lds + wave * kSegF4;
    uint32_t *my_hist = hist;
    if constexpr (CCDF) {
        for (uint32_t k = threadIdx.x; k < P.table_words; k += BLOCK)
            tab[k] = table[k];
        for (uint32_t k = threadIdx.x; k < P.copies * nbins; k += BLOCK)
            hist[k] = 0;
        __syncthreads();
        my_hist = hist + (wave % P.copies) * nbins;
    }
    const uint2 *lut = reinterpret_cast<const uint2 *>(tab);
    auto count = [&](float pw) {
        if constexpr (CCDF) {
            const uint32_t k = lut_bin(__float_as_uint(pw), lut, P);
            if (k)
                atomicAdd(&my_hist[k], 1u);
        }
    };

    const uint64_t nwaves = (uint64_t)gridDim.x * kWaves;
    // software pipeline: the next segment's loads are in flight while this one is reduced
    uint64_t item = (uint64_t)blockIdx.x * kWaves + wave;
    float4 x[kRows], nx[kRows];
    int curE = PAPR_EXACT_ZERO, nextE = PAPR_EXACT_ZERO;
    uint64_t seg = 0, nseg = 0;
    if (item < nsegs) {
        seg = seg_of(item);
        curE = tile_E[seg >> 1];
        const float4 *p = data + src_of(item, seg) * kSegF4 + lane;
#pragma unroll
        for (int r = 0; r < kRows; r++)
            x[r] = load16<true>(p + r * kWave);
    }
    for (; item < nsegs; item += nwaves, seg = nseg) {
        if (item + nwaves < nsegs) {
            nseg = seg_of(item + nwaves);
            nextE = tile_E[nseg >> 1];  // fetched a whole iteration before it is needed
            const float4 *p = data + src_of(item + nwaves, nseg) * kSegF4 + lane;
#pragma unroll
            for (int r = 0; r < kRows; r++)
                nx[r] = load16<true>(p + r * kWave);
        }
        const int E = __builtin_amdgcn_readfirstlane(curE);
        if (E != PAPR_EXACT_AMBIG && E != PAPR_EXACT_ZERO) {
#pragma unroll
            for (int r = 0; r < kRows; r++) {
                const int f = r * kWave + lane;  // float4 slot within the segment, file order
                mine[transpose_slot(f >> 3, f & 7)] = x[r];
            }
            // same wave wrote and reads: LDS operations of one wave complete in order
            const double m0 = two_pow(E), m1 = m0 + two_pow(E - 52);
            double x0 = m0, x1 = m1;
#pragma unroll
            for (int k = 0; k < kRows; k++) {
                const float4 y = mine[transpose_slot(lane, k)];
                const float p0 = power_of(y.x, y.y), p1 = power_of(y.z, y.w);
                const double v0 = (double)p0, v1 = (double)p1;
                x0 += v0;
                x1 += v0;
                x0 += v1;
                x1 += v1;
                count(p0);
                count(p1);
            }
            Pair f;
            f.d0 = x0 - m0;  // exact: multiples of u inside the binade
            f.d1 = x1 - m1;
            f = wave_compose(f, m0);
            if (lane == 0)
                seg_D[seg] = make_double2(f.d0, f.d1);
        } else if constexpr (CCDF) {
            // no rounding function for this tile (added raw on the host, or all zeros): binning only
#pragma unroll
            for (int r = 0; r < kRows; r++) {
                count(power_of(x[r].x, x[r].y));
                count(power_of(x[r].z, x[r].w));
            }
        }
#pragma unroll
        for (int r = 0; r < kRows; r++)
            x[r] = nx[r];
        curE = nextE;
    }
    if constexpr (CCDF) {
        if (blockIdx.x == gridDim.x - 1)
            for (uint32_t k = threadIdx.x; k < tail_samples; k += BLOCK) {
                const float2 t = tail[k];
                count(power_of(t.x, t.y));
            }
        hist_flush<BLOCK>(hist, nbins, P.copies, ghist);
    }
}

// ---- pre-composition of 128-tile groups -------------------------------------------------
// One wave per group: if every non-zero tile of the group is safe in the SAME binade the
// group collapses to one pair, otherwise it is flagged for tile-by-tile handling on the host.
__global__ __launch_bounds__(256) void papr_exact_group_kernel(const int32_t *__restrict__ tile_E, uint64_t ntiles,
                                                                const double2 *__restrict__ seg_D, uint64_t ngroups,
                                                                papr_exact_group *__restrict__ out,
                                                                papr_exact_group *__restrict__ program_table)
{
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint64_t g = (uint64_t)blockIdx.x * (256 / kWave) + wave;
    if (g >= ngroups)
        return;
    const uint64_t tile0 = g * PAPR_EXACT_GROUP_TILES + 2 * (uint64_t)lane;
    int32_t e[2];
    e[0] = tile0 < ntiles ? tile_E[tile0] : PAPR_EXACT_ZERO;
    e[1] = tile0 + 1 < ntiles ? tile_E[tile0 + 1] : PAPR_EXACT_ZERO;
    const bool bad = e[0] == PAPR_EXACT_AMBIG || e[1] == PAPR_EXACT_AMBIG ||
                     (e[0] != PAPR_EXACT_ZERO && e[1] != PAPR_EXACT_ZERO && e[0] != e[1]);
    const int32_t mine = e[0] != PAPR_EXACT_ZERO ? e[0] : e[1];
    const unsigned long long has = __ballot(mine != PAPR_EXACT_ZERO);
    papr_exact_group rec;
    rec.E = PAPR_EXACT_ZERO;
    rec.pad = 0;
    rec.D0 = rec.D1 = 0.0;
    if (has != 0) {
        const int first = __ffsll((long long)has) - 1;
        const int32_t Eg = __shfl(mine, first, kWave);
        const bool mixed = __any(bad || (mine != PAPR_EXACT_ZERO && mine != Eg));
        if (mixed) {
            rec.E = PAPR_EXACT_AMBIG;
        } else {
            const double m0 = two_pow(Eg);
            Pair f = {0.0, 0.0};
            for (int k = 0; k < 4; k++) {
                if (e[k >> 1] == PAPR_EXACT_ZERO)
                    continue;
                const double2 d = seg_D[2 * tile0 + k];
                Pair gk = {d.x, d.y};
                f = compose(f, gk, m0);
            }
            f = wave_compose(f, m0);
            rec.E = Eg;
            rec.D0 = f.d0;
            rec.D1 = f.d1;
        }
    }
    if (lane == 0) {
        out[g] = rec;
        if (program_table)  // (the sum program's group table, where the program is gathered: 24 bytes the pack kernel need not copy)
            program_table[g] = rec;
    }
}

// ---- re-streamed shards: keep the unprovable tiles while their chunk is on the device ------------

// ascending order of the (few) tile numbers collected by papr_exact_classify: rank sort, one workgroup
__global__ __launch_bounds__(512) void papr_exact_sort_list_kernel(const uint32_t *__restrict__ list,
                                                                    const uint32_t *__restrict__ count, uint32_t cap,
                                                                    uint32_t *__restrict__ sorted)
{
    const uint32_t n = min(*count, cap);
    for (uint32_t i = threadIdx.x; i < n; i += 512) {
        const uint32_t v = list[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; j++)
            rank += list[j] < v;
        sorted[rank] = v;  // tile numbers are distinct
    }
}

// copy the listed tiles that lie in the chunk now staged on the device into raw_store[k]
__global__ __launch_bounds__(256) void papr_exact_capture_kernel(const float *__restrict__ chunk, uint64_t chunk_tile0,
                                                                  uint64_t chunk_ntiles,
                                                                  const uint32_t *__restrict__ sorted,
                                                                  const uint32_t *__restrict__ count, uint32_t cap,
                                                                  float *__restrict__ raw_store)
{
    const uint32_t k = blockIdx.x;
    if (k >= min(*count, cap))
        return;
    const uint64_t t = sorted[k];
    if (t < chunk_tile0 || t >= chunk_tile0 + chunk_ntiles)
        return;
    const unsigned long long *src =
        reinterpret_cast<const unsigned long long *>(chunk + 2 * (t - chunk_tile0) * PAPR_EXACT_TILE_SAMPLES);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(raw_store + 2 * (uint64_t)k * PAPR_EXACT_TILE_SAMPLES);
    for (uint32_t i = threadIdx.x; i < PAPR_EXACT_TILE_SAMPLES; i += 256)
        dst[i] = src[i];
}

// ---- program assembly on the device ---------------------------------------------------------
// The sum program (papr_exact_format.h) is gathered by two small kernels straight into mapped
// pinned host memory, so the host does one stream synchronisation instead of ~50 small copies.

// ordered lists of the mixed groups and of their unprovable tiles
// (a device function: every workgroup of papr_exact_pack_kernel makes the plan for itself, into LDS — a kernel of its own
// for it was 13 us of one workgroup's dependent loops plus a launch in the exact-sum step's chain of small kernels)
// Ordered compaction through a bit map in LDS: the flags are gathered with coalesced loads, several in flight per thread
// (one thread walking its own range of the group table was twenty dependent trips to the L2 per pass, 25 us of the
// kernel's 35), then every thread owns a few words of the map, counts them, and — behind one exclusive scan — writes out
// the positions of its set bits in ascending order.
constexpr uint32_t kPlanWords = 8192;  // 262 144 groups: a 512 GiB shard
template <typename F>
__device__ __forceinline__ uint32_t plan_compact(const uint32_t *bits, uint32_t nwords, uint32_t *sh_scan, uint32_t cap, F emit)
{
    const uint32_t per = (nwords + 255u) / 256u;
    const uint32_t w0 = threadIdx.x * per, w1 = min(w0 + per, nwords);
    uint32_t cnt = 0;
    for (uint32_t w = w0; w < w1; w++)
        cnt += (uint32_t)__popc(bits[w]);
    uint32_t total = 0;
    uint32_t pos = block_exclusive_scan<uint32_t, 256>(cnt, sh_scan, &total);
    for (uint32_t w = w0; w < w1; w++) {
        uint32_t m = bits[w];
        while (m) {
            const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
            m &= m - 1u;
            if (pos < cap)
                emit(pos, w * 32u + b);
            pos++;
        }
    }
    return total;
}

__device__ __forceinline__ void exact_plan(const papr_exact_group *__restrict__ groups, uint64_t ngroups,
                                           const int32_t *__restrict__ tile_E, uint64_t ntiles, uint32_t *mixed_list,
                                           uint32_t cap_mixed, uint32_t *raw_list, uint32_t cap_raw, papr_exact_plan *out)
{
    __shared__ uint32_t sh[256 / kWave];
    __shared__ uint32_t bits[kPlanWords];
    const uint32_t t = threadIdx.x;
    constexpr int CH = 8;
    // ---- mixed groups ----
    const uint64_t ng = min(ngroups, (uint64_t)kPlanWords * 32u);  // (beyond that: reported as overflow below)
    const uint32_t gwords = (uint32_t)((ng + 31) / 32);
    for (uint32_t w = t; w < gwords; w += 256)
        bits[w] = 0;
    __syncthreads();
    for (uint64_t g0 = t; g0 < ng; g0 += 256ull * CH) {
        int32_t e[CH];
#pragma unroll
        for (int k = 0; k < CH; k++)
            e[k] = g0 + 256ull * k < ng ? groups[g0 + 256ull * k].E : PAPR_EXACT_ZERO;
#pragma unroll
        for (int k = 0; k < CH; k++)
            if (e[k] == PAPR_EXACT_AMBIG) {
                const uint64_t g = g0 + 256ull * k;
                atomicOr(&bits[g >> 5], 1u << (g & 31));
            }
    }
    __syncthreads();
    const uint32_t nmixed_all = plan_compact(bits, gwords, sh, cap_mixed, [&](uint32_t pos, uint32_t g) { mixed_list[pos] = g; });
    const uint32_t nmixed = min(nmixed_all, cap_mixed);
    __syncthreads();
    // ---- their tiles that have to travel raw ----
    const uint32_t items = nmixed * PAPR_EXACT_GROUP_TILES;  // (<= 256 * 128 bits: 1024 words)
    const uint32_t iwords = (items + 31) / 32;
    for (uint32_t w = t; w < iwords; w += 256)
        bits[w] = 0;
    __syncthreads();
    auto tile_of = [&](uint32_t item) -> uint64_t {
        return (uint64_t)mixed_list[item / PAPR_EXACT_GROUP_TILES] * PAPR_EXACT_GROUP_TILES + item % PAPR_EXACT_GROUP_TILES;
    };
    for (uint32_t i0 = t; i0 < items; i0 += 256u * CH) {
        int32_t e[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const uint32_t it = i0 + 256u * k;
            const uint64_t tile = it < items ? tile_of(it) : ntiles;
            e[k] = tile < ntiles ? tile_E[tile] : PAPR_EXACT_ZERO;
        }
#pragma unroll
        for (int k = 0; k < CH; k++)
            if (e[k] == PAPR_EXACT_AMBIG) {
                const uint32_t it = i0 + 256u * k;
                atomicOr(&bits[it >> 5], 1u << (it & 31));
            }
    }
    __syncthreads();
    const uint32_t nraw_all = plan_compact(bits, iwords, sh, cap_raw, [&](uint32_t pos, uint32_t it) { raw_list[pos] = (uint32_t)tile_of(it); });
    if (t == 0) {
        out->nmixed = nmixed;
        out->nraw = min(nraw_all, cap_raw);
        out->overflow = (nmixed_all > cap_mixed || nraw_all > cap_raw || ngroups > ng) ? 1u : 0u;
        out->pad = 0;
    }
    __syncthreads();
}

// one workgroup per piece of the program: group-table chunks, mixed records, raw tiles, header + tail
__global__ __launch_bounds__(256) void papr_exact_pack_kernel(papr_exact_plan *__restrict__ plan_out,
                                                               uint32_t *__restrict__ mixed_list_out,
                                                               uint32_t *__restrict__ raw_list_out,
                                                               const papr_exact_group *__restrict__ groups,
                                                               uint64_t ngroups, const int32_t *__restrict__ tile_E,
                                                               uint64_t ntiles, const double *__restrict__ seg_D,
                                                               const float *__restrict__ data,
                                                               const float *__restrict__ raw_store,
                                                               const float *__restrict__ tail_src, uint64_t nsamples,
                                                               uint32_t tail_samples, uint32_t group_blocks,
                                                               uint32_t cap_mixed, uint32_t cap_raw,
                                                               unsigned char *__restrict__ out,
                                                               const uint32_t *__restrict__ count_src,
                                                               uint32_t *__restrict__ count_dst, uint64_t out_cap,
                                                               uint32_t redo_cap, const papr_exact_prefix_src prefix)
{
    __shared__ uint32_t mixed_list[256], raw_list[1024];  // (kCapMixed, kCapRaw)
    __shared__ papr_exact_plan plan_sh;
    papr_exact_plan *plan = &plan_sh;
    exact_plan(groups, ngroups, tile_E, ntiles, mixed_list, cap_mixed < 256u ? cap_mixed : 256u, raw_list,
               cap_raw < 1024u ? cap_raw : 1024u, plan);
    if (blockIdx.x + 1 == gridDim.x) {  // (the plan also goes where the host-side fallbacks look for it)
        for (uint32_t k = threadIdx.x; k < plan->nmixed; k += 256)
            mixed_list_out[k] = mixed_list[k];
        for (uint32_t k = threadIdx.x; k < plan->nraw; k += 256)
            raw_list_out[k] = raw_list[k];
        if (threadIdx.x == 0)
            *plan_out = plan_sh;
    }
    const uint32_t nmixed = plan->nmixed, nraw = plan->nraw;
    const size_t off_groups = 48;  // sizeof(papr_exact_header)
    const size_t off_mixed = off_groups + ngroups * 24;
    const size_t off_raw = off_mixed + (size_t)nmixed * 4616;
    const size_t off_tail = off_raw + (size_t)nraw * kRawRecBytes;
    // a bounded destination (the slot of the in-stream program exchange): a program that does not fit leaves only its
    // header, marked — every rank sees that and all of them take the host path
    const bool fits = out_cap == 0 || off_tail + (size_t)tail_samples * 8 <= out_cap;
    if (!fits && blockIdx.x + 1 != gridDim.x)
        return;
    uint32_t b = blockIdx.x;
    if (b < group_blocks) {  // group table, 8 bytes per thread per step
        const uint64_t words = ngroups * 3;
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(groups);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(out + off_groups);
        for (uint64_t k = (uint64_t)b * 256 + threadIdx.x; k < words; k += (uint64_t)group_blocks * 256)
            dst[k] = src[k];
        return;
    }
    b -= group_blocks;
    if (b < cap_mixed) {
        if (b >= nmixed)
            return;
        const uint64_t g = mixed_list[b], t0 = g * PAPR_EXACT_GROUP_TILES;
        unsigned char *rec = out + off_mixed + (size_t)b * 4616;
        if (threadIdx.x == 0)
            *reinterpret_cast<unsigned long long *>(rec) = g;
        int32_t *te = reinterpret_cast<int32_t *>(rec + 8);
        for (uint32_t j = threadIdx.x; j < PAPR_EXACT_GROUP_TILES; j += 256)
            te[j] = t0 + j < ntiles ? tile_E[t0 + j] : PAPR_EXACT_ZERO;
        double *sd = reinterpret_cast<double *>(rec + 8 + 4 * PAPR_EXACT_GROUP_TILES);
        for (uint32_t j = threadIdx.x; j < 4 * PAPR_EXACT_GROUP_TILES; j += 256)
            sd[j] = t0 + j / 4 < ntiles ? seg_D[4 * t0 + j] : 0.0;
        return;
    }
    b -= cap_mixed;
    if (b < cap_raw) {
        if (b >= nraw)
            return;
        const uint64_t t = raw_list[b];
        unsigned char *rec = out + off_raw + (size_t)b * kRawRecBytes;
        if (threadIdx.x == 0)
            *reinterpret_cast<unsigned long long *>(rec) = t;
        // resident shard: the tile itself; re-streamed shard: the copy captured while its chunk was staged
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(
            raw_store ? raw_store + 2 * (uint64_t)b * PAPR_EXACT_TILE_SAMPLES : data + 2 * t * PAPR_EXACT_TILE_SAMPLES);
        float *dst_pw = reinterpret_cast<float *>(rec + 8);  // (the powers travel, not the samples: half the bytes)
        // ---- the tile's 16-sample runs, each with the pair of the binade the running sum is in when it gets there ----
        // The sum changes binade somewhere in this tile (or may), so the tile as a whole has no pair — but 127 of its 128
        // runs do, and an approximate prefix says for which binade: the host applies a run's pair when the sum is in
        // that binade before and after it and adds the run sample by sample when not, so nothing here has to be
        // exact but the pair itself (the additions themselves, from the binade's two canonical entry states).
        int32_t *run_E = reinterpret_cast<int32_t *>(rec + 8 + 4 * PAPR_EXACT_TILE_SAMPLES);
        double *run_D = reinterpret_cast<double *>(rec + 8 + 4 * PAPR_EXACT_TILE_SAMPLES + 4 * kTileRuns);
        __shared__ double sh_red[256 / kWave];
        double P_tile = 0.0;
        if (prefix.kind) {
            const uint64_t blk = t / kTilesPerBlock;
            double a = 0.0;
            for (uint64_t k = threadIdx.x; k < blk; k += 256)
                a += prefix.block_sums[k];
            for (uint64_t k = blk * kTilesPerBlock + threadIdx.x; k < t; k += 256)
                a += prefix.kind == 1 ? tile_sum_of<true>(prefix.sums, k) : tile_sum_of<false>(prefix.sums, k);
            a = wave_reduce_sum(a);
            if ((threadIdx.x & (kWave - 1)) == 0)
                sh_red[threadIdx.x / kWave] = a;
            __syncthreads();
            P_tile = (prefix.before_dev ? *prefix.before_dev : prefix.before) + ((sh_red[0] + sh_red[1]) + (sh_red[2] + sh_red[3]));
        }
        // (coalesced: thread t takes samples t, t + 256, ...; the run's owner reads its sixteen powers back out of LDS)
        __shared__ float sh_pw[PAPR_EXACT_TILE_SAMPLES];
        for (uint32_t k = threadIdx.x; k < PAPR_EXACT_TILE_SAMPLES; k += 256) {
            const float2 v = reinterpret_cast<const float2 *>(src)[k];
            const float p = power_of(v.x, v.y);
            sh_pw[k] = p;
            dst_pw[k] = p;
        }
        __syncthreads();
        const uint32_t r = threadIdx.x;  // (threads 128 .. 255 only take part in the scan)
        float pw[kRunSamples];
        double s = 0.0;
        if (r < kTileRuns) {
#pragma unroll
            for (int k = 0; k < kRunSamples; k++) {
                pw[k] = sh_pw[r * kRunSamples + k];
                s += (double)pw[k];
            }
        }
        __shared__ double sh_scan[256 / kWave];
        const double P = P_tile + block_exclusive_scan<double, 256>(s, sh_scan, (double *)nullptr);
        if (r < kTileRuns) {
            int32_t cls = PAPR_EXACT_AMBIG;
            double d0 = 0.0, d1 = 0.0;
            if (s == 0.0) {
                cls = PAPR_EXACT_ZERO;  // (sixteen + 0.0)
            } else if (prefix.kind && s > 0.0 && s < 1.0e300 && P > 0.0 && P < 1.0e300) {
                const int biased = (int)((__double_as_longlong(P) >> 52) & 0x7ff);
                const int E = biased - 1023;
                if (biased != 0 && E >= -960) {
                    const double m0 = two_pow(E), m1 = m0 + two_pow(E - 52);
                    double x0 = m0, x1 = m1;
#pragma unroll
                    for (int k = 0; k < kRunSamples; k++) {
                        const double v = (double)pw[k];
                        x0 += v;
                        x1 += v;
                    }
                    if (x1 < 2.0 * m0) {  // both chains stayed inside the binade: one ulp throughout
                        cls = E;
                        d0 = x0 - m0;
                        d1 = x1 - m1;
                    }
                }
            }
            run_E[r] = cls;
            run_D[2 * r] = d0;
            run_D[2 * r + 1] = d1;
        }
        return;
    }
    // last workgroup: the tail samples and the header (layout of papr_exact_header)
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(tail_src);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(out + off_tail);
    for (uint32_t k = threadIdx.x; fits && k < tail_samples; k += 256)
        dst[k] = src[k];
    if (threadIdx.x == 0) {
        uint32_t *h32 = reinterpret_cast<uint32_t *>(out);
        unsigned long long *h64 = reinterpret_cast<unsigned long long *>(out);
        h32[0] = 0x31535850u;  // PAPR_EXACT_MAGIC
        h32[1] = 3u;           // PAPR_EXACT_VERSION
        h64[1] = nsamples;
        h64[2] = ntiles;
        h64[3] = ngroups;
        h32[8] = tail_samples;
        h32[9] = nmixed;
        h32[10] = nraw;
        // reserved word: non-zero = do not use this program (1: the lists were truncated; 2: it did not fit its slot;
        // 4: more tiles to redo than the list holds — the pairs are not final)
        h32[11] = plan->overflow | (fits ? 0u : 2u) | ((redo_cap && count_src && *count_src > redo_cap) ? 4u : 0u);
        if (count_dst)
            *count_dst = *count_src;  // (the redo count, to mapped host memory with the program: saves a D2H copy)
    }
}

// ---- launch wrappers -------------------------------------------------------------------------

void papr_launch_exact_classify(hipStream_t st, const double *tile_wave_sums, uint64_t ntiles, double *block_sums,
                                double before, double delta, int32_t *tile_E, uint32_t *ambig_list, uint32_t ambig_cap,
                                uint32_t *ambig_count, uint32_t *ambig_sorted)
{
    const uint32_t nb = (uint32_t)((ntiles + kTilesPerBlock - 1) / kTilesPerBlock);
    if (nb == 0)
        return;
    hipLaunchKernelGGL(papr_exact_block_sums<false>, dim3(nb), dim3(256), 0, st, tile_wave_sums, ntiles, block_sums,
                       (uint32_t *)nullptr);
    hipLaunchKernelGGL(papr_exact_classify<false>, dim3(nb), dim3(256), 0, st, tile_wave_sums, ntiles, block_sums, delta,
                       tile_E, ambig_list, ambig_cap, ambig_count, (const int32_t *)nullptr, (uint32_t *)nullptr, 0u,
                       (uint32_t *)nullptr, (const unsigned long long *)nullptr, 0ull, before, (const double *)nullptr);
    if (ambig_list)
        hipLaunchKernelGGL(papr_exact_sort_list_kernel, dim3(1), dim3(512), 0, st, ambig_list, ambig_count, ambig_cap,
                           ambig_sorted);
}

// One-read sweep: the tile sums are D0 of the segments' pairs (seg_D), and every tile whose proven binade differs
// from the speculated one (`spec`) is listed for papr_launch_exact_redo.  *redo_count is zeroed here.
void papr_launch_exact_classify_swept(hipStream_t st, const void *seg_D, uint64_t ntiles, double *block_sums, double before,
                                      double delta, int32_t *tile_E, const int32_t *spec, uint32_t *redo_list,
                                      uint32_t redo_cap, uint32_t *redo_count, uint32_t *ambig_list, uint32_t ambig_cap,
                                      uint32_t *ambig_count, uint32_t *ambig_sorted, const double *before_dev,
                                      const unsigned long long *n_total_dev, uint64_t n_shard)
{
    const uint32_t nb = (uint32_t)((ntiles + kTilesPerBlock - 1) / kTilesPerBlock);
    if (nb == 0)
        return;
    const double *sums = (const double *)seg_D;
    hipLaunchKernelGGL(papr_exact_block_sums<true>, dim3(nb), dim3(256), 0, st, sums, ntiles, block_sums, redo_count);
    hipLaunchKernelGGL(papr_exact_classify<true>, dim3(nb), dim3(256), 0, st, sums, ntiles, block_sums, delta, tile_E,
                       ambig_list, ambig_cap, ambig_count, spec, redo_list, redo_cap, redo_count, n_total_dev,
                       (unsigned long long)n_shard, before, before_dev);
    if (ambig_list)  // a streamed shard: the unprovable tiles, ascending (they are read back from the file for the program)
        hipLaunchKernelGGL(papr_exact_sort_list_kernel, dim3(1), dim3(512), 0, st, ambig_list, ambig_count, ambig_cap,
                           ambig_sorted);
}

// the segment sums a one-read sweep left (D0 of either segment's pair) in the layout pass 1 writes its per-tile wave
// sums in, so that papr_launch_exact_classify can take over from them (same order of addition: s0 + s1)
__global__ __launch_bounds__(256) void papr_exact_segsums_to_tilesums_kernel(const double *__restrict__ seg_D, uint64_t ntiles,
                                                                             double *__restrict__ tws)
{
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < ntiles; t += (uint64_t)gridDim.x * 256) {
        tws[t * PAPR_EXACT_TILE_WAVES + 0] = seg_D[4 * t];
        tws[t * PAPR_EXACT_TILE_WAVES + 1] = seg_D[4 * t + 2];
        for (int k = 2; k < PAPR_EXACT_TILE_WAVES; k++)
            tws[t * PAPR_EXACT_TILE_WAVES + k] = 0.0;
    }
}

void papr_launch_exact_segsums_to_tilesums(hipStream_t st, const void *seg_D, uint64_t ntiles, double *tile_wave_sums)
{
    if (ntiles == 0)
        return;
    const int blocks = (int)std::min<uint64_t>((ntiles + 255) / 256, 4096);
    hipLaunchKernelGGL(papr_exact_segsums_to_tilesums_kernel, dim3(blocks), dim3(256), 0, st, (const double *)seg_D, ntiles,
                       tile_wave_sums);
}

void papr_launch_exact_capture(hipStream_t st, const void *chunk, uint64_t chunk_tile0, uint64_t chunk_ntiles,
                               const uint32_t *sorted, const uint32_t *count, uint32_t cap, void *raw_store)
{
    hipLaunchKernelGGL(papr_exact_capture_kernel, dim3(cap), dim3(256), 0, st, (const float *)chunk, chunk_tile0,
                       chunk_ntiles, sorted, count, cap, (float *)raw_store);
}

// workgroup sizes: 4 waves for the plain sweep; 8 waves for the fused one so that the level table is
// shared by more waves and two workgroups (16 waves) fit a CU's 160 KiB of LDS
constexpr int kSegBlock = 256, kFusedBlock = 512;

void papr_exact_prepare_device(void)  // per device: the fused sweep asks for > 64 KiB of dynamic LDS
{
    (void)hipFuncSetAttribute((const void *)papr_exact_seg_kernel<true, kFusedBlock>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
}

size_t papr_exact_transpose_lds_bytes(void)
{
    return (size_t)(kFusedBlock / kWave) * kSegF4 * sizeof(float4);
}

int papr_exact_fused_waves(void)
{
    return kFusedBlock / kWave;
}

void papr_launch_exact_segments(hipStream_t st, int blocks, const void *data, uint64_t nsegs, const int32_t *tile_E,
                                void *seg_D)
{
    if (nsegs == 0)
        return;
    papr_ccdf_params none;
    memset(&none, 0, sizeof(none));
    hipLaunchKernelGGL((papr_exact_seg_kernel<false, kSegBlock>), dim3(blocks), dim3(kSegBlock),
                       (size_t)(kSegBlock / kWave) * kSegF4 * sizeof(float4), st, (const float4 *)data, nsegs, tile_E,
                       (double2 *)seg_D, (const float2 *)nullptr, 0u, (const uint32_t *)nullptr, none,
                       (unsigned long long *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u, 0u);
}

// the rounding functions of the LISTED tiles only (count read on the device: no host round trip in between)
void papr_launch_exact_redo(hipStream_t st, int blocks, const void *data, const int32_t *tile_E, void *seg_D,
                            const uint32_t *tile_list, const uint32_t *tile_count, uint32_t list_cap, int compact)
{
    papr_ccdf_params none;
    memset(&none, 0, sizeof(none));
    hipLaunchKernelGGL((papr_exact_seg_kernel<false, kSegBlock>), dim3(blocks), dim3(kSegBlock),
                       (size_t)(kSegBlock / kWave) * kSegF4 * sizeof(float4), st, (const float4 *)data, (uint64_t)0, tile_E,
                       (double2 *)seg_D, (const float2 *)nullptr, 0u, (const uint32_t *)nullptr, none,
                       (unsigned long long *)nullptr, tile_list, tile_count, list_cap, compact ? 1u : 0u);
}

// ---- one-read sweep: speculated binades from the mean estimate's per-group sums ----------------------------
// group g of the estimate = tiles [g * ratio, (g + 1) * ratio); group_sums[4 g .. 4 g + 3] (one per wave of the
// estimate kernel) add up to the sum of ONE tile's worth of its rows, so `scale` (= ratio) times that estimates
// the group's sum.  Exclusive scan over the groups (one workgroup) ...
__global__ __launch_bounds__(1024) void papr_exact_spec_scan_kernel(const double *__restrict__ group_sums, uint64_t ngroups,
                                                                    double scale, double before,
                                                                    double *__restrict__ group_prefix,
                                                                    const double *__restrict__ before_dev)
{
    if (before_dev)
        before = *before_dev;  // (peers: the estimated sum in front of this shard, from every shard's estimate record)
    __shared__ double sh[1024 / kWave];
    papr_exact_spec_scan_body(group_sums, ngroups, scale, before, group_prefix, sh);
}

// ... then per tile: the binade of the estimated prefix if the estimated tile lies well inside it, else none.
// A wrong guess costs a redo of that tile, never a result (papr_exact_classify<true> checks every tile against
// the TRUE prefix afterwards).
__global__ __launch_bounds__(256) void papr_exact_spec_tiles_kernel(const double *__restrict__ group_sums,
                                                                    const double *__restrict__ group_prefix,
                                                                    uint64_t ngroups, uint32_t ratio, double scale,
                                                                    uint64_t ntiles, int32_t *__restrict__ spec)
{
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < ntiles; t += (uint64_t)gridDim.x * 256) {
        uint64_t g = t / ratio;
        if (g >= ngroups)
            g = ngroups - 1;  // the tiles behind the last whole group: carried on at its rate
        const double per_tile = (((group_sums[4 * g] + group_sums[4 * g + 1]) + group_sums[4 * g + 2]) + group_sums[4 * g + 3]) *
                                scale / (double)ratio;
        const double P = group_prefix[g] + (double)(t - g * ratio) * per_tile;
        int32_t cls = PAPR_EXACT_AMBIG;
        if (P > 0.0 && P < 1.0e300 && per_tile >= 0.0 && per_tile < 1.0e300) {
            const int biased = (int)((__double_as_longlong(P) >> 52) & 0x7ff);
            const int E = biased - 1023;
            if (biased != 0 && E >= -960) {
                const double m0 = two_pow(E);
                if (P + per_tile < 2.0 * m0 && per_tile <= 0.25 * m0)
                    cls = E;
            }
        }
        spec[t] = cls;
    }
}

__global__ __launch_bounds__(256) void papr_exact_fill_spec_kernel(int32_t *__restrict__ spec, uint64_t ntiles, int32_t E)
{
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < ntiles; t += (uint64_t)gridDim.x * 256)
        spec[t] = E;
}

void papr_launch_exact_spec(hipStream_t st, const double *group_sums, uint64_t ngroups, uint32_t ratio, double scale,
                            double before, double *group_prefix, uint64_t ntiles, int32_t *spec, const double *before_dev,
                            bool scan_done)
{
    if (ntiles == 0)
        return;
    const int blocks = (int)std::min<uint64_t>((ntiles + 255) / 256, 1024);
    if (ngroups == 0) {
        hipLaunchKernelGGL(papr_exact_fill_spec_kernel, dim3(blocks), dim3(256), 0, st, spec, ntiles,
                           (int32_t)PAPR_EXACT_AMBIG);
        return;
    }
    if (!scan_done)  // (else papr_guess_bands_kernel's second workgroup did it, beside the guess)
        hipLaunchKernelGGL(papr_exact_spec_scan_kernel, dim3(1), dim3(1024), 0, st, group_sums, ngroups, scale, before,
                           group_prefix, before_dev);
    hipLaunchKernelGGL(papr_exact_spec_tiles_kernel, dim3(blocks), dim3(256), 0, st, group_sums, group_prefix, ngroups,
                       ratio, scale, ntiles, spec);
}

void papr_launch_exact_fill_spec(hipStream_t st, int32_t *spec, uint64_t ntiles, int32_t E)
{
    if (ntiles)
        hipLaunchKernelGGL(papr_exact_fill_spec_kernel, dim3((int)std::min<uint64_t>((ntiles + 255) / 256, 1024)),
                           dim3(256), 0, st, spec, ntiles, E);
}

// the same sweep with pass 2 fused in (LUT form of the level table only); lds_table_bytes = table + histograms;
// `blocks` counts kFusedBlock-thread workgroups
void papr_launch_exact_segments_ccdf(hipStream_t st, int blocks, const void *data, uint64_t nsegs,
                                     const int32_t *tile_E, void *seg_D, const void *tail, uint32_t tail_samples,
                                     const uint32_t *table, const papr_ccdf_params &P, size_t lds_table_bytes,
                                     unsigned long long *ghist)
{
    hipLaunchKernelGGL((papr_exact_seg_kernel<true, kFusedBlock>), dim3(blocks), dim3(kFusedBlock),
                       papr_exact_transpose_lds_bytes() + lds_table_bytes, st, (const float4 *)data, nsegs, tile_E,
                       (double2 *)seg_D, (const float2 *)tail, tail_samples, table, P, ghist, (const uint32_t *)nullptr,
                       (const uint32_t *)nullptr, 0u, 0u);
}

void papr_launch_exact_groups(hipStream_t st, const int32_t *tile_E, uint64_t ntiles, const void *seg_D,
                              uint64_t ngroups, papr_exact_group *out, unsigned char *program)
{
    if (ngroups == 0)
        return;
    const uint32_t nb = (uint32_t)((ngroups + 3) / 4);
    hipLaunchKernelGGL(papr_exact_group_kernel, dim3(nb), dim3(256), 0, st, tile_E, ntiles, (const double2 *)seg_D,
                       ngroups, out, program ? reinterpret_cast<papr_exact_group *>(program + 48) : nullptr);
}

void papr_launch_exact_pack(hipStream_t st, const papr_exact_group *groups, uint64_t ngroups, const int32_t *tile_E,
                            uint64_t ntiles, const void *seg_D, const void *data, const void *raw_store,
                            const void *tail_src, uint64_t nsamples, uint32_t tail_samples, uint32_t *mixed_list,
                            uint32_t cap_mixed, uint32_t *raw_list, uint32_t cap_raw, papr_exact_plan *plan,
                            unsigned char *out_mapped, const uint32_t *count_src, uint32_t *count_dst, uint64_t out_cap,
                            uint32_t redo_cap, papr_exact_prefix_src prefix)
{
    // (groups_in_place: papr_launch_exact_groups already wrote the group table into this program)
    const uint32_t group_blocks = prefix.groups_in_place ? 0u : (uint32_t)std::min<uint64_t>(64, (ngroups * 3 + 255) / 256 + 1);
    hipLaunchKernelGGL(papr_exact_pack_kernel, dim3(group_blocks + cap_mixed + cap_raw + 1), dim3(256), 0, st, plan,
                       mixed_list, raw_list, groups, ngroups, tile_E, ntiles, (const double *)seg_D, (const float *)data,
                       (const float *)raw_store, (const float *)tail_src, nsamples, tail_samples, group_blocks, cap_mixed,
                       cap_raw, out_mapped, count_src, count_dst, out_cap, redo_cap, prefix);
}

// ---- the in-stream exchange of the sum programs (peers, single-wait step) ------------------------------------------
// After the all-gather of the ranks' fixed-size slots: one workgroup per rank copies the USED bytes of that rank's slot
// into the same slot of a mapped host buffer (the host replays all programs in rank order while the stream goes on with
// the recount).  A slot whose header is marked (or is no header at all) travels as its 48 header bytes.
__global__ __launch_bounds__(1024) void papr_exact_programs_to_host_kernel(const unsigned char *__restrict__ slots, const papr_xprog_layout lay,
                                                                           unsigned char *__restrict__ host)
{
    const uint64_t off = lay.offs[blockIdx.x], slot_bytes = lay.offs[blockIdx.x + 1] - off;
    const unsigned char *src = slots + off;
    unsigned char *dst = host + off;
    const uint32_t *h32 = reinterpret_cast<const uint32_t *>(src);
    const unsigned long long *h64 = reinterpret_cast<const unsigned long long *>(src);
    uint64_t used = 48;
    if (h32[0] == 0x31535850u && h32[11] == 0) {
        const uint64_t want = 48 + h64[3] * 24 + (uint64_t)h32[9] * 4616 + (uint64_t)h32[10] * kRawRecBytes + (uint64_t)h32[8] * 8;
        if (want <= slot_bytes)
            used = want;
    }
    const unsigned long long *s8 = reinterpret_cast<const unsigned long long *>(src);
    unsigned long long *d8 = reinterpret_cast<unsigned long long *>(dst);
    for (uint64_t k = threadIdx.x; k < used / 8; k += 1024)
        d8[k] = s8[k];
}

void papr_launch_exact_programs_to_host(hipStream_t st, const void *slots, const papr_xprog_layout &lay, void *host_mapped)
{
    hipLaunchKernelGGL(papr_exact_programs_to_host_kernel, dim3(lay.world), dim3(1024), 0, st, (const unsigned char *)slots, lay,
                       (unsigned char *)host_mapped);
}
