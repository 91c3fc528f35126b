// measure/papr_sweep_lab.hip — the laboratory of the one-sweep kernels (built by `make MEASURE=1` only; the product is
// ../papr_sweep.hip): every geometry, stash form and ablation that was measured on the way to the product kernels,
// selectable through the same variant ids (papr_hip_tuning.sweep_variant, PAPR_HIP_TUNE wvariant=) by tools/ and by the
// geometry tests —
//   papr_sweep_lab_kernel<BLOCK, U, NT, PIPE, ABL, LUT2, SMODE>  the generic form of papr_sweep_kernel: workgroup size x
//                              loads per lane x loop form x ablation bits x compact table x stash mode
//   papr_sweep_split_kernel    loader waves / binner waves
//   papr_sweep2_kernel         wave-private segments, compact table, ring stash; <EXACT>: the exact-sum sweep's first form
// What each of them showed is DESIGN.md section 4b / 5.

#include "../papr_sweep_dev.h"

namespace {

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
                // (MODE bit 7: the stores left out; bit 8: every spill over the segment's first bytes — both measurement
                // only, to tell what of a spill's cost is the HBM write)
                if constexpr ((MODE & 128) != 0)
                    asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                else if constexpr ((MODE & 256) != 0)
                    store16<(MODE & 16) ? 0 : 2>(seg + i, v);
                else if (pos + i + 4 <= seg_cap)
                    store16<(MODE & 16) ? 0 : (MODE & 512) ? 1 : 2>(seg + pos + i, v);  // (bit 9: nontemporal)
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
typedef WaveStashT<0> WaveStash0;  // (the returning-atomic form: the split kernel's binners)

}  // namespace


// ABL (measurement only): leave out one ingredient to see what it costs — 1 stash, 2 histogram,
// 4 LUT lookup, 8 trackers, 16 sum, 32 spill check.  The results of such a launch are meaningless.
// LUT2: the compact band-edge table of papr_kernels.h (two edges per cell: 1-8 KiB instead of 32-40), which lets small
// workgroups — the geometry papr_stats_kernel runs best in — afford a table of their own.
template <int BLOCK, int U, bool NT, int PIPE, int ABL = 0, bool LUT2 = false, int SMODE = 0>
__global__ __launch_bounds__(BLOCK) void papr_sweep_lab_kernel(const float4 *__restrict__ data, uint64_t ntiles,
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
    uint32_t last_epoch = 0;
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
        if constexpr (!(ABL & 32) && !(SMODE & 64)) {
            uint32_t limit = SLICE - 2 * U * kWave - ((SMODE & 8) ? kWave : 0);  // the next tile might not fit
            // SMODE bit 10 / 11: every wave of the chip spills at the same moment — every 32nd tile / whenever the 100 MHz
            // clock enters a new 82 us epoch — so that the memory side sees the stash as a few large bursts of writes
            // between long stretches of pure reads instead of a trickle that turns its buses around all the time
            if constexpr ((SMODE & 1024) != 0)
                limit = (it & 31u) == 31u ? kWave : limit;
            if constexpr ((SMODE & 2048) != 0) {
                constexpr int EP_SHIFT = ((SMODE >> 12) & 3) == 0 ? 13 : ((SMODE >> 12) & 3) == 1 ? 12 : ((SMODE >> 12) & 3) == 2 ? 14 : 11;  // (bits 12-13: the epoch)
                const uint32_t ep = (uint32_t)(__builtin_amdgcn_s_memrealtime() >> EP_SHIFT);
                limit = ep != last_epoch ? kWave : limit;
                last_epoch = ep;
            }
            ws.spill_if_above(limit, (it + 1) * (uint32_t)(2 * TILE_F4));
        }
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
        WaveStash0 ws{0u, slices + bidx * SLICE, &wave_fill[bidx], stash + (uint64_t)blockIdx.x * seg_cap, &seg_fill,
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


// ---- exact-sum pairs (the algebra is papr_exact.hip's; restated here because both files keep their helpers
// in anonymous namespaces) ----
struct Pair2 {
    double d0, d1;
};
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


// Geometry variants of the sweep (ids as in papr_kernels.hip's table).
#define PAPR_FOR_EACH_SWEEP_VARIANT(X) \
    X(0, 256, 8, 0) X(1, 256, 4, 1) X(2, 256, 8, 1) X(3, 512, 8, 0) X(4, 1024, 4, 0) X(6, 512, 4, 1) X(7, 256, 4, 0) \
    X(8, 1024, 4, 1) X(9, 1024, 2, 1) X(10, 512, 2, 1) X(11, 256, 2, 1) X(12, 1024, 2, 0) X(13, 512, 4, 0)              \
    X(14, 256, 4, 2) X(15, 512, 4, 2) X(16, 256, 8, 2) X(17, 1024, 4, 2)

// the same kernel with the compact two-edges-per-cell table (small workgroups can afford a table of their own)
#define PAPR_FOR_EACH_SWEEP_LUT2_VARIANT(X)                                                                  \
    X(20, 256, 4, 1) X(21, 256, 4, 0) X(22, 512, 4, 1) X(23, 512, 4, 0) X(24, 1024, 4, 0) X(25, 256, 8, 1)    \
    X(26, 256, 2, 1) X(27, 512, 2, 1) X(28, 1024, 4, 1) X(29, 256, 8, 0)

// papr_sweep_kernel with other stash forms: id, workgroup size, loads per lane, loop form, compact table,
// stash mode (bit 0: 16-byte spills, bit 1: ballot compaction, bit 2: [bin][copy] histogram sets, bit 3: no branch and
// no exec-masked region in the per-sample code, bit 4: plain instead of write-through spill stores, bit 5: twice the
// LDS slice per wave).  40 — 512 threads x 8 loads per lane, ONE persistent workgroup per CU, branch-free ballot stash,
// 16-byte spills out of a double slice — is the default: eight waves per CU with eight 16-byte loads each in flight is
// the shape a stripped kernel reads fastest in (tools/work_probe.hip); with only two waves per SIMD nothing hides a
// v_cmp -> s_and_saveexec round per sample or a spill's store latency, hence the branch-free code and the rarer, wider
// spills.  The others did not pay and are built by `make MEASURE=1` only.
#define PAPR_FOR_EACH_SWEEP_SP16_VARIANT(X) \
    X(40, 512, 8, 0, false, 43) X(100, 512, 8, 0, false, 10)                                                              \
    X(5, 1024, 4, 0, false, 1) X(18, 1024, 4, 0, true, 1) X(19, 512, 4, 0, false, 1) X(36, 1024, 4, 0, false, 2)          \
    X(37, 1024, 4, 0, false, 3) X(38, 1024, 4, 0, true, 2) X(39, 512, 4, 0, false, 2)                                  \
    X(80, 256, 8, 0, false, 2) X(81, 256, 8, 0, false, 3) X(82, 256, 8, 1, false, 2) X(83, 512, 8, 0, false, 2)       \
    X(84, 256, 8, 0, false, 6) X(85, 256, 8, 0, false, 14) X(86, 256, 8, 0, false, 10) X(87, 512, 8, 0, false, 14)         \
    X(88, 1024, 4, 0, false, 14) X(89, 256, 8, 1, false, 14) X(101, 512, 8, 1, false, 10)                                  \
    X(102, 512, 4, 0, false, 10) X(103, 1024, 8, 0, false, 10) X(104, 512, 8, 0, false, 11) X(105, 512, 8, 0, false, 26)   \
    X(106, 512, 8, 0, false, 27) X(107, 512, 8, 0, false, 42) X(109, 512, 8, 0, false, 59)   \
    X(110, 512, 8, 1, false, 11) X(112, 256, 8, 0, false, 43) X(113, 1024, 4, 0, false, 43) \
    X(114, 512, 8, 1, false, 107) X(140, 512, 8, 1, false, 43) X(141, 512, 8, 1, false, 171) X(142, 512, 8, 1, false, 299) \
    X(143, 512, 8, 1, false, 315) X(144, 512, 8, 1, false, 555) X(145, 512, 8, 2, false, 43) X(146, 512, 8, 1, false, 1067) \
    X(147, 512, 8, 1, false, 2091) X(148, 512, 8, 1, false, 6187) X(149, 512, 8, 1, false, 10283) X(153, 512, 8, 1, false, 14379) \
    X(154, 512, 8, 0, false, 2091)

// loader / binner split (papr_sweep_split_kernel): id, loader waves, binners per loader, loads per lane per tile, ring depth
// (measured slower than papr_sweep_kernel in every shape — DESIGN.md section 4b — so only `make MEASURE=1` builds it)
#define PAPR_FOR_EACH_SWEEP_SPLIT_VARIANT(X) X(70, 8, 1, 4, 4) X(71, 4, 3, 8, 6) X(72, 4, 3, 4, 6) X(73, 4, 2, 8, 4) X(74, 5, 2, 8, 4) X(75, 2, 7, 8, 14)

int papr_lab_sweep_variant(int variant)
{
    if ((variant >= 60 && variant <= 69) || (variant >= 90 && variant <= 99 && variant != 93 && variant != 96) ||
        (variant >= 120 && variant <= 129 && variant != 123 && variant != 126) || (variant >= 150 && variant <= 152))
        return variant;  // ablations of <1024, 4> / <256, 8> (measurement only)
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

int papr_lab_sweep_geometry(int variant, int *threads, uint64_t *tile_samples, size_t *stash_lds)
{
    if (variant >= 60 && variant <= 69)
        variant = 4;
    if (variant >= 90 && variant <= 99)
        variant = 0;
    if ((variant >= 120 && variant <= 129) || (variant >= 150 && variant <= 152))
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

#define PAPR_FOR_EACH_ABLATION(X) X(60, 1) X(61, 2) X(62, 3) X(63, 4) X(64, 8) X(65, 16) X(66, 63) X(67, 32) X(68, 7) X(69, 24)
// (the same of <256, 8>: eight loads in flight per lane, two workgroups per CU — the geometry a stripped kernel reads fastest in)
#define PAPR_FOR_EACH_ABLATION2(X) X(90, 1) X(91, 2) X(92, 3) X(94, 8) X(95, 16) X(97, 32) X(98, 7) X(99, 24)
// (and of the default, <512, 8> with the branch-free stash)
#define PAPR_FOR_EACH_ABLATION3(X) X(120, 1) X(121, 2) X(122, 3) X(124, 8) X(125, 16) X(127, 32) X(128, 7) X(129, 24)
// (and of the product form: the same with the next tile's loads in flight)
#define PAPR_FOR_EACH_ABLATION4(X) X(150, 1) X(151, 2) X(152, 3)

void papr_lab_launch_sweep(hipStream_t st, int variant, int blocks, size_t lds_bytes, const void *data, uint64_t ntiles,
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
        launch_maybe_timed((papr_sweep_lab_kernel<B, U, true, PP, 0, L2, SM>), dim3(blocks), dim3(B), lds_bytes, st,    \
                           (const float4 *)data, ntiles, base_index, map, out, (const float2 *)tail, tail_samples,    \
                           table, P, ghist, stash, seg_counts, seg_cap, gave_up, seg_real, Pdev);                           \
        break;
        PAPR_FOR_EACH_SWEEP_SP16_VARIANT(X)
#undef X
#define X(V, A)                                                                                                      \
    case V:                                                                                                           \
        launch_maybe_timed((papr_sweep_lab_kernel<1024, 4, true, 0, A>), dim3(blocks), dim3(1024), lds_bytes, st,         \
                           (const float4 *)data, ntiles, base_index, map, out, (const float2 *)tail, tail_samples,    \
                           table, P, ghist, stash, seg_counts, seg_cap, gave_up, seg_real, Pdev);                                            \
        break;
        PAPR_FOR_EACH_ABLATION(X)
#undef X
#define X(V, A)                                                                                                      \
    case V:                                                                                                           \
        launch_maybe_timed((papr_sweep_lab_kernel<256, 8, true, 0, A>), dim3(blocks), dim3(256), lds_bytes, st,           \
                           (const float4 *)data, ntiles, base_index, map, out, (const float2 *)tail, tail_samples,    \
                           table, P, ghist, stash, seg_counts, seg_cap, gave_up, seg_real, Pdev);                     \
        break;
        PAPR_FOR_EACH_ABLATION2(X)
#undef X
#define X(V, A)                                                                                                      \
    case V:                                                                                                           \
        launch_maybe_timed((papr_sweep_lab_kernel<512, 8, true, 0, A, false, 43>), dim3(blocks), dim3(512), lds_bytes, st, \
                           (const float4 *)data, ntiles, base_index, map, out, (const float2 *)tail, tail_samples,    \
                           table, P, ghist, stash, seg_counts, seg_cap, gave_up, seg_real, Pdev);                     \
        break;
        PAPR_FOR_EACH_ABLATION3(X)
#undef X
#define X(V, A)                                                                                                      \
    case V:                                                                                                           \
        launch_maybe_timed((papr_sweep_lab_kernel<512, 8, true, 1, A, false, 43>), dim3(blocks), dim3(512), lds_bytes, st, \
                           (const float4 *)data, ntiles, base_index, map, out, (const float2 *)tail, tail_samples,    \
                           table, P, ghist, stash, seg_counts, seg_cap, gave_up, seg_real, Pdev);                     \
        break;
        PAPR_FOR_EACH_ABLATION4(X)
#undef X
#define X(V, B, U, PP)                                                                                               \
    case V:                                                                                                           \
        launch_maybe_timed((papr_sweep_lab_kernel<B, U, true, PP, 0, true>), dim3(blocks), dim3(B), lds_bytes, st,        \
                           (const float4 *)data, ntiles, base_index, map, out, (const float2 *)tail, tail_samples,    \
                           table, P, ghist, stash, seg_counts, seg_cap, gave_up, seg_real, Pdev);                                            \
        break;
        PAPR_FOR_EACH_SWEEP_LUT2_VARIANT(X)
#undef X
#define X(V, B, U, PP)                                                                                               \
    case V:                                                                                                           \
        launch_maybe_timed((papr_sweep_lab_kernel<B, U, true, PP>), dim3(blocks), dim3(B), lds_bytes, st,                 \
                           (const float4 *)data, ntiles, base_index, map, out, (const float2 *)tail, tail_samples,    \
                           table, P, ghist, stash, seg_counts, seg_cap, gave_up, seg_real, Pdev);                                            \
        break;
        PAPR_FOR_EACH_SWEEP_VARIANT(X)
#undef X
    }
}

// Geometry variants of the second-generation sweep: id, waves per workgroup, 16-byte loads per lane per segment,
// next-segment prefetch, exact-sum pairs, stash store policy (0 plain, 1 nontemporal, 2 write-through).
#define PAPR_FOR_EACH_SWEEP2_VARIANT(X)                                                                         \
    X(32, 16, 8, 0, false, 2) X(34, 16, 4, 0, false, 2) X(35, 16, 4, 1, false, 2) X(30, 12, 8, 0, false, 2)      \
    X(41, 12, 8, 1, false, 2) X(42, 8, 4, 1, false, 2) X(44, 16, 8, 0, false, 6) X(45, 12, 8, 1, false, 6)       \
    X(48, 12, 8, 0, true, 2) X(49, 12, 8, 0, true, 6) X(50, 12, 8, 0, true, 10) X(51, 12, 8, 0, true, 0)         \
    X(52, 12, 8, 0, true, 14) X(54, 11, 8, 0, true, 10) X(55, 12, 8, 0, true, 18) X(56, 12, 8, 0, true, 26)                \
    X(57, 16, 8, 0, false, 18) X(58, 12, 8, 1, false, 18) X(59, 12, 8, 0, true, 50) X(53, 12, 8, 0, true, 16)        \
    X(46, 12, 8, 0, true, 17) X(47, 12, 8, 0, true, 58) X(43, 14, 8, 0, true, 26) X(33, 13, 8, 0, true, 26)       \
    X(31, 12, 8, 0, true, 218) X(29, 12, 8, 0, true, 90) X(28, 12, 8, 0, true, 154)

int papr_lab_sweep2_geometry(int variant, int *threads, uint64_t *seg_samples, size_t *lds_fixed, int *exact)
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

void papr_lab_launch_sweep2(hipStream_t st, int variant, int blocks, size_t lds_bytes, const papr_sweep2_params &p)
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

void papr_lab_prepare_device(void)
{
    const int want = papr_ccdf_max_dynamic_lds();
#define X(V, B, U, PP)                                                                                               \
    (void)hipFuncSetAttribute((const void *)papr_sweep_lab_kernel<B, U, true, PP>,                                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    PAPR_FOR_EACH_SWEEP_VARIANT(X)
#undef X
#define X(V, A)                                                                                                      \
    (void)hipFuncSetAttribute((const void *)papr_sweep_lab_kernel<1024, 4, true, 0, A>,                                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    PAPR_FOR_EACH_ABLATION(X)
#undef X
#define X(V, A)                                                                                                      \
    (void)hipFuncSetAttribute((const void *)papr_sweep_lab_kernel<256, 8, true, 0, A>,                                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    PAPR_FOR_EACH_ABLATION2(X)
#undef X
#define X(V, A)                                                                                                      \
    (void)hipFuncSetAttribute((const void *)papr_sweep_lab_kernel<512, 8, true, 0, A, false, 43>,                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    PAPR_FOR_EACH_ABLATION3(X)
#undef X
#define X(V, B, U, PP)                                                                                               \
    (void)hipFuncSetAttribute((const void *)papr_sweep_lab_kernel<B, U, true, PP, 0, true>,                              \
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
    (void)hipFuncSetAttribute((const void *)papr_sweep_lab_kernel<B, U, true, PP, 0, L2, SM>,                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, want);
    PAPR_FOR_EACH_SWEEP_SP16_VARIANT(X)
#undef X

}

