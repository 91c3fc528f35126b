// papr_sweep_dev.h — device building blocks shared by the one-sweep kernels (papr_sweep.hip; with MEASURE=1 also
// measure/papr_sweep_lab.hip): table geometry and lookups, the per-wave stash, per-tile trackers and the workgroup
// record, the segment's exact-sum pair.  Internal; included by device code only.
#ifndef PAPR_SWEEP_DEV_H
#define PAPR_SWEEP_DEV_H

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "papr_kernels.h"
#include "papr_device.h"
#include "papr_stream.h"
#include "papr_skew_walk.h"

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

// a wave gives the sweep up for its workgroup: the LUT in LDS becomes "bin 0 everywhere" (no counter, no stash: the rest of
// the launch runs at pass-1 speed) and the segment's length is pushed past its capacity, which the host reads as `stash
// full` and answers with the plain pass 2.  Pass-1 results do not depend on the LUT.
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


__device__ __forceinline__ void store16_wt(float *p, f32x4s v) { store16<2>(p, v); }

// Per-wave append buffer in LDS + its spill to the workgroup's segment of the HBM stash.  The slice is this wave's alone,
// so the address of its next free slot lives in a scalar register and slots are handed out by ballot + mbcnt: no
// returning LDS atomic, no wait, no branch and no exec-masked region per sample — every lane writes, its power to its
// slot or to a trash word of its own at the end of the slice (with two waves per SIMD beside it a wave cannot hide a
// v_cmp -> s_and_saveexec round per sample).  Spills go out as 16-byte write-through stores (a partial quad padded with
// quiet NaNs, which the recount ignores).
// A sweep whose bands catch most of the stream (a constant-envelope capture: every power sits next to the mean)
// cannot be answered from the stash, and must not cost more than the pass it replaces: each spill compares what the
// workgroup stashed in this launch with what it folded, and once more than half of it was in band — or the segment is
// full — the wave GIVES UP for the whole workgroup (sweep_give_up).
constexpr int kSpillEpochShift = 10;  // 1024 ticks of the 100 MHz real-time clock
constexpr uint32_t kSpillUnarmed = 0xFFFFFFFFu;

struct WaveStash {
    float *buf;                         // this wave's slice of LDS
    float *__restrict__ seg;            // this workgroup's stash segment
    unsigned long long *seg_fill;       // LDS: floats reserved in the segment so far (may run past seg_cap)
    unsigned long long *seg_real;       // LDS: powers stashed, without padding
    uint64_t seg_cap;
    uint32_t *tab;                      // LDS: the LUT (sweep_give_up)
    uint32_t table_words;
    unsigned long long seg_start;       // the segment's length when this launch began
    unsigned long long *gave_up;        // device counter of give-ups
    uint32_t trash;                     // this lane's own word at the end of the slice, where what is not in band goes
    uint32_t sbase, sbytes;             // LDS byte address of the slice, and of its next free slot (wave-uniform)
    uint32_t epoch, epochs, n_ref;      // spill_in_step: the clock epoch last seen; whole epochs since the fill was n_ref
    uint32_t rate;                      // ... powers per epoch, times 9/8
    bool in_step;

    __device__ __forceinline__ void init(float *slice, uint32_t slice_floats, float *segment, unsigned long long *fill,
                                         unsigned long long *real, uint64_t cap, uint32_t *table, uint32_t words,
                                         unsigned long long *gave_up_counter)
    {
        buf = slice;
        seg = segment;
        seg_fill = fill;
        seg_real = real;
        seg_cap = cap;
        tab = table;
        table_words = words;
        seg_start = *fill;
        gave_up = gave_up_counter;
        trash = slice_floats - kWave + (threadIdx.x & (kWave - 1));
        sbase = sbytes = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u32 *)slice);
        epoch = (uint32_t)(__builtin_amdgcn_s_memrealtime() >> kSpillEpochShift);
        epochs = kSpillUnarmed;
        n_ref = 0;
        rate = 0;
        in_step = false;
    }
    __device__ __forceinline__ void put(float pw, bool take) { put_masked(pw, __ballot(take)); }
    // the same with the takers' ballot handed in (a tile's ballots are taken first, to know what the tile needs: reserve)
    __device__ __forceinline__ void put_masked(float pw, unsigned long long m)
    {
        // (the select is written out: from `take ? at : trash` hipcc makes an exec-masked region per sample)
        // Addresses in bytes: slot = rank among the takers * 4 + the next free slot's address, the latter wave-uniform
        // (one scalar operand of the v_lshl_add) — no copy of the fill count into a vector register per sample.
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        uint32_t a_slot;
        asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a_slot) : "v"(rank), "s"(__builtin_amdgcn_readfirstlane(sbytes)));
        const uint32_t a_trash = (uint32_t)(uintptr_t)(lds_u32 *)&buf[trash];
        uint32_t a;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a) : "v"(a_trash), "v"(a_slot), "s"(m));
        *(__attribute__((address_space(3))) float *)(uintptr_t)a = pw;
        // (s_lshl2_add_u32 writes SCC: said, so that the compiler never schedules it between a compare and its consumer)
        asm("s_lshl2_add_u32 %0, %1, %2" : "=s"(sbytes) : "s"((uint32_t)__popcll(m)), "s"(__builtin_amdgcn_readfirstlane(sbytes)) : "scc");
    }
    // A whole tile's in-band powers at once — N per lane, flag[u] = 1 where pw[u] is in band, cnt = the lane's number of
    // them.  Slots are handed out per LANE: one prefix sum over the wave's lanes per tile (six DPP additions) gives every
    // lane a run of cnt slots, and a sample then costs two vector instructions and its write — the slot's address as
    // trash + flag * (next - trash), and next += 4 * flag — instead of a ballot, two mbcnt, a shift-add, a select and two
    // scalar instructions per sample (a sixth of the kernel's scalar work): no compare, no scalar register in the chain.
    // The stash is an unordered multiset of powers: lane-major order inside a tile is as good as sample-major.
    // Makes room first (reserve: the tile's actual need), so a tile never straddles a spill.
    template <int N>
    __device__ __forceinline__ void put_tile(const float (&pw)[N], const uint32_t (&flag)[N], uint32_t cnt, uint32_t slice_floats,
                                             uint32_t folded)
    {
        const uint32_t incl = wave_inclusive_scan_u32(cnt);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
        reserve(total, slice_floats, folded);
        const int32_t a_trash = (int32_t)(uintptr_t)(lds_u32 *)&buf[trash];
        // byte address of this lane's next free slot, relative to its trash word
        int32_t delta = (int32_t)(__builtin_amdgcn_readfirstlane(sbytes) + 4u * (incl - cnt)) - a_trash;
#pragma unroll
        for (int u = 0; u < N; u++) {
            int32_t a;
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(a) : "v"(flag[u]), "v"(delta), "v"(a_trash));
            *(__attribute__((address_space(3))) float *)(uintptr_t)(uint32_t)a = pw[u];
            asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(delta) : "v"(flag[u]), "v"(delta));
        }
        sbytes = __builtin_amdgcn_readfirstlane(sbytes) + 4u * total;
    }
    // The per-tile spill check.  When every wave spills as its own slice fills up the memory side sees the stash as a
    // trickle of small writes from 2048 waves at 2048 different moments, and each of them turns a channel's bus around
    // under the read stream: measured on the 0.1 dB table, 250 MB of stash cost as much time as 1.3 GB of reads
    // (spills with the stores left out cost nothing).  So the waves of the whole chip spill IN STEP, at the ticks of a
    // clock they all see — the 100 MHz real-time counter, in epochs of 10.24 us: the writes then arrive as bursts of up
    // to 12 MB between long stretches of pure reads.  At the beginning of epoch number e a wave asks whether its slice
    // lasts until the next epoch whose number has MORE trailing zeros than e — 2^ctz(e) epochs from now — at the rate it
    // has been filling, an eighth to spare, and spills if not: epochs with many trailing zeros are where everybody
    // meets, and a wave only spills on the way there if it must.  All waves see the same stream statistics, so they
    // take the same decisions; the ones that do not still spill on a common tick.  A slice that fills up in between is
    // spilled at once, as before.
    // `now`: __builtin_amdgcn_s_memrealtime() read at the START of the tile (its latency then hides under the fold);
    // `limit`: what may wait (the next tile must fit behind it).
    __device__ __forceinline__ void spill_in_step(uint64_t now, uint32_t limit, uint32_t folded)
    {
        const uint32_t ep = (uint32_t)(now >> kSpillEpochShift);
        if (ep != epoch) {  // (wave-uniform)
            epoch = ep;
            const uint32_t n = (sbytes - sbase) >> 2;
            if (epochs == kSpillUnarmed) {  // the first whole epoch since the start / since the slice ran over begins here
                epochs = 0;
                n_ref = n;
            } else if (++epochs >= 2u || rate == 0u) {  // (one epoch alone is a noisy measure: the last spill's rate stands)
                rate = ((n - n_ref) * 9u) / epochs;     // per epoch, times 9/8 (n < 2^12)
            }
            const uint32_t zeros = ep ? min((uint32_t)__builtin_ctz(ep), 12u) : 12u;
            if (n + ((rate << zeros) >> 3) + kWave > limit) {
                limit = 0;
                in_step = true;
            }
        }
        spill_if_above(limit, folded);
        in_step = false;
    }
    // room for `want` more powers (what the tile at hand will put: the popcounts of its ballots)?  If not, the slice goes
    // out now, out of step.  Reserving the tile's ACTUAL need instead of its worst case (every sample of every lane in
    // band: a third of a 12 KiB slice, most of a 6 KiB one) is what lets a slice collect for 80 us between two ticks.
    __device__ __forceinline__ void reserve(uint32_t want, uint32_t slice_floats, uint32_t folded)
    {
        const uint32_t n = (sbytes - sbase) >> 2;
        if (n + want > slice_floats - kWave)  // (the last kWave words are the lanes' trash words)
            spill_if_above(0, folded);
    }
    // spill if more than `limit` entries are waiting (wave-uniform decision); `folded` = samples this workgroup has
    // folded in this launch, about
    __device__ __forceinline__ void spill_if_above(uint32_t limit, uint32_t folded)
    {
        const uint32_t n = (sbytes - sbase) >> 2;
        if (n <= limit)
            return;
        sbytes = sbase;
        epochs = in_step ? 0u : kSpillUnarmed;  // (a spill in step starts whole epochs; one out of step does not)
        n_ref = 0;
        __builtin_amdgcn_wave_barrier();  // LDS is in-order per wave; this pins the compiler's order too
        const uint32_t lane = threadIdx.x & (kWave - 1);
        const uint32_t nres = (n + 3u) & ~3u;  // floats reserved in the segment
        unsigned long long pos = 0;
        if (lane == 0) {
            pos = atomicAdd(seg_fill, (unsigned long long)nres);  // counts even what no longer fits: the host sees the overflow
            atomicAdd(seg_real, (unsigned long long)n);
        }
        pos = uniform_u64(pos);  // lane 0's value, in scalar registers
        for (uint32_t i = 4 * lane; i < nres; i += 4 * kWave) {  // (the slice and the segment are 16-byte aligned)
            f32x4s v = *reinterpret_cast<const f32x4s *>(buf + i);
            const float pad = __uint_as_float(PAPR_STASH_PAD_BITS);
            v.y = i + 1 < n ? v.y : pad;
            v.z = i + 2 < n ? v.z : pad;
            v.w = i + 3 < n ? v.w : pad;
            if (pos + i + 4 <= seg_cap)
                store16_wt(seg + pos + i, v);
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t got = (uint32_t)(pos - seg_start) + nres;  // (a workgroup folds < 2^32 samples per launch)
        if (pos <= seg_cap && (pos + nres > seg_cap || (got >= kGiveUpMin && got > folded / 2)))
            sweep_give_up(tab, table_words, 0u, seg_fill, seg_cap, gave_up);  // (pos > seg_cap: someone already did)
    }
};


// WHICH tiles a persistent workgroup folds: grid stride — concurrently running workgroups read neighbouring tiles — with a
// SKEW between the XCDs.  The chip's XCDs do not read at one speed: with equal shares the workgroups of XCDs 1, 3, 5, 7 finish
// behind those of 0, 2, 4, 6 (a stripped read kernel: by 7 %; the sweep kernels: by 25-55 us of 1.6 ms — tools/wg_skew_probe.hip,
// tools/wg_finish_probe.py, profiles/r05_xcd_skew.txt), and a launch lasts as long as its slowest workgroup.  So in every
// period of R rounds the last round belongs to the workgroups on the even XCDs alone: the others fold (R - 1) / R of what they
// fold.  Workgroup b runs on XCD (b + first) mod 8, `first` being the queue's (6 alone, 5 beside RCCL's queues): the host finds it
// out (papr_sweep_rt.cpp: xcd_even_slow) and hands the kernels the PARITY to skew — one value for the whole launch, so that which
// tiles are folded never depends on what a workgroup reads from its own hardware registers.  Still a fixed function of (workgroup, iteration): nothing is drawn at run time,
// results do not depend on timing, a lane meets its tiles in increasing order.  (Handing the tiles out from a counter in
// device memory evens the finish times out completely and costs per tile what that is worth: measured, not kept.)
// the XCD this wave runs on (gfx940+: HW_REG_XCC_ID, bits 3:0)
__device__ __forceinline__ uint32_t papr_xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu; }

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
        q.pad = ((uint32_t)__builtin_amdgcn_s_memrealtime() & 0x0FFFFFFFu) | (papr_xcc_id() << 28);  // when this workgroup was done (100 MHz ticks) and on which XCD (papr_hip_get_wg_finish)
        out[blockIdx.x] = q;
    }
}


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


__device__ __forceinline__ double pow2_f64(int e) { return __longlong_as_double((long long)(e + 1023) << 52); }
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
        q.pad = ((uint32_t)__builtin_amdgcn_s_memrealtime() & 0x0FFFFFFFu) | (papr_xcc_id() << 28);  // when this workgroup was done (100 MHz ticks) and on which XCD (papr_hip_get_wg_finish)
        out[blockIdx.x] = q;
    }
}

// The pair (D0, D1) of a segment from its 64 lanes' runs, without an ordered tree (see papr_sweep3_kernel): x0 / x1 = the
// lane's sums from the even / odd canonical entry state, d0 / d1 = their increments; result valid in lane 63.
__device__ __forceinline__ double2 segment_pair(double x0, double x1, double d0, double d1, double ulp)
{
    const unsigned long long A = __ballot((__double2loint(x0) & 1) != 0), B = __ballot((__double2loint(x1) & 1) != 0);
    const unsigned long long up = __ballot(d1 > d0), dn = __ballot(d1 < d0);
    const unsigned long long C = ~(A ^ B), N = A & ~B;  // lanes whose map is constant / swaps the parity
    unsigned long long px = N;                // prefix XOR (inclusive), then exclusive
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
    return make_double2(S + (double)k0 * ulp, S + (double)k1 * ulp);
}

}  // namespace

// hipLaunchKernelGGL, or — when papr_time_next_launch armed a timer for this thread — the same dispatch with the timer's
// events bound to it (papr_sweep.hip keeps the timer)
bool papr_take_launch_timer(papr_launch_timer *out);
template <typename... Args, typename F = void (*)(Args...)>
static inline void launch_maybe_timed(F kernel, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, Args... args)
{
    papr_launch_timer tm;
    if (papr_take_launch_timer(&tm))
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds_bytes, st, tm.start, tm.stop, 0, args...);
    else
        hipLaunchKernelGGL(kernel, grid, block, lds_bytes, st, args...);
}

#endif
