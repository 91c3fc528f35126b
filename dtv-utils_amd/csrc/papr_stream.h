// papr_stream.h — device building blocks of the streaming kernels (papr_kernels.hip, papr_sweep.hip):
// which tiles a workgroup walks, the per-lane pass-1 state and its fold over one tile of loads,
// and the workgroup-wide merge.  Internal; included by device code only.
#ifndef PAPR_STREAM_H
#define PAPR_STREAM_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "papr_kernels.h"
#include "papr_device.h"

namespace {

// Which tiles a workgroup walks, and in what order.  Every mapping visits a
// lane's samples in increasing index order, which is what makes the per-lane
// strict-compare trackers keep the FIRST occurrence.
struct TileWalk {
    uint64_t first;   // first tile
    uint64_t stride;  // tiles between consecutive iterations
    uint32_t count;   // iterations
};

__device__ __forceinline__ TileWalk tile_walk(uint32_t b, uint32_t nblocks, uint64_t ntiles, int map)
{
    TileWalk w;
    if (map == PAPR_MAP_BLOCK_SPAN) {
        // one contiguous span of tiles per workgroup
        uint64_t per = (ntiles + nblocks - 1) / nblocks;
        w.first = (uint64_t)b * per;
        w.stride = 1;
        uint64_t left = w.first < ntiles ? ntiles - w.first : 0;
        w.count = (uint32_t)(left < per ? left : per);
    } else if (map == PAPR_MAP_XCD_SPAN && (nblocks % 8u) == 0) {
        // workgroup b is observed to run on XCD b % 8: give each XCD one
        // contiguous eighth of the shard and stride its workgroups inside it
        uint32_t xcd = b & 7u, slot = b >> 3, per_x = nblocks >> 3;
        uint64_t span = (ntiles + 7) / 8;
        uint64_t x0 = (uint64_t)xcd * span;
        uint64_t xn = x0 < ntiles ? ntiles - x0 : 0;
        if (xn > span) xn = span;
        w.first = x0 + slot;
        w.stride = per_x;
        w.count = xn > slot ? (uint32_t)((xn - slot + per_x - 1) / per_x) : 0;
    } else {
        // grid-stride over tiles: concurrently running workgroups read
        // neighbouring tiles
        w.first = b;
        w.stride = nblocks;
        w.count = ntiles > b ? (uint32_t)((ntiles - b + nblocks - 1) / nblocks) : 0;
    }
    return w;
}

// ---- (value, index) trackers ------------------------------------------------

template <bool IS_MIN>
__device__ __forceinline__ void track(float x, uint32_t code, float &best, uint32_t &best_code)
{
    const bool win = IS_MIN ? (x < best) : (x > best);  // strict; NaN never wins
    best = win ? x : best;
    best_code = win ? code : best_code;
}

template <bool IS_MIN>
__device__ __forceinline__ bool beats(float av, uint64_t ai, float bv, uint64_t bi)
{
    // "more extreme value, else smaller index" — order-independent merge rule
    const bool more = IS_MIN ? (av < bv) : (av > bv);
    return more || (av == bv && ai < bi);
}

template <bool IS_MIN>
__device__ __forceinline__ void wave_reduce_pair(float &v, uint64_t &i)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const float ov = __shfl_down(v, off, kWave);
        const uint64_t oi = __shfl_down((unsigned long long)i, off, kWave);
        if (beats<IS_MIN>(ov, oi, v, i)) {
            v = ov;
            i = oi;
        }
    }
}

__device__ __forceinline__ double wave_reduce_sum(double s)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1)
        s += __shfl_down(s, off, kWave);
    return s;
}

struct LaneStats {
    double sum;
    float val[5];      // peak, re_pos, re_neg, im_pos, im_neg
    uint64_t idx[5];
};

__device__ __forceinline__ void lane_stats_sample(LaneStats &s, float re, float im, uint64_t index)
{
    const float pw = power_of(re, im);
    s.sum += (double)pw;
    if (pw > s.val[0]) { s.val[0] = pw; s.idx[0] = index; }
    if (re > s.val[1]) { s.val[1] = re; s.idx[1] = index; }
    if (re < s.val[2]) { s.val[2] = re; s.idx[2] = index; }
    if (im > s.val[3]) { s.val[3] = im; s.idx[3] = index; }
    if (im < s.val[4]) { s.val[4] = im; s.idx[4] = index; }
}

// Workgroup-wide merge of LaneStats; the result is valid in thread 0.
template <int kWaves>
__device__ __forceinline__ void block_reduce_stats(LaneStats &s)
{
    __shared__ double sh_sum[kWaves];
    __shared__ float sh_val[kWaves][5];
    __shared__ uint64_t sh_idx[kWaves][5];

    s.sum = wave_reduce_sum(s.sum);
    wave_reduce_pair<false>(s.val[0], s.idx[0]);
    wave_reduce_pair<false>(s.val[1], s.idx[1]);
    wave_reduce_pair<true>(s.val[2], s.idx[2]);
    wave_reduce_pair<false>(s.val[3], s.idx[3]);
    wave_reduce_pair<true>(s.val[4], s.idx[4]);

    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) {
        sh_sum[wave] = s.sum;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            sh_val[wave][k] = s.val[k];
            sh_idx[wave][k] = s.idx[k];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWaves; w++) {  // fixed order => deterministic sum
            s.sum += sh_sum[w];
            if (beats<false>(sh_val[w][0], sh_idx[w][0], s.val[0], s.idx[0])) { s.val[0] = sh_val[w][0]; s.idx[0] = sh_idx[w][0]; }
            if (beats<false>(sh_val[w][1], sh_idx[w][1], s.val[1], s.idx[1])) { s.val[1] = sh_val[w][1]; s.idx[1] = sh_idx[w][1]; }
            if (beats<true>(sh_val[w][2], sh_idx[w][2], s.val[2], s.idx[2])) { s.val[2] = sh_val[w][2]; s.idx[2] = sh_idx[w][2]; }
            if (beats<false>(sh_val[w][3], sh_idx[w][3], s.val[3], s.idx[3])) { s.val[3] = sh_val[w][3]; s.idx[3] = sh_idx[w][3]; }
            if (beats<true>(sh_val[w][4], sh_idx[w][4], s.val[4], s.idx[4])) { s.val[4] = sh_val[w][4]; s.idx[4] = sh_idx[w][4]; }
        }
    }
}


// ---- pass-1 per-lane state and its fold over one tile (U 16-byte loads per lane) ----


struct StatsRegs {
    double sum;
    float v_pk, v_rp, v_rn, v_ip, v_in;
    uint32_t c_pk, c_rp, c_rn, c_ip, c_in;
};

// TSUM: also return this iteration's per-lane sum (the exact-sum path needs a sum per tile)
template <int U, bool TSUM>
__device__ __forceinline__ double stats_fold(StatsRegs &r, const float4 (&x)[U], uint32_t code)
{
    double it_sum = 0.0;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const float p0 = power_of(x[u].x, x[u].y);
        const float p1 = power_of(x[u].z, x[u].w);
        if constexpr (TSUM) {
            it_sum += (double)p0;
            it_sum += (double)p1;
        } else {
            r.sum += (double)p0;
            r.sum += (double)p1;
        }
        const uint32_t c0 = code + 2 * u, c1 = c0 + 1;
        track<false>(p0, c0, r.v_pk, r.c_pk);
        track<false>(p1, c1, r.v_pk, r.c_pk);
        track<false>(x[u].x, c0, r.v_rp, r.c_rp);
        track<false>(x[u].z, c1, r.v_rp, r.c_rp);
        track<true>(x[u].x, c0, r.v_rn, r.c_rn);
        track<true>(x[u].z, c1, r.v_rn, r.c_rn);
        track<false>(x[u].y, c0, r.v_ip, r.c_ip);
        track<false>(x[u].w, c1, r.v_ip, r.c_ip);
        track<true>(x[u].y, c0, r.v_in, r.c_in);
        track<true>(x[u].w, c1, r.v_in, r.c_in);
    }
    if constexpr (TSUM)
        r.sum += it_sum;
    return it_sum;
}

template <int BLOCK, int U, bool NT>
__device__ __forceinline__ void load_tile(float4 (&x)[U], const float4 *p)
{
#pragma unroll
    for (int u = 0; u < U; u++)
        x[u] = load16<NT>(p + u * BLOCK);
}

// Expand the trackers' 32-bit codes to global sample indices, merge the workgroup and store its
// record.  A tracker that never fired keeps value 0 and reports index 0 like the reference's
// initialisers.
template <int BLOCK, int U>
__device__ __forceinline__ void stats_finish(const StatsRegs &r, const TileWalk &w, uint64_t base_index,
                                             papr_partial *__restrict__ out)
{
    constexpr uint64_t TILE_F4 = (uint64_t)BLOCK * U;
    const uint32_t t = threadIdx.x;
    LaneStats s;
    s.sum = r.sum;
    const float vals[5] = {r.v_pk, r.v_rp, r.v_rn, r.v_ip, r.v_in};
    const uint32_t codes[5] = {r.c_pk, r.c_rp, r.c_rn, r.c_ip, r.c_in};
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint32_t c = codes[k];
        const uint64_t tile = w.first + (uint64_t)(c / (2 * U)) * w.stride;
        const uint32_t slot = (c % (2 * U)) >> 1, half = c & 1u;
        const uint64_t idx = base_index + 2 * (tile * TILE_F4 + (uint64_t)slot * BLOCK + t) + half;
        s.val[k] = vals[k];
        s.idx[k] = vals[k] != 0.f ? idx : 0;
    }
    block_reduce_stats<BLOCK / kWave>(s);
    if (t == 0) {
        papr_partial q;
        q.sum = s.sum;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            q.idx[k] = s.idx[k];
            q.val[k] = s.val[k];
        }
        q.pad = 0;
        out[blockIdx.x] = q;
    }
}

}  // namespace

#endif
