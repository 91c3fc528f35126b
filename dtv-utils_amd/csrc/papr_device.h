// papr_device.h — device helpers shared by papr_kernels.hip and papr_exact.hip.
#ifndef PAPR_DEVICE_H
#define PAPR_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "papr_kernels.h"

namespace {

constexpr int kWave = 64;

__device__ __forceinline__ float power_of(float re, float im)
{
    return __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one global_load_dwordx4 per lane (optionally with the nontemporal hint: the
// shard is streamed exactly once per pass, nothing is worth keeping in L2/MALL)
template <bool NT>
__device__ __forceinline__ float4 load16(const float4 *p)
{
    const f32x4 *q = reinterpret_cast<const f32x4 *>(p);
    const f32x4 v = NT ? __builtin_nontemporal_load(q) : *q;
    return make_float4(v.x, v.y, v.z, v.w);
}

// The same from a wave-uniform base address (scalar registers) + a per-lane element index: the compiler addresses
// with a scalar base and one 32-bit offset register instead of a 64-bit pointer per lane.  (The base arrives as an
// integer — v_readfirstlane'd — so the global address space has to be said: a plain pointer cast gives flat_load.)
__device__ __forceinline__ float4 load16_nt_at(unsigned long long uniform_base, uint32_t element)
{
    typedef __attribute__((address_space(1))) const f32x4 gf32x4;
    gf32x4 *q = reinterpret_cast<gf32x4 *>(uniform_base) + element;
    const f32x4 v = __builtin_nontemporal_load(q);
    return make_float4(v.x, v.y, v.z, v.w);
}

// Wave64 sum of a double with DPP row shifts/broadcasts (VALU latency only, no LDS
// round trips): inclusive scan inside each 16-lane row, then row_bcast:15 / :31 carry
// the row totals forward.  The total ends up in LANE 63; fixed order => deterministic.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_add_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
    return v + __hiloint2double(hi, lo);  // lanes without a valid source add +0.0
}

__device__ __forceinline__ double wave_sum_to_lane63(double v)
{
    v = dpp_add_f64<0x111, 0xf, 0xf>(v);  // row_shr:1
    v = dpp_add_f64<0x112, 0xf, 0xf>(v);  // row_shr:2
    v = dpp_add_f64<0x114, 0xf, 0xf>(v);  // row_shr:4
    v = dpp_add_f64<0x118, 0xf, 0xf>(v);  // row_shr:8  -> lane 15 of every row holds the row sum
    v = dpp_add_f64<0x142, 0xa, 0xf>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add_f64<0x143, 0xc, 0xf>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
    return v;
}

// inclusive prefix sum over the wave's lanes (the same six DPP steps), of small counts
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp_add_u32(uint32_t v)
{
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}

__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v)
{
    v = dpp_add_u32<0x111, 0xf, 0xf>(v);  // row_shr:1
    v = dpp_add_u32<0x112, 0xf, 0xf>(v);  // row_shr:2
    v = dpp_add_u32<0x114, 0xf, 0xf>(v);  // row_shr:4
    v = dpp_add_u32<0x118, 0xf, 0xf>(v);  // row_shr:8
    v = dpp_add_u32<0x142, 0xa, 0xf>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add_u32<0x143, 0xc, 0xf>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// ---- pass-2 binning helpers (see the comment block above papr_ccdf_kernel) -----------

template <int BLOCK>
__device__ __forceinline__ void hist_flush(const uint32_t *hist, uint32_t nbins, uint32_t copies,
                                           unsigned long long *__restrict__ ghist)
{
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbins; b += BLOCK) {
        unsigned long long s = 0;
        for (uint32_t c = 0; c < copies; c++)
            s += hist[c * nbins + b];
        if (s)
            atomicAdd(&ghist[b], s);
    }
}

__device__ __forceinline__ uint32_t lut_bin(uint32_t bits, const uint2 *lut, const papr_ccdf_params &P)
{
    const uint32_t rel = (bits >> P.shift) - P.cell_lo;  // wraps to huge below the table
    uint32_t k;
    if (rel < P.ncells) {
        const uint2 e = lut[rel];
        k = e.x + (bits >= e.y ? 1u : 0u);
    } else {
        // above the table but not NaN => above every level; below or NaN => 0
        k = (bits - P.above_lo) < P.above_count ? P.nkeys : 0u;
    }
    return k;
}

__device__ __forceinline__ uint32_t search_bin(uint32_t bits, const uint32_t *keys, const papr_ccdf_params &P)
{
    if (bits > 0x7F800000u)  // NaN (either sign): above nothing
        return 0;
    uint32_t lo = 0;
    for (uint32_t step = P.search_step; step; step >>= 1) {
        const uint32_t mid = lo + step;
        if (mid <= P.nkeys && keys[mid - 1] <= bits)
            lo = mid;
    }
    return lo;
}


// ---- shared by papr_exact.hip and papr_sweep.hip ---------------------------------------------------------------------
// exclusive scan of one value per thread over the workgroup, in thread order (wave shuffles, then the wave totals):
// what a single thread walking an LDS array did in 256 / 1024 dependent steps (17 us / 37 us per kernel,
// profiles/r02_step_timeline.txt).  `wave_tot`: BLOCK / 64 entries of LDS; `total` (optional): the sum of all values.
template <typename T, int BLOCK>
__device__ __forceinline__ T block_exclusive_scan(T v, T *wave_tot, T *total)
{
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    T inc = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const T o = __shfl_up(inc, d);
        if (lane >= d)
            inc += o;
    }
    T ex = __shfl_up(inc, 1);
    if (lane == 0)
        ex = T(0);
    __syncthreads();  // (wave_tot may still be read from a previous call)
    if (lane == kWave - 1)
        wave_tot[w] = inc;
    __syncthreads();
    T base = T(0), all = T(0);
#pragma unroll
    for (int k = 0; k < BLOCK / kWave; k++) {
        const T x = wave_tot[k];
        if (k < w)
            base += x;
        all += x;
    }
    if (total)
        *total = all;
    return base + ex;
}


// The exact one-read sweep's binade speculation, first half: group g of the estimate = tiles [g * ratio, (g + 1) * ratio);
// group_sums[4 g .. 4 g + 3] (one per wave of the estimate kernel) add up to the sum of ONE tile's worth of its rows, so
// `scale` (= ratio) times that estimates the group's sum.  Exclusive scan over the groups by one workgroup of 1024
// (papr_exact_spec_scan_kernel — or, without peers, the second workgroup of papr_guess_bands_kernel, beside the guess).
__device__ __forceinline__ void papr_exact_spec_scan_body(const double *__restrict__ group_sums, uint64_t ngroups, double scale,
                                                          double before, double *__restrict__ group_prefix, double *sh /* 16 */)
{
    // Every wave takes one contiguous sixteenth of the groups, 64 at a time: lane l loads group (base + l) — coalesced, eight
    // rounds in flight — and the 64 values are scanned inside the wave; the carry from round to round is the wave's own,
    // and ONE barrier at the end gives every wave what lies in front of its sixteenth.  (A thread walking its own range of
    // groups read 32 bytes at a stride of 320: uncoalesced dependent trips to the L2, 24 us for 10 240 groups.)
    // (an estimate: the order of these additions decides nothing but which tiles get redone)
    constexpr int W = 1024 / kWave, CH = 8;
    const uint32_t lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    const uint64_t per_wave = ((ngroups + W - 1) / W + kWave - 1) / kWave * kWave;
    const uint64_t a = w * per_wave, e = a + per_wave < ngroups ? a + per_wave : ngroups;
    auto gsum = [&](uint64_t k) {
        const double2 *p = reinterpret_cast<const double2 *>(group_sums + 4 * k);  // (32-byte records: 16-byte aligned)
        const double2 lo = p[0], hi = p[1];
        return (((lo.x + lo.y) + hi.x) + hi.y) * scale;
    };
    auto wave_inclusive = [&](double v) {
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const double o = __shfl_up(v, d, kWave);
            if ((int)lane >= d)
                v += o;
        }
        return v;
    };
    // pass 1: the wave's total
    double s = 0.0;
    for (uint64_t k0 = a; k0 < e; k0 += (uint64_t)CH * kWave) {
        double g[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const uint64_t k = k0 + (uint64_t)j * kWave + lane;
            g[j] = k < e ? gsum(k) : 0.0;
        }
#pragma unroll
        for (int j = 0; j < CH; j++)
            s += g[j];
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1)
        s += __shfl_xor(s, off, kWave);
    __syncthreads();  // (sh may still be read from a previous use)
    if (lane == 0)
        sh[w] = s;
    __syncthreads();
    double run = before;
    for (int k = 0; k < W; k++)
        run += (uint32_t)k < w ? sh[k] : 0.0;
    // pass 2: the prefixes (the values come out of the L2 / L1 this time)
    for (uint64_t k0 = a; k0 < e; k0 += (uint64_t)CH * kWave) {
        double g[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const uint64_t k = k0 + (uint64_t)j * kWave + lane;
            g[j] = k < e ? gsum(k) : 0.0;
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const uint64_t k = k0 + (uint64_t)j * kWave + lane;
            const double inc = wave_inclusive(g[j]);
            if (k < e)
                group_prefix[k] = run + (inc - g[j]);
            run += __shfl(inc, kWave - 1, kWave);
        }
    }
}

}  // namespace

#endif
