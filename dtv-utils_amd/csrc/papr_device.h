// papr_device.h — device helpers shared by papr_kernels.hip and papr_exact.hip.
#ifndef PAPR_DEVICE_H
#define PAPR_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

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

}  // namespace

#endif
