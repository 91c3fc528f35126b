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

}  // namespace

#endif
