// exact_form_probe — what bounds the FORM of the exact-sum sweep (papr_sweep3_kernel, papr_sweep.hip)?
// The product kernel reads the shard once through a per-wave LDS transposition (so that a lane owns 16 CONSECUTIVE samples:
// the order papr.c:104 adds them in), runs the reference's additions from the two canonical entry states of the running
// sum's binade as two dependent fp64 chains, and composes the 64 lanes' rounding-function pairs with ballots and scalar
// arithmetic — beside everything the tree-sum kernel does (table lookups, histogram, stash, trackers).  This probe is that
// kernel STRIPPED to the parts the exact sum adds, one at a time, in the product's geometry (one persistent workgroup of
// 8 waves per CU, wave-private 1024-sample segments, gfx950's 16-byte LDS-direct loads with the swizzle in the source
// address, the next segment's loads issued as soon as the buffer has been read out):
//   form 0  the segment through LDS and back into registers, the powers summed in f32           (the transposition alone)
//   form 1  + the two fp64 chains (x0 += v; x1 += v from 2^E and 2^E + ulp), d0 summed             (the chains)
//   form 2  + the pair composition (two ballots on the sums' parities, two on the increments, the scalar prefix XOR,
//             popcounts, one wave sum) and the 16-byte store per segment                            (the full exact-sum part)
//   form 3  form 2 with the composition's prefix XOR done with DPP row shifts on the lanes' bits instead of 64-bit scalar
//             chains ("move the pair composition off SALU")
//   form 5  form 2 with a segment's composition moved to the top of the NEXT segment's fold (software-pipelined: the ballots'
//             scalar chain and the wave sum then have the next fold's independent work around them)
//   forms 6, 7  the composition without its store; the store without the composition
//   form 4  no LDS at all: plain 16-byte nontemporal loads, grid stride, powers summed (the read ceiling of the geometry)
// Each form's time over the same 10 GiB bounds what the product kernel can reach with that much of the work in it.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/exact_form_probe.hip -o bin/exact_form_probe && bin/exact_form_probe [GiB]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kThreads = 512, kWave = 64, kWaves = kThreads / kWave, U = 8;
constexpr uint64_t SEG_F4 = 64ull * U;  // float4s per 1024-sample segment (8 KiB)

__device__ __forceinline__ double pow2_f64(int e) { return __longlong_as_double((long long)(e + 1023) << 52); }
__device__ __forceinline__ int xpose_slot(int run, int w) { return run * 8 + (w ^ ((run >> 1) & 7)); }
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// the product's wave sum (papr_device.h): six DPP steps, VALU latency only
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_add_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_to_lane63(double v)
{
    v = dpp_add_f64<0x111, 0xf, 0xf>(v);
    v = dpp_add_f64<0x112, 0xf, 0xf>(v);
    v = dpp_add_f64<0x114, 0xf, 0xf>(v);
    v = dpp_add_f64<0x118, 0xf, 0xf>(v);
    v = dpp_add_f64<0x142, 0xa, 0xf>(v);
    v = dpp_add_f64<0x143, 0xc, 0xf>(v);
    return v;
}
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_add_i32(int v)
{
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ int wave_isum_to_lane63(int v)
{
    v = dpp_add_i32<0x111, 0xf, 0xf>(v);
    v = dpp_add_i32<0x112, 0xf, 0xf>(v);
    v = dpp_add_i32<0x114, 0xf, 0xf>(v);
    v = dpp_add_i32<0x118, 0xf, 0xf>(v);
    v = dpp_add_i32<0x142, 0xa, 0xf>(v);
    v = dpp_add_i32<0x143, 0xc, 0xf>(v);
    return v;
}

// the product's composition (papr_sweep_dev.h: segment_pair), copied in structure so that the probe prices the same work
__device__ __forceinline__ double2 pair_scalar(double x0, double x1, double d0, double d1, double ulp)
{
    const unsigned long long A = __ballot((__double2loint(x0) & 1) != 0), B = __ballot((__double2loint(x1) & 1) != 0);
    const unsigned long long up = __ballot(d1 > d0), dn = __ballot(d1 < d0);
    const unsigned long long C = ~(A ^ B), N = A & ~B;
    unsigned long long px = N;
    px ^= px << 1;
    px ^= px << 2;
    px ^= px << 4;
    px ^= px << 8;
    px ^= px << 16;
    px ^= px << 32;
    px <<= 1;
    const unsigned long long Z = ~C, Y = (A ^ px) & C;
    const unsigned long long fwd = (Z + (Y << 1)) ^ Z;
    const unsigned long long has = (Z + (C << 1)) ^ Z;
    const unsigned long long odd0 = px ^ fwd, odd1 = odd0 ^ ~has;
    const int k0 = __popcll(up & odd0) - __popcll(dn & odd0), k1 = __popcll(up & odd1) - __popcll(dn & odd1);
    const double S = wave_sum_to_lane63(d0);
    return make_double2(S + (double)k0 * ulp, S + (double)k1 * ulp);
}

// The same result with the parity walk done by the LANES: every lane's map of the entry parity is one of four functions
// (identity, swap, constant 0, constant 1), coded in two bits (image of 0, image of 1); maps compose associatively, so an
// inclusive prefix composition over the lanes is six DPP / shuffle steps on a 2-bit value — no ballot, no 64-bit scalar chain.
// The increments' corrections are then summed with the wave sum that is needed anyway (as small integers beside d0).
__device__ __forceinline__ double2 pair_lanes(double x0, double x1, double d0, double d1, double ulp)
{
    const uint32_t a = (uint32_t)__double2loint(x0) & 1u, b = (uint32_t)__double2loint(x1) & 1u;
    uint32_t f = a | (b << 1);  // this lane's map: bit p = the parity the sum LEAVES with when it entered with parity p
    const int lane = threadIdx.x & 63;
    // inclusive prefix composition: g = f_lane o ... o f_0 (apply the lower lanes first)
    uint32_t g = f;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)g, off, kWave);  // composition of the `off` lanes below this prefix
        // h = g o lo: h(p) = g(lo(p))
        const uint32_t h = ((g >> (lo & 1u)) & 1u) | (((g >> ((lo >> 1) & 1u)) & 1u) << 1);
        g = lane >= off ? h : g;
    }
    // the parity with which the sum ENTERS this lane = the exclusive prefix applied to the segment's entry parity p
    uint32_t ex = (uint32_t)__shfl_up((int)g, 1, kWave);
    ex = lane == 0 ? 2u : ex;  // (identity: image of 0 is 0, image of 1 is 1)
    const int delta = d1 > d0 ? 1 : (d1 < d0 ? -1 : 0);
    const int c0 = (ex & 1u) ? delta : 0, c1 = (ex & 2u) ? delta : 0;
    // one wave sum carries the three numbers: d0 (a multiple of the ulp: exact in any order) and the two small counts
    const double S = wave_sum_to_lane63(d0);
    const int k0 = wave_isum_to_lane63(c0), k1 = wave_isum_to_lane63(c1);
    return make_double2(S + (double)k0 * ulp, S + (double)k1 * ulp);
}

template <int FORM>
__global__ __launch_bounds__(kThreads) void form_kernel(const float4 *__restrict__ data, uint64_t nsegs, int E, double2 *__restrict__ seg_D,
                                                         double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *xpose = reinterpret_cast<float4 *>(smem);
    const uint32_t t = threadIdx.x, lane = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t / kWave);
    float4 *mine = xpose + wave * (kWave * 8);
    typedef __attribute__((address_space(1))) const void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;
    const uint32_t voff_even = 16u * ((lane & ~7u) | ((lane & 7u) ^ ((lane >> 4) & 7u)));
    const uint32_t voff_odd = 16u * ((lane & ~7u) | ((lane & 7u) ^ (((lane >> 4) + 4u) & 7u)));
    auto load_seg_lds = [&](float4 *dst, uint64_t seg) {
        const char *mid = reinterpret_cast<const char *>(uniform_u64((unsigned long long)(data + seg * SEG_F4 + 4 * kWave)));
        lvoid *lmid = (lvoid *)(dst + 4 * kWave);
#define ROW(r) __builtin_amdgcn_global_load_lds((gvoid *)(mid + (((r) & 1) ? voff_odd : voff_even)), lmid, 16, ((r) - 4) * 1024, 2)
        ROW(0); ROW(1); ROW(2); ROW(3); ROW(4); ROW(5); ROW(6); ROW(7);
#undef ROW
    };
    const uint64_t stride = (uint64_t)gridDim.x * kWaves;
    uint64_t seg = (uint64_t)blockIdx.x * kWaves + wave, prev = ~0ull;
    double sum = 0.0;
    float fsum = 0.f;
    double2 D_prev = make_double2(0.0, 0.0);
    double px0 = 0, px1 = 0, pd0 = 0, pd1 = 0;
    const double m0 = pow2_f64(E), ulp = pow2_f64(E - 52), m1 = m0 + ulp;
    if (seg < nsegs)
        load_seg_lds(mine, seg);
    while (seg < nsegs) {
        const uint64_t nseg = seg + stride;
        float4 y[U];
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < U; j++)
            y[j] = mine[xpose_slot((int)lane, j)];
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        asm volatile("" ::: "memory");
        if (FORM >= 2 && FORM != 5 && FORM != 6 && prev != ~0ull && lane == kWave - 1)
            seg_D[prev] = D_prev;
        asm volatile("" ::: "memory");
        if (nseg < nsegs)
            load_seg_lds(mine, nseg);
        if (FORM == 5 && prev != ~0ull) {
            // the PREVIOUS segment's composition here, with this segment's loads in flight and its whole fold to hide behind
            const double2 D = pair_scalar(px0, px1, pd0, pd1, ulp);
            if (lane == kWave - 1)
                seg_D[prev] = D;
        }
        float pw[2 * U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            typedef float f32x2v __attribute__((ext_vector_type(2)));
            const f32x2v a = {y[u].x, y[u].y}, b = {y[u].z, y[u].w};
            const f32x2v aa = a * a, bb = b * b;
            asm("v_add_f32 %0, %1, %2" : "=v"(pw[2 * u]) : "v"(aa.x), "v"(aa.y));
            asm("v_add_f32 %0, %1, %2" : "=v"(pw[2 * u + 1]) : "v"(bb.x), "v"(bb.y));
        }
        if (FORM == 0) {
#pragma unroll
            for (int u = 0; u < 2 * U; u++)
                fsum += pw[u];
        } else {
            double x0 = m0, x1 = m1;
#pragma unroll
            for (int u = 0; u < 2 * U; u++) {
                const double v = (double)pw[u];
                x0 += v;
                x1 += v;
            }
            const double d0 = x0 - m0, d1 = x1 - m1;
            sum += d0;
            if (FORM == 5) {
                px0 = x0, px1 = x1, pd0 = d0, pd1 = d1;
            } else if (FORM == 2)
                D_prev = pair_scalar(x0, x1, d0, d1, ulp);
            else if (FORM == 6) {
                const double2 D = pair_scalar(x0, x1, d0, d1, ulp);
                sum += lane == kWave - 1 ? D.y * 1e-30 : 0.0;
            } else if (FORM == 7)
                D_prev = make_double2(d0, d1);
            else if (FORM == 3)
                D_prev = pair_lanes(x0, x1, d0, d1, ulp);
            else
                sum += d1 * 1e-30;  // (keeps the second chain alive)
        }
        prev = seg;
        seg = nseg;
    }
    if (FORM == 5 && prev != ~0ull)
        D_prev = pair_scalar(px0, px1, pd0, pd1, ulp);
    if (FORM >= 2 && FORM != 6 && prev != ~0ull && lane == kWave - 1)
        seg_D[prev] = D_prev;
    sum += (double)fsum;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        sum += __shfl_down(sum, off, kWave);
    if (lane == 0)
        atomicAdd(out, sum);
}

// ---- geometry: the same per-sample work at two occupancies --------------------------------------------------------------
// The product kernel in ONE more stripped form, with pass 2's per-sample work put back in (a 28 KiB one-edge-per-cell table in
// LDS: a ds_read_b64 lookup, a compare, an LDS atomic on one of 4 x 64 bins; five integer-max trackers), in two geometries:
//   A  8 waves per CU, a wave owns a 1024-sample segment at a time (8 KiB of LDS per wave, 16 samples per lane): the product's
//   B  16 waves per CU, a wave takes its segment as two halves of 512 samples through a 4 KiB buffer (8 samples per lane, the
//      halves' pairs composed in registers): the same 64 KiB of transpose buffers and bytes in flight, twice the waves to hide
//      a wave's LDS round trips and waits behind
template <int WAVES, int UU>
__global__ __launch_bounds__(WAVES * 64) void geom_kernel(const float4 *__restrict__ data, uint64_t nsegs, int E, const uint2 *__restrict__ table,
                                                           uint32_t ncells, uint32_t cell_lo, uint32_t shift, double *__restrict__ out,
                                                           unsigned long long *__restrict__ ghist)
{
    constexpr int HALVES = 8 / UU;  // batches per 1024-sample segment
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint2 *tab = reinterpret_cast<uint2 *>(smem);
    uint32_t *hist = reinterpret_cast<uint32_t *>(tab + ncells);
    float4 *xpose = reinterpret_cast<float4 *>(hist + 4 * 64);
    const uint32_t t = threadIdx.x, lane = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t / kWave);
    for (uint32_t k = t; k < ncells; k += WAVES * 64)
        tab[k] = table[k];
    for (uint32_t k = t; k < 4 * 64; k += WAVES * 64)
        hist[k] = 0;
    __syncthreads();
    float4 *mine = xpose + wave * (kWave * UU);
    uint32_t *my = hist + (wave & 3u) * 64;
    typedef __attribute__((address_space(1))) const void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    // UU = 8: the product's swizzle (two vector offsets); UU = 4: slot(run, w) = run * 4 + (w ^ ((run >> 2) & 3)), ONE vector offset
    const uint32_t voff_even = UU == 8 ? 16u * ((lane & ~7u) | ((lane & 7u) ^ ((lane >> 4) & 7u))) : 16u * ((lane & ~3u) | ((lane & 3u) ^ ((lane >> 4) & 3u)));
    const uint32_t voff_odd = UU == 8 ? 16u * ((lane & ~7u) | ((lane & 7u) ^ (((lane >> 4) + 4u) & 7u))) : voff_even;
    auto load_batch = [&](float4 *dst, uint64_t batch) {  // batch: UU * 64 float4 (UU KiB) of the stream
        const char *mid = reinterpret_cast<const char *>(uniform_u64((unsigned long long)(data + batch * (uint64_t)(UU * kWave) + (UU / 2) * kWave)));
        lvoid *lmid = (lvoid *)(dst + (UU / 2) * kWave);
#define ROW(r) __builtin_amdgcn_global_load_lds((gvoid *)(mid + (((r) & 1) ? voff_odd : voff_even)), lmid, 16, ((r) - UU / 2) * 1024, 2)
        ROW(0); ROW(1); ROW(2); ROW(3);
        if (UU == 8) { ROW(4); ROW(5); ROW(6); ROW(7); }
#undef ROW
    };
    const uint64_t nbatches = nsegs * HALVES, stride = (uint64_t)gridDim.x * WAVES;
    // a wave's batches: segment (it * grid + block) * WAVES + wave, its HALVES batches one after the other
    uint64_t seg = (uint64_t)blockIdx.x * WAVES + wave;
    int half = 0;
    double sum = 0.0;
    int32_t best[5] = {0, 0, 0, INT32_MIN, INT32_MIN};
    const double m0 = pow2_f64(E), ulp = pow2_f64(E - 52), m1 = m0 + ulp;
    const int32_t cfirst = (int32_t)cell_lo, clast = (int32_t)(cell_lo + ncells - 1);
    double2 D_acc = make_double2(0.0, 0.0);
    uint32_t map_acc = 2u;  // identity
    if (seg < nsegs)
        load_batch(mine, seg * HALVES);
    while (seg < nsegs) {
        uint64_t nseg = seg;
        int nhalf = half + 1;
        if (nhalf == HALVES) {
            nhalf = 0;
            nseg = seg + stride;
        }
        float4 y[UU];
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < UU; j++)
            y[j] = UU == 8 ? mine[xpose_slot((int)lane, j)] : mine[(int)lane * 4 + (j ^ (((int)lane >> 2) & 3))];
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        asm volatile("" ::: "memory");
        if (nseg < nsegs)
            load_batch(mine, nseg * HALVES + nhalf);
        float pw[2 * UU];
#pragma unroll
        for (int u = 0; u < UU; u++) {
            typedef float f32x2v __attribute__((ext_vector_type(2)));
            const f32x2v a = {y[u].x, y[u].y}, b = {y[u].z, y[u].w};
            const f32x2v aa = a * a, bb = b * b;
            asm("v_add_f32 %0, %1, %2" : "=v"(pw[2 * u]) : "v"(aa.x), "v"(aa.y));
            asm("v_add_f32 %0, %1, %2" : "=v"(pw[2 * u + 1]) : "v"(bb.x), "v"(bb.y));
            // trackers (integer max on bit patterns, as the product's SegMax)
            const int32_t pa = __float_as_int(pw[2 * u]), pb = __float_as_int(pw[2 * u + 1]);
            best[0] = max(best[0], max(pa, pb));
            best[1] = max(best[1], max(__float_as_int(y[u].x), __float_as_int(y[u].z)));
            best[2] = max(best[2], max(__float_as_int(y[u].y), __float_as_int(y[u].w)));
            best[3] = max(best[3], max(__float_as_int(-y[u].x), __float_as_int(-y[u].z)));
            best[4] = max(best[4], max(__float_as_int(-y[u].y), __float_as_int(-y[u].w)));
        }
        uint2 e_lut[2 * UU];
#pragma unroll
        for (int u = 0; u < 2 * UU; u++) {
            int32_t cell = __float_as_int(pw[u]) >> shift;
            cell = cell < cfirst ? cfirst : (cell > clast ? clast : cell);
            e_lut[u] = tab[cell - cfirst];
        }
        double x0 = m0, x1 = m1;
#pragma unroll
        for (int u = 0; u < 2 * UU; u++) {
            const double v = (double)pw[u];
            x0 += v;
            x1 += v;
        }
#pragma unroll
        for (int u = 0; u < 2 * UU; u++) {
            const uint32_t k = e_lut[u].x + (__float_as_uint(pw[u]) >= e_lut[u].y ? 1u : 0u);
            (void)__hip_atomic_fetch_add((lds_u32 *)&my[k & 63u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const double d0 = x0 - m0, d1 = x1 - m1;
        sum += d0;
        // the batch's pair and its parity map, composed onto the segment's
        {
            const unsigned long long A = __ballot((__double2loint(x0) & 1) != 0), B = __ballot((__double2loint(x1) & 1) != 0);
            const unsigned long long up = __ballot(d1 > d0), dn = __ballot(d1 < d0);
            const unsigned long long C = ~(A ^ B), N = A & ~B;
            unsigned long long px = N;
            px ^= px << 1; px ^= px << 2; px ^= px << 4; px ^= px << 8; px ^= px << 16; px ^= px << 32;
            const unsigned long long incl = px;
            px <<= 1;
            const unsigned long long Z = ~C, Y = (A ^ px) & C;
            const unsigned long long fwd = (Z + (Y << 1)) ^ Z, has = (Z + (C << 1)) ^ Z;
            const unsigned long long odd0 = px ^ fwd, odd1 = odd0 ^ ~has;
            const int k0 = __popcll(up & odd0) - __popcll(dn & odd0), k1 = __popcll(up & odd1) - __popcll(dn & odd1);
            const double S = wave_sum_to_lane63(d0);
            const double2 Db = make_double2(S + (double)k0 * ulp, S + (double)k1 * ulp);
            // exit parity of the batch for entry parity p: lane 63's map applied to the parity it is entered with
            const uint32_t e0 = (uint32_t)(((odd0 >> 63) & 1ull) ? (B >> 63) & 1ull : (A >> 63) & 1ull);
            const uint32_t e1 = (uint32_t)(((odd1 >> 63) & 1ull) ? (B >> 63) & 1ull : (A >> 63) & 1ull);
            (void)incl;
            // compose: the segment so far (D_acc, map_acc), then this batch
            const uint32_t a0 = map_acc & 1u, a1 = (map_acc >> 1) & 1u;
            D_acc = make_double2(D_acc.x + (a0 ? Db.y : Db.x), D_acc.y + (a1 ? Db.y : Db.x));
            map_acc = (a0 ? e1 : e0) | ((a1 ? e1 : e0) << 1);
        }
        if (nhalf == 0) {  // the segment is complete: its pair would be kept for the batched store
            sum += lane == kWave - 1 ? D_acc.y * 1e-30 : 0.0;
            D_acc = make_double2(0.0, 0.0);
            map_acc = 2u;
        }
        seg = nseg;
        half = nhalf;
    }
    (void)nbatches;
    int32_t b = best[0] ^ best[1] ^ best[2] ^ best[3] ^ best[4];
    sum += (double)(b & 1) * 1e-30;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        sum += __shfl_down(sum, off, kWave);
    if (lane == 0)
        atomicAdd(out, sum);
    __syncthreads();
    for (uint32_t k = t; k < 64; k += WAVES * 64)
        atomicAdd(&ghist[k], (unsigned long long)hist[k] + hist[64 + k] + hist[128 + k] + hist[192 + k]);
}


__global__ __launch_bounds__(kThreads) void plain_kernel(const float4 *__restrict__ data, uint64_t ntiles, double *__restrict__ out)
{
    float acc = 0.f;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const float4 *p = data + tile * (uint64_t)(kThreads * U) + threadIdx.x;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 x[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            x[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p + (uint64_t)u * kThreads));
#pragma unroll
        for (int u = 0; u < U; u++)
            acc += x[u].x * x[u].x + x[u].y * x[u].y + x[u].z * x[u].z + x[u].w * x[u].w;
    }
    double sum = acc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        sum += __shfl_down(sum, off, kWave);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(out, sum);
}

__global__ void fill_kernel(float4 *data, uint64_t n4)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 20);
        auto f = [&](uint32_t k) {
            h = h * 1664525u + 1013904223u + k;
            return ((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * 0.7f;
        };
        data[i] = make_float4(f(1), f(2), f(3), f(4));
    }
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 10.0;
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30)) / (SEG_F4 * 16 * kWaves) * (SEG_F4 * 16 * kWaves);
    const uint64_t nsegs = bytes / (SEG_F4 * 16), ntiles = nsegs / kWaves;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int wgs = prop.multiProcessorCount;
    float4 *data;
    double2 *seg_D;
    double *out;
    hipMalloc(&data, bytes);
    hipMalloc(&seg_D, nsegs * sizeof(double2));
    hipMalloc(&out, 8);
    fill_kernel<<<wgs * 8, 256>>>(data, bytes / 16);
    hipDeviceSynchronize();
    const size_t lds = (size_t)kWaves * 8192;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[8] = {"0: the segment through LDS (DMA in, transposed read), powers summed in f32",
                            "1: + the two fp64 chains from the canonical entry states",
                            "2: + the pair composition (ballots + scalar prefix XOR + wave sum) and the pair's store",
                            "3: form 2 with the composition on the lanes (2-bit maps, six shuffle steps) instead of SALU",
                            "4: no LDS: plain 16-byte nontemporal loads, grid stride (the geometry's read ceiling)",
                            "5: form 2 with a segment's composition done at the top of the NEXT segment's fold (its loads in flight)",
                            "6: form 2 without the pair's store (the composition alone)",
                            "7: form 1 + a 16-byte store per segment by lane 63 (the store alone)"};
    // the mean power is 2 * 0.7^2 / 3 = 0.327 per sample: the running sum of 1.3e9 samples ends near 2^28.7; binade 28 is typical
    const int E = 28;
    printf("# exact_form_probe: %.2f GiB, %d workgroups x %d threads, %llu segments of 8 KiB; 20 timed launches after 5\n", bytes / 1073741824.0, wgs,
           kThreads, (unsigned long long)nsegs);
    for (int rep = 0; rep < 2; rep++)
        for (int form = 0; form < 8; form++) {
            std::vector<float> ms;
            double got = 0;
            for (int it = 0; it < 25; it++) {
                hipMemsetAsync(out, 0, 8, 0);
                hipEventRecord(e0, 0);
                switch (form) {
                case 0: form_kernel<0><<<wgs, kThreads, lds>>>(data, nsegs, E, seg_D, out); break;
                case 1: form_kernel<1><<<wgs, kThreads, lds>>>(data, nsegs, E, seg_D, out); break;
                case 2: form_kernel<2><<<wgs, kThreads, lds>>>(data, nsegs, E, seg_D, out); break;
                case 3: form_kernel<3><<<wgs, kThreads, lds>>>(data, nsegs, E, seg_D, out); break;
                case 5: form_kernel<5><<<wgs, kThreads, lds>>>(data, nsegs, E, seg_D, out); break;
                case 6: form_kernel<6><<<wgs, kThreads, lds>>>(data, nsegs, E, seg_D, out); break;
                case 7: form_kernel<7><<<wgs, kThreads, lds>>>(data, nsegs, E, seg_D, out); break;
                default: plain_kernel<<<wgs, kThreads>>>(data, ntiles, out); break;
                }
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float t;
                hipEventElapsedTime(&t, e0, e1);
                if (it >= 5)
                    ms.push_back(t);
            }
            hipMemcpy(&got, out, 8, hipMemcpyDeviceToHost);
            std::sort(ms.begin(), ms.end());
            double mean = 0;
            for (float v : ms)
                mean += v;
            mean /= ms.size();
            printf("pass %d form %-100s  min %.4f  median %.4f  mean %.4f ms  = %.0f GB/s = %.3f of 8 TB/s   (sum %.6e)\n", rep, names[form], ms.front(),
                   ms[ms.size() / 2], mean, bytes / (mean * 1e-3) / 1e9, bytes / (mean * 1e-3) / 1e9 / 8000.0, got);
        }
    // forms 2 and 3 must agree on every pair
    {
        std::vector<double2> a(4096), b(4096);
        hipMemsetAsync(out, 0, 8, 0);
        form_kernel<2><<<wgs, kThreads, lds>>>(data, nsegs, E, seg_D, out);
        hipMemcpy(a.data(), seg_D, a.size() * sizeof(double2), hipMemcpyDeviceToHost);
        form_kernel<3><<<wgs, kThreads, lds>>>(data, nsegs, E, seg_D, out);
        hipMemcpy(b.data(), seg_D, b.size() * sizeof(double2), hipMemcpyDeviceToHost);
        size_t bad = 0, differ = 0;
        for (size_t i = 0; i < a.size(); i++) {
            bad += a[i].x != b[i].x || a[i].y != b[i].y;
            differ += a[i].x != a[i].y;
        }
        printf("# pairs of the first %zu segments: forms 2 and 3 disagree on %zu; D0 != D1 in %zu of them\n", a.size(), bad, differ);
        form_kernel<5><<<wgs, kThreads, lds>>>(data, nsegs, E, seg_D, out);
        hipMemcpy(b.data(), seg_D, b.size() * sizeof(double2), hipMemcpyDeviceToHost);
        bad = 0;
        for (size_t i = 0; i < a.size(); i++)
            bad += a[i].x != b[i].x || a[i].y != b[i].y;
        printf("# ... forms 2 and 5 disagree on %zu\n", bad);
    }
    // ---- geometry A / B with pass 2's per-sample work in it ----
    {
        const uint32_t shift = 15, cell_lo = (0x3A800000u >> shift), ncells = 3584;  // 2^-10 ... 2^4 in cells of 2^15 bit patterns: 28 KiB
        std::vector<uint2> h_tab(ncells);
        for (uint32_t c = 0; c < ncells; c++) {  // ~one bin per 112 cells (32 bins), the threshold in the middle of every 112th cell
            h_tab[c].x = c / 112u;
            h_tab[c].y = (c % 112u == 111u) ? (((cell_lo + c) << shift) + (1u << (shift - 1))) : 0xFFFFFFFFu;
        }
        uint2 *d_tab;
        unsigned long long *d_hist;
        hipMalloc(&d_tab, ncells * sizeof(uint2));
        hipMalloc(&d_hist, 64 * 8);
        hipMemcpy(d_tab, h_tab.data(), ncells * sizeof(uint2), hipMemcpyHostToDevice);
        const size_t lds_a = ncells * 8 + 4 * 64 * 4 + (size_t)8 * 8192, lds_b = ncells * 8 + 4 * 64 * 4 + (size_t)16 * 4096;
        hipFuncSetAttribute((const void *)geom_kernel<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a);
        hipFuncSetAttribute((const void *)geom_kernel<16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        const char *gnames[2] = {"A: 8 waves per CU x 1024-sample segments (the product's geometry), pass 2's per-sample work in it",
                                 "B: 16 waves per CU x two 512-sample halves per segment through a 4 KiB buffer, the same work"};
        unsigned long long hsum[2] = {0, 0};
        for (int rep = 0; rep < 2; rep++)
            for (int g = 0; g < 2; g++) {
                std::vector<float> ms;
                for (int it = 0; it < 25; it++) {
                    hipMemsetAsync(out, 0, 8, 0);
                    hipMemsetAsync(d_hist, 0, 64 * 8, 0);
                    hipEventRecord(e0, 0);
                    if (g == 0)
                        geom_kernel<8, 8><<<wgs, 512, lds_a>>>(data, nsegs, E, d_tab, ncells, cell_lo, shift, out, d_hist);
                    else
                        geom_kernel<16, 4><<<wgs, 1024, lds_b>>>(data, nsegs, E, d_tab, ncells, cell_lo, shift, out, d_hist);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    float t;
                    hipEventElapsedTime(&t, e0, e1);
                    if (it >= 5)
                        ms.push_back(t);
                }
                unsigned long long hh[64];
                hipMemcpy(hh, d_hist, sizeof(hh), hipMemcpyDeviceToHost);
                hsum[g] = 0;
                for (int k = 0; k < 64; k++)
                    hsum[g] += hh[k] * (unsigned long long)(k + 1);
                std::sort(ms.begin(), ms.end());
                double mean = 0;
                for (float v : ms)
                    mean += v;
                mean /= ms.size();
                printf("pass %d geometry %-112s  min %.4f  median %.4f  mean %.4f ms  = %.0f GB/s = %.3f of 8 TB/s   (hist checksum %llu)%s\n", rep, gnames[g],
                       ms.front(), ms[ms.size() / 2], mean, bytes / (mean * 1e-3) / 1e9, bytes / (mean * 1e-3) / 1e9 / 8000.0, hsum[g],
                       hipGetLastError() == hipSuccess ? "" : "  LAUNCH ERROR");
            }
        printf("# histograms of A and B %s\n", hsum[0] == hsum[1] ? "agree" : "DIFFER");
    }
    return 0;
}
