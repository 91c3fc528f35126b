// stride_read_probe — does the read ceiling survive when a lane reads R CONSECUTIVE 16-byte words (R = 1, 2, 4, 8)
// instead of the wave reading 1 KiB rows?  Geometry of the exact-sum sweep: one persistent 512-thread workgroup per CU,
// a wave owns 8 KiB segments (1024 samples), eight 16-byte loads per lane in flight.  R = 1 is the coalesced form
// (lane l: words u*64 + l); with R > 1 lane l reads words row*(64R) + l*R + j — each instruction then touches 64R
// bytes-strided addresses, the R instructions of a row together cover whole 128-byte lines.  If R = 4 or 8 reads as fast,
// a lane's run of the sequential sum comes consecutive from memory and the LDS transposition can go.
// Measurement tool only; one JSON line per form.
//   hipcc --offload-arch=gfx950 -O3 tools/stride_read_probe.hip -o bin/stride_read_probe ; bin/stride_read_probe [GiB] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int R, bool NT, bool PREFETCH>
__global__ __launch_bounds__(512) void seg_read_kernel(const f32x4 *__restrict__ data, uint64_t nsegs, unsigned *out)
{
    constexpr int U = 8, WAVES = 8;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t stride = (uint64_t)gridDim.x * WAVES;
    unsigned acc = 0;
    auto load = [&](f32x4(&x)[U], uint64_t seg) {
        const f32x4 *base = data + seg * 512;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int row = u / R, j = u % R;
            const f32x4 *p = base + row * (64 * R) + lane * R + j;
            x[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
    };
    auto fold = [&](const f32x4(&x)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            u32x4 b = __builtin_bit_cast(u32x4, x[u]);
            acc ^= b.x ^ b.y ^ b.z ^ b.w;
        }
    };
    uint64_t seg = (uint64_t)blockIdx.x * WAVES + wave;
    if (PREFETCH) {
        f32x4 cur[U], nxt[U];
        if (seg < nsegs)
            load(cur, seg);
        for (; seg < nsegs; seg += stride) {
            if (seg + stride < nsegs)
                load(nxt, seg + stride);
            fold(cur);
#pragma unroll
            for (int u = 0; u < U; u++)
                cur[u] = nxt[u];
        }
    } else {
        for (; seg < nsegs; seg += stride) {
            f32x4 x[U];
            load(x, seg);
            fold(x);
        }
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}

__global__ void fill_kernel(unsigned *p, uint64_t n)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x)
        p[k] = (unsigned)(k * 2654435761u) >> 3;
}

template <int R, bool NT, bool PF>
static void run(const void *d, size_t bytes, unsigned *out, int blocks, int rounds)
{
    const uint64_t nsegs = bytes / 8192;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::vector<float> ms;
    for (int r = 0; r < rounds + 2; r++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((seg_read_kernel<R, NT, PF>), dim3(blocks), dim3(512), 0, 0, (const f32x4 *)d, nsegs, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float t;
        hipEventElapsedTime(&t, a, b);
        if (r >= 2)
            ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double used = (double)nsegs * 8192;
    printf("{\"probe\": \"segment read\", \"consecutive_words_per_lane\": %d, \"lane_stride_bytes\": %d, \"nt\": %d, \"prefetch\": %d, "
           "\"blocks\": %d, \"median_ms\": %.4f, \"min_ms\": %.4f, \"GB/s_median\": %.1f, \"frac_of_8TBs\": %.4f}\n",
           R, 16 * R, (int)NT, (int)PF, blocks, ms[ms.size() / 2], ms[0], used / ms[ms.size() / 2] / 1e6,
           used / ms[ms.size() / 2] / 1e6 / 8000.0);
    fflush(stdout);
    hipEventDestroy(a);
    hipEventDestroy(b);
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 10.0;
    const int rounds = argc > 2 ? atoi(argv[2]) : 10;
    const size_t bytes = (size_t)(gib * (1 << 30)) / 65536 * 65536;
    void *d;
    unsigned *out;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc((void **)&out, 4) != hipSuccess) {
        fprintf(stderr, "hipMalloc failed\n");
        return 1;
    }
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned *)d, bytes / 4);
    hipDeviceSynchronize();
    for (int pass = 0; pass < 2; pass++) {
        run<1, true, true>(d, bytes, out, cus, rounds);
        run<2, true, true>(d, bytes, out, cus, rounds);
        run<4, true, true>(d, bytes, out, cus, rounds);
        run<8, true, true>(d, bytes, out, cus, rounds);
        run<1, false, true>(d, bytes, out, cus, rounds);
        run<2, false, true>(d, bytes, out, cus, rounds);
        run<4, false, true>(d, bytes, out, cus, rounds);
        run<8, false, true>(d, bytes, out, cus, rounds);
    }
    run<1, true, false>(d, bytes, out, cus, rounds);
    run<4, true, false>(d, bytes, out, cus, rounds);
    run<8, true, false>(d, bytes, out, cus, rounds);
    run<4, true, true>(d, bytes, out, 2 * cus, rounds);
    run<8, true, true>(d, bytes, out, 2 * cus, rounds);
    run<4, false, true>(d, bytes, out, 2 * cus, rounds);
    hipFree(d);
    hipFree(out);
    return 0;
}
