// wg_skew_probe — do the chip's XCDs start and finish a persistent-workgroup read kernel together?
// One workgroup of 512 threads per CU (workgroup i runs on XCD i mod 8) streams its share of a buffer with 16-byte nontemporal
// loads, eight in flight per lane, and records the 100 MHz real-time counter when it starts and when it is done; shares are
// either one contiguous range per workgroup or tiles taken round robin (grid stride) or tiles taken from ONE atomic counter.
//   hipcc --offload-arch=gfx950 -O3 tools/wg_skew_probe.hip -o bin/wg_skew_probe && bin/wg_skew_probe [GiB]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kThreads = 512, kLoads = 8;
constexpr uint64_t kTileF4 = (uint64_t)kThreads * kLoads;  // 64 KiB

__global__ __launch_bounds__(kThreads) void read_kernel(const float4 *__restrict__ data, uint64_t ntiles, int mode,
                                                         unsigned long long *__restrict__ next, unsigned long long *__restrict__ ticks,
                                                         float *__restrict__ sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    __shared__ unsigned long long s_tile;
    auto fold = [&](uint64_t tile) {
        const float4 *p = data + tile * kTileF4 + threadIdx.x;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 x[kLoads];
#pragma unroll
        for (int u = 0; u < kLoads; u++)
            x[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p + (uint64_t)u * kThreads));
#pragma unroll
        for (int u = 0; u < kLoads; u++)
            acc += x[u].x * x[u].x + x[u].y * x[u].y + x[u].z * x[u].z + x[u].w * x[u].w;
    };
    if (mode == 0) {  // contiguous share
        const uint64_t per = (ntiles + gridDim.x - 1) / gridDim.x, a = per * blockIdx.x, b = a + per < ntiles ? a + per : ntiles;
        for (uint64_t t = a; t < b; t++)
            fold(t);
    } else if (mode == 1) {  // grid stride
        for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x)
            fold(t);
    } else {  // one atomic counter, four tiles at a time
        for (;;) {
            if (threadIdx.x == 0)
                s_tile = atomicAdd(next, 4ull);
            __syncthreads();
            const uint64_t t = s_tile;
            __syncthreads();
            if (t >= ntiles)
                break;
            for (uint64_t k = t; k < t + 4 && k < ntiles; k++)
                fold(k);
        }
    }
    if (acc == 12345.678f)
        sink[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        ticks[2 * blockIdx.x] = t0;
        ticks[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 10.0;
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30)) / (kTileF4 * 16) * (kTileF4 * 16), ntiles = bytes / (kTileF4 * 16);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int wgs = prop.multiProcessorCount;
    float4 *data;
    unsigned long long *ticks, *next;
    float *sink;
    hipMalloc(&data, bytes);
    hipMemset(data, 0, bytes);
    hipMalloc(&ticks, 2 * wgs * sizeof(unsigned long long));
    hipMalloc(&next, 8);
    hipMalloc(&sink, 4);
    std::vector<unsigned long long> h(2 * wgs);
    const char *names[3] = {"contiguous shares", "grid stride      ", "one atomic counter"};
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 12; rep++) {
            hipMemset(next, 0, 8);
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipEventRecord(a);
            hipLaunchKernelGGL(read_kernel, dim3(wgs), dim3(kThreads), 0, 0, data, ntiles, mode, next, ticks, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            if (rep < 9)
                continue;
            hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost);
            unsigned long long s0 = ~0ull, e0 = ~0ull, e1 = 0;
            for (int w = 0; w < wgs; w++) {
                s0 = std::min(s0, h[2 * w]);
                e0 = std::min(e0, h[2 * w + 1]);
                e1 = std::max(e1, h[2 * w + 1]);
            }
            double start_x[8] = {0}, end_x[8] = {0};
            int cnt[8] = {0};
            for (int w = 0; w < wgs; w++) {
                start_x[w % 8] += (h[2 * w] - s0) / 100.0;
                end_x[w % 8] += (h[2 * w + 1] - s0) / 100.0;
                cnt[w % 8]++;
            }
            printf("%s  %.4f ms = %.0f GB/s; first start -> last end %.1f us, first end -> last end %.1f us; per XCD start [", names[mode], ms,
                   bytes / (ms * 1e-3) / 1e9, (e1 - s0) / 100.0, (e1 - e0) / 100.0);
            for (int x = 0; x < 8; x++)
                printf("%.1f%s", start_x[x] / cnt[x], x < 7 ? " " : "] end [");
            for (int x = 0; x < 8; x++)
                printf("%.1f%s", end_x[x] / cnt[x], x < 7 ? " " : "]\n");
        }
    }
    return 0;
}
