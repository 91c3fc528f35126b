// zero_copy_probe — can a KERNEL pull a pinned host buffer over the link as fast as the copy engine does?  (If so the
// ingest's pass-1 kernel could read every chunk straight out of the staging buffer and store it into the shard itself:
// no copy engine, no per-copy overhead, no events between two streams.)  For each grid: 16 MiB chunks out of four
// pinned buffers, back to back on one stream, for `secs` seconds; beside it the same with hipMemcpyAsync.
//   hipcc --offload-arch=gfx950 -O3 tools/zero_copy_probe.hip -o bin/zero_copy_probe ; bin/zero_copy_probe [secs]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(256) void pull_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n16; i += stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            v[u] = i + u * 256 < n16 ? __builtin_nontemporal_load(src + i + u * 256) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; u++)
            if (i + u * 256 < n16)
                dst[i + u * 256] = v[u];
    }
}

int main(int argc, char **argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 0.5;
    hipSetDevice(0);
    const size_t chunk = 16u << 20;
    void *h[4], *hm[4], *d;
    for (int b = 0; b < 4; b++) {
        hipHostMalloc(&h[b], chunk, hipHostMallocDefault);
        memset(h[b], b + 1, chunk);
        hipHostGetDevicePointer(&hm[b], h[b], 0);
    }
    hipMalloc(&d, (size_t)1 << 30);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    auto run = [&](const char *name, int grid, int unroll) {
        const double t0 = now();
        unsigned long long copied = 0;
        unsigned n = 0;
        while (now() - t0 < secs) {
            for (int k = 0; k < 8; k++, n++) {
                char *dst = (char *)d + (size_t)(n % 64) * chunk;
                if (grid == 0)
                    hipMemcpyAsync(dst, h[n % 4], chunk, hipMemcpyHostToDevice, s);
                else if (unroll == 4)
                    hipLaunchKernelGGL(pull_kernel<4>, dim3(grid), dim3(256), 0, s, (const u32x4 *)hm[n % 4], (u32x4 *)dst, chunk / 16);
                else
                    hipLaunchKernelGGL(pull_kernel<8>, dim3(grid), dim3(256), 0, s, (const u32x4 *)hm[n % 4], (u32x4 *)dst, chunk / 16);
                copied += chunk;
            }
            hipStreamSynchronize(s);
        }
        const double t1 = now();
        printf("{\"form\": \"%s\", \"grid\": %d, \"loads_per_lane\": %d, \"GBps\": %.1f}\n", name, grid, unroll, copied / (t1 - t0) / 1e9);
        fflush(stdout);
    };
    run("hipMemcpyAsync", 0, 0);
    for (int grid : {16, 32, 64, 128, 256, 512, 1024})
        for (int u : {4, 8})
            run("kernel pull", grid, u);
    run("hipMemcpyAsync", 0, 0);
    // did the bytes arrive?
    unsigned char probe[4];
    hipMemcpy(probe, (char *)d + 5 * chunk + 12345, 4, hipMemcpyDeviceToHost);
    printf("{\"check\": [%u, %u, %u, %u]}\n", probe[0], probe[1], probe[2], probe[3]);
    return 0;
}
