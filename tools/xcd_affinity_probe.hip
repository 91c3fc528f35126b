// xcd_affinity_probe — is some memory nearer to some XCDs?  The buffer is cut into chunks of G bytes, chunk c belongs to class
// c mod 8 (mod 4, mod 16), and every workgroup reads only the chunks of ONE class: class = (the XCD it runs on + shift) mod
// classes.  If HBM stacks / channels are interleaved at granularity G and an XCD reaches some of them faster, the kernel's time
// depends on `shift`; if nothing depends on it, there is no affinity to exploit at that granularity.  One persistent 512-thread
// workgroup per CU, eight 16-byte nontemporal loads in flight per lane, the whole buffer read once per launch.
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_affinity_probe.hip -o bin/xcd_affinity_probe ; bin/xcd_affinity_probe [GiB] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void read_classes(const unsigned char *__restrict__ data, uint64_t nbytes, uint64_t G, uint32_t classes,
                                                     uint32_t shift, float *sink, unsigned *xcd_out)
{
    const uint32_t xcd = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu;
    const uint32_t cls = (xcd + shift) % classes;
    // the workgroups that share this class: those on XCDs x with (x + shift) % classes == cls — 8 / classes XCDs (classes <= 8)
    // or, for classes = 16, half a class each; rank among them from blockIdx (workgroup b runs on XCD (b + first) % 8)
    const uint32_t per_xcd = gridDim.x / 8, mine = blockIdx.x / 8;  // rank within my XCD
    uint32_t share_n, share_i;
    if (classes <= 8) {
        const uint32_t xcds_per_class = 8 / classes;
        share_n = per_xcd * xcds_per_class;
        share_i = mine * xcds_per_class + (xcd / classes);
    } else {
        share_n = per_xcd;
        share_i = mine;
    }
    const uint64_t nchunks_class = nbytes / G / classes;       // chunks of my class
    const uint64_t vbytes = nchunks_class * G;                 // my class as one virtual stream
    float acc = 0.f;
    constexpr uint32_t kTile = 512 * 8 * 16;  // 64 KiB per workgroup iteration
    for (uint64_t v0 = (uint64_t)share_i * kTile; v0 < vbytes; v0 += (uint64_t)share_n * kTile) {
        f32x4 x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint64_t v = v0 + ((uint64_t)u * 512 + threadIdx.x) * 16;
            const uint64_t q = v / G, within = v % G;
            const uint64_t phys = (q * classes + cls) * G + within;
            x[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(data + (v < vbytes ? phys : 0)));
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            acc += x[u].x + x[u].y + x[u].z + x[u].w;
    }
    if (acc == 12345.678f)
        sink[0] = acc;
    if (threadIdx.x == 0)
        xcd_out[blockIdx.x] = xcd;
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const int rounds = argc > 2 ? atoi(argv[2]) : 7;
    const uint64_t nbytes = ((uint64_t)(gib * (1ull << 30))) & ~((1ull << 26) - 1);
    unsigned char *d;
    float *sink;
    unsigned *xcd;
    if (hipMalloc((void **)&d, nbytes) != hipSuccess || hipMalloc((void **)&sink, 64) != hipSuccess || hipMalloc((void **)&xcd, 4096) != hipSuccess)
        return 1;
    hipMemset(d, 0, nbytes);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    printf("# %.1f GiB, %d workgroups; buffer at %p\n", nbytes / 1073741824.0, cus, (void *)d);
    const uint64_t Gs[] = {256, 1024, 4096, 8192, 16384, 65536, 1u << 20, 2u << 20, 16u << 20};
    for (uint32_t classes : {8u, 4u, 2u, 16u})
        for (uint64_t G : Gs) {
            if (nbytes / G / classes < (uint64_t)cus * 64)
                continue;
            printf("classes %2u G %8llu:", classes, (unsigned long long)G);
            double lo = 1e9, hi = 0;
            for (uint32_t shift = 0; shift < classes; shift++) {
                std::vector<float> ms;
                for (int i = 0; i < rounds + 2; i++) {
                    hipEventRecord(a);
                    hipLaunchKernelGGL(read_classes, dim3(cus), dim3(512), 0, 0, d, nbytes, G, classes, shift, sink, xcd);
                    hipEventRecord(b);
                    hipEventSynchronize(b);
                    float m;
                    hipEventElapsedTime(&m, a, b);
                    if (i >= 2)
                        ms.push_back(m);
                }
                std::sort(ms.begin(), ms.end());
                const double gbs = (double)nbytes / ms[ms.size() / 2] / 1e6;
                printf(" %5.0f", gbs);
                lo = std::min(lo, gbs);
                hi = std::max(hi, gbs);
            }
            printf("   GB/s by shift  (spread %.1f %%)\n", (hi - lo) / hi * 100.0);
            fflush(stdout);
        }
    return 0;
}
