// ts_stride_probe — the read ceiling of the packet scan's access pattern: 8 aligned bytes at a stride of 188, one lane per
// packet, one persistent workgroup per CU over a contiguous span, a barrier per block of packets — with 1, 2 or 3 blocks'
// header words in flight per lane.  The scan kernel keeps ONE block ahead (ts_kernels.hip: pre_w0 / pre_w1): does the memory
// system stand still between the batches?  Measurement tool only.
//   hipcc --offload-arch=gfx950 -O3 tools/ts_stride_probe.hip -o bin/ts_stride_probe ; bin/ts_stride_probe [GiB] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int BLOCK, int DEPTH, int WORK>
__global__ __launch_bounds__(BLOCK) void scan_like(const unsigned char *__restrict__ data, uint64_t npackets, unsigned *out)
{
    __shared__ unsigned s_stop, s_tab[64];
    const uint32_t t = threadIdx.x;
    const uint64_t per = (npackets + gridDim.x - 1) / gridDim.x;
    const uint64_t k0 = per * blockIdx.x, k1 = k0 + per < npackets ? k0 + per : npackets;
    if (t < 64)
        s_tab[t] = 0;
    if (t == 0)
        s_stop = 0xFFFFFFFFu;
    __syncthreads();
    unsigned acc = 0;
    uint32_t w0[DEPTH], w1[DEPTH];
    auto ask = [&](int slot, uint64_t k) {
        const uint64_t a = k + t < k1 ? ((k + t) * 188ull) & ~3ull : 0ull;
        w0[slot] = *reinterpret_cast<const uint32_t *>(data + a);
        w1[slot] = *reinterpret_cast<const uint32_t *>(data + a + 4);
    };
    uint64_t k = k0;
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        ask(d, k + (uint64_t)d * BLOCK);
    while (k < k1) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {  // (static slots: the compiler can wait for the oldest batch alone)
            if (k >= k1)
                break;
            const uint32_t a = w0[d], b = w1[d];
            if ((a & 0xffu) == 0x00u && (b >> 24) == 0x99u)  // (never: the stream's bytes are a counter pattern)
                atomicMin(&s_stop, t);
            __syncthreads();
            if (s_stop != 0xFFFFFFFFu)
                acc ^= 1u;
            ask(d, k + (uint64_t)DEPTH * BLOCK);
            acc ^= a ^ b;
            if (WORK) {  // what a block's commit costs, roughly: three LDS atomics per packet on a handful of addresses + two barriers
                const uint32_t pid = (a >> 8) & 7u;
                atomicAdd(&s_tab[pid], 1u);
                atomicMin(&s_tab[8 + pid], a);
                atomicMax(&s_tab[16 + pid], b);
                __syncthreads();
                __syncthreads();
            }
            k += BLOCK;
        }
    }
    if (acc == 0x12345678u || s_tab[t & 63] == 0xdeadbeefu)
        out[0] = acc;
}

template <int BLOCK, int DEPTH, int WORK>
static void run(const unsigned char *d, uint64_t npackets, int blocks, int rounds, unsigned *out)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; i++)
        hipLaunchKernelGGL((scan_like<BLOCK, DEPTH, WORK>), dim3(blocks), dim3(BLOCK), 0, 0, d, npackets, out);
    std::vector<float> ms;
    for (int i = 0; i < rounds; i++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((scan_like<BLOCK, DEPTH, WORK>), dim3(blocks), dim3(BLOCK), 0, 0, d, npackets, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float m;
        hipEventElapsedTime(&m, a, b);
        ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    const double med = ms[ms.size() / 2], bytes = (double)npackets * 1.03125 * 128.0;
    printf("{\"block\": %d, \"workgroups\": %d, \"batches_in_flight\": %d, \"commit_work\": %d, \"ms\": %.4f, \"header_lines_GBps\": %.0f, \"frac_of_8TBps\": %.3f}\n",
           BLOCK, blocks, DEPTH, WORK, med, bytes / med / 1e6, bytes / med / 1e6 / 8000.0);
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 10.0;
    const int rounds = argc > 2 ? atoi(argv[2]) : 15;
    const uint64_t npackets = (uint64_t)(gib * (1ull << 30)) / 188;
    unsigned char *d;
    unsigned *out;
    if (hipMalloc((void **)&d, npackets * 188 + 4096) != hipSuccess || hipMalloc((void **)&out, 64) != hipSuccess)
        return 1;
    hipMemset(d, 0x5a, npackets * 188 + 4096);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    run<1024, 1, 0>(d, npackets, cus, rounds, out);
    run<1024, 2, 0>(d, npackets, cus, rounds, out);
    run<1024, 3, 0>(d, npackets, cus, rounds, out);
    run<1024, 1, 1>(d, npackets, cus, rounds, out);
    run<1024, 2, 1>(d, npackets, cus, rounds, out);
    run<1024, 3, 1>(d, npackets, cus, rounds, out);
    run<512, 1, 1>(d, npackets, 2 * cus, rounds, out);
    run<512, 2, 1>(d, npackets, 2 * cus, rounds, out);
    return 0;
}
