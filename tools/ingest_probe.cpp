// ingest_probe — where does file -> HBM time go on this host?  Measures, separately:
// context creation, hipMalloc of the shard, pinned allocation, pinned->device copy
// bandwidth, and parallel pread bandwidth from a file into pinned memory.
//   hipcc -O2 tools/ingest_probe.cpp -o bin/ingest_probe -lpthread ;  bin/ingest_probe FILE
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: ingest_probe FILE\n");
        return 2;
    }
    double t0 = now();
    hipSetDevice(0);
    hipFree(nullptr);
    double t1 = now();
    printf("{\"context_s\": %.4f", t1 - t0);
    int fd = open(argv[1], O_RDONLY);
    struct stat sb;
    fstat(fd, &sb);
    size_t bytes = (size_t)sb.st_size;
    void *d = nullptr;
    t0 = now();
    hipMalloc(&d, bytes);
    hipDeviceSynchronize();
    t1 = now();
    printf(", \"hipMalloc_GiB\": %.2f, \"hipMalloc_s\": %.4f", bytes / 1073741824.0, t1 - t0);
    const size_t chunk = 256u << 20;
    void *h[2];
    t0 = now();
    hipHostMalloc(&h[0], chunk, hipHostMallocDefault);
    hipHostMalloc(&h[1], chunk, hipHostMallocDefault);
    t1 = now();
    printf(", \"pinned_2x256MiB_s\": %.4f", t1 - t0);
    // touch
    for (size_t k = 0; k < chunk; k += 4096) ((char *)h[0])[k] = 1, ((char *)h[1])[k] = 1;
    hipStream_t s;
    hipStreamCreate(&s);
    for (size_t sz : {(size_t)16 << 20, (size_t)64 << 20, (size_t)256 << 20}) {
        hipMemcpyAsync(d, h[0], sz, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        t0 = now();
        for (int r = 0; r < 8; r++)
            hipMemcpyAsync((char *)d + (size_t)r * sz, h[r & 1], sz, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        t1 = now();
        printf(", \"h2d_%zuMiB_GBs\": %.1f", sz >> 20, 8.0 * sz / (t1 - t0) / 1e9);
    }
    for (int nthr : {1, 4, 8, 16, 32, 64}) {
        size_t total = std::min(bytes, (size_t)4 << 30), done = 0;
        t0 = now();
        while (done < total) {
            size_t cnt = std::min(chunk, total - done);
            std::vector<std::thread> th;
            size_t per = (cnt + nthr - 1) / nthr;
            for (int t = 0; t < nthr; t++) {
                size_t a = std::min((size_t)t * per, cnt), e = std::min(a + per, cnt);
                if (e > a)
                    th.emplace_back([=] {
                        size_t off = a;
                        while (off < e) {
                            ssize_t g = pread(fd, (char *)h[0] + off, e - off, (off_t)(done + off));
                            if (g <= 0) break;
                            off += (size_t)g;
                        }
                    });
            }
            for (auto &x : th) x.join();
            done += cnt;
        }
        t1 = now();
        printf(", \"pread_%dthr_GBs\": %.1f", nthr, total / (t1 - t0) / 1e9);
    }
    // mmap + hipHostRegister: DMA straight out of the page cache, no staging copy
    {
        const size_t total = std::min(bytes, (size_t)4 << 30);
        void *m = mmap(nullptr, total, PROT_READ, MAP_SHARED, fd, 0);
        if (m != MAP_FAILED) {
            const size_t piece = 256u << 20;
            double treg = 0, tcopy = 0, tunreg = 0;
            bool ok = true;
            for (size_t off = 0; off < total && ok; off += piece) {
                const size_t len = std::min(piece, total - off);
                t0 = now();
                hipError_t e = hipHostRegister((char *)m + off, len, hipHostRegisterDefault);
                t1 = now();
                if (e != hipSuccess) {
                    printf(", \"mmap_register_error\": \"%s\"", hipGetErrorString(e));
                    ok = false;
                    break;
                }
                treg += t1 - t0;
                t0 = now();
                hipMemcpyAsync((char *)d + off, (char *)m + off, len, hipMemcpyHostToDevice, s);
                hipStreamSynchronize(s);
                t1 = now();
                tcopy += t1 - t0;
                t0 = now();
                hipHostUnregister((char *)m + off);
                tunreg += now() - t0;
            }
            if (ok)
                printf(", \"mmap_register_GBs\": %.1f, \"mmap_h2d_GBs\": %.1f, \"mmap_unregister_GBs\": %.1f, \"mmap_total_GBs\": %.1f",
                       total / treg / 1e9, total / tcopy / 1e9, total / tunreg / 1e9, total / (treg + tcopy + tunreg) / 1e9);
            munmap(m, total);
        }
    }
    // does page pinning scale across threads?  N threads register disjoint 256 MiB pieces of a FRESH 2 GiB
    // window each time (unregistering is lazy, a window registered before would be free the second time)
    {
        void *m = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
        if (m != MAP_FAILED) {
            const size_t piece = 256u << 20, window = (size_t)2 << 30;
            size_t off = (size_t)4 << 30;  // [0, 4 GiB) was used above
            for (int nthr : {2, 4, 8}) {
                if (off + window > bytes)
                    break;
                const int npieces = (int)(window / piece);
                char *base = (char *)m + off;
                std::vector<std::thread> th;
                t0 = now();
                for (int t = 0; t < nthr; t++)
                    th.emplace_back([=] {
                        hipSetDevice(0);
                        for (int k = t; k < npieces; k += nthr)
                            hipHostRegister(base + (size_t)k * piece, piece, hipHostRegisterDefault);
                    });
                for (auto &x : th) x.join();
                t1 = now();
                printf(", \"register_%dthr_GBs\": %.1f", nthr, (double)window / (t1 - t0) / 1e9);
                for (int k = 0; k < npieces; k++)
                    hipHostUnregister(base + (size_t)k * piece);
                off += window;
            }
            munmap(m, bytes);
        }
    }
    printf("}\n");
    return 0;
}
