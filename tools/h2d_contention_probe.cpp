// h2d_contention_probe — what the H2D copy engine gives the ingest's copy pattern (16 MiB pinned -> device copies back to
// back on one stream, an event behind each) alone, and while reader threads fill OTHER pinned buffers from a page-cache
// file at the same time, by reader form:
//   pread      pread(2) into the pinned buffer (the kernel's copy_to_user)                       — what the ingest does
//   mmap       memcpy out of a MAP_SHARED mapping of the file (user-space copy, cached stores)
//   mmap_nt    the same with non-temporal 32-byte stores (no read-for-ownership, no cache pollution)
// Prints one JSON line per configuration: copy GB/s, read GB/s (both directions run unsynchronised for `secs` seconds).
//   hipcc -O2 -mavx2 tools/h2d_contention_probe.cpp -o bin/h2d_contention_probe -lpthread
//   bin/h2d_contention_probe FILE [secs]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <immintrin.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void copy_nt(void *dst, const void *src, size_t n)
{
    const __m256i *s = (const __m256i *)src;
    __m256i *d = (__m256i *)dst;
    for (size_t k = 0; k < n / 32; k += 4) {
        const __m256i a = _mm256_loadu_si256(s + k), b = _mm256_loadu_si256(s + k + 1), c = _mm256_loadu_si256(s + k + 2),
                      e = _mm256_loadu_si256(s + k + 3);
        _mm256_stream_si256(d + k, a);
        _mm256_stream_si256(d + k + 1, b);
        _mm256_stream_si256(d + k + 2, c);
        _mm256_stream_si256(d + k + 3, e);
    }
    _mm_sfence();
}

static bool cpus_near_gpu(cpu_set_t *set)
{
    char bus[64] = {0}, path[160];
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), 0) != hipSuccess)
        return false;
    for (char *p = bus; *p; p++)
        *p = (char)tolower(*p);
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bus);
    FILE *f = fopen(path, "r");
    if (!f)
        return false;
    char line[4096] = {0};
    const bool got = fgets(line, sizeof(line), f) != nullptr;
    fclose(f);
    if (!got)
        return false;
    CPU_ZERO(set);
    int n = 0;
    for (char *tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a, b;
        if (sscanf(tok, "%d-%d", &a, &b) == 2) {
            for (int c = a; c <= b; c++)
                CPU_SET(c, set), n++;
        } else if (sscanf(tok, "%d", &a) == 1)
            CPU_SET(a, set), n++;
    }
    return n > 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: h2d_contention_probe FILE [secs]\n");
        return 2;
    }
    const double secs = argc > 2 ? atof(argv[2]) : 0.6;
    hipSetDevice(0);
    hipFree(nullptr);
    cpu_set_t near;
    const bool have_near = cpus_near_gpu(&near);
    if (have_near)
        sched_setaffinity(0, sizeof(near), &near);  // (threads created below inherit it)
    const int fd = open(argv[1], O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb)) {
        perror(argv[1]);
        return 1;
    }
    const size_t bytes = (size_t)sb.st_size;
    const char *map = (const char *)mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
    void *d = nullptr;
    hipMalloc(&d, (size_t)1 << 30);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    printf("{\"file_GiB\": %.2f, \"near_gpu_cpus\": %d, \"secs\": %.2f}\n", bytes / 1073741824.0, have_near ? CPU_COUNT(&near) : 0, secs);
    for (size_t chunk : {(size_t)16 << 20, (size_t)64 << 20}) {
        constexpr int NB = 4, RB = 4;
        void *h[NB], *r[RB];
        hipEvent_t ev[NB];
        for (int b = 0; b < NB; b++) {
            hipHostMalloc(&h[b], chunk, hipHostMallocDefault);
            memset(h[b], 1, chunk);
            hipEventCreateWithFlags(&ev[b], hipEventDisableTiming);
        }
        for (int b = 0; b < RB; b++) {
            hipHostMalloc(&r[b], chunk, hipHostMallocDefault);
            memset(r[b], 1, chunk);
        }
        struct Cfg {
            const char *form;
            int threads;
        };
        const Cfg cfgs[] = {{"none", 0},     {"pread", 4},   {"pread", 8},    {"pread", 16},  {"pread", 32},
                            {"mmap", 8},     {"mmap", 16},   {"mmap_nt", 8},  {"mmap_nt", 16}, {"mmap_nt", 32}, {"none", 0}};
        for (const Cfg &c : cfgs) {
            std::atomic<bool> stop{false};
            std::atomic<unsigned long long> read_bytes{0}, next_piece{0};
            const size_t piece = 1u << 20, npieces = bytes / piece;
            std::vector<std::thread> th;
            const int form = !strcmp(c.form, "pread") ? 0 : !strcmp(c.form, "mmap") ? 1 : 2;
            for (int t = 0; t < c.threads; t++)
                th.emplace_back([&, t] {
                    unsigned long long mine = 0;
                    while (!stop.load(std::memory_order_relaxed)) {
                        const size_t k = (size_t)(next_piece.fetch_add(1) % npieces);
                        char *dst = (char *)r[(k / (chunk / piece)) % RB] + (k % (chunk / piece)) * piece;
                        if (form == 0) {
                            if (pread(fd, dst, piece, (off_t)(k * piece)) != (ssize_t)piece)
                                break;
                        } else if (form == 1)
                            memcpy(dst, map + k * piece, piece);
                        else
                            copy_nt(dst, map + k * piece, piece);
                        mine += piece;
                    }
                    read_bytes += mine;
                });
            if (c.threads)
                std::this_thread::sleep_for(std::chrono::milliseconds(50));
            // the ingest's copy pattern: at most two copies queued, a buffer reused once its copy's event is through
            const double t0 = now();
            unsigned long long copied = 0;
            unsigned n = 0;
            while (now() - t0 < secs) {
                const int b = n % NB;
                if (n >= 2)
                    hipEventSynchronize(ev[(n - 2) % NB]);
                hipMemcpyAsync((char *)d + (size_t)(n % ((size_t)(1 << 30) / chunk)) * chunk, h[b], chunk, hipMemcpyHostToDevice, s);
                hipEventRecord(ev[b], s);
                copied += chunk;
                n++;
            }
            hipStreamSynchronize(s);
            const double t1 = now();
            stop = true;
            for (auto &x : th)
                x.join();
            const double t2 = now();
            printf("{\"chunk_MiB\": %zu, \"readers\": \"%s\", \"threads\": %d, \"h2d_GBps\": %.1f, \"read_GBps\": %.1f}\n", chunk >> 20, c.form,
                   c.threads, copied / (t1 - t0) / 1e9, read_bytes.load() / (t2 - t0 + 0.05) / 1e9);
            fflush(stdout);
        }
        for (int b = 0; b < NB; b++)
            hipHostFree(h[b]), hipEventDestroy(ev[b]);
        for (int b = 0; b < RB; b++)
            hipHostFree(r[b]);
    }
    return 0;
}
