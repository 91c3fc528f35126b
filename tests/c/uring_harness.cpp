// uring_harness.cpp — GPU-free check of the ingest's io_uring reader (dtv-utils_amd/csrc/papr_uring.h) against pread:
//   uring_harness <file> <direct 0|1> <piece bytes> <chunk bytes>
// reads the whole file chunk by chunk the way stream_file does (whole 4 KiB blocks through the O_DIRECT descriptor when
// direct = 1 and the filesystem has one, the rest buffered; two chunks in flight), and compares every byte.
// prints "unsupported" (exit 0) where the kernel refuses io_uring_setup.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>

#include <vector>

#include "papr_uring.h"

using namespace papr_rt;

int main(int argc, char **argv)
{
    if (argc < 5)
        return 2;
    const char *path = argv[1];
    const int want_direct = atoi(argv[2]);
    const uint64_t piece = strtoull(argv[3], nullptr, 0), chunk = strtoull(argv[4], nullptr, 0);
    UringReader *ring = UringReader::create(8);  // a small ring on purpose: the backlog path gets used
    if (!ring) {
        printf("unsupported\n");
        return 0;
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0)
        return 3;
    struct stat sb;
    fstat(fd, &sb);
    const uint64_t size = (uint64_t)sb.st_size;
    const int fdd = want_direct ? open(path, O_RDONLY | O_DIRECT) : -1;
    const uint64_t nchunks = (size + chunk - 1) / chunk;
    unsigned char *buf[2];
    for (int b = 0; b < 2; b++)
        if (posix_memalign((void **)&buf[b], 4096, chunk + 4096))
            return 4;
    std::vector<unsigned char> ref(chunk);
    std::vector<ReadBatch> batches(nchunks);
    auto submit = [&](uint64_t c) {
        const uint64_t off = c * chunk, want = size - off < chunk ? size - off : chunk;
        memset(buf[c & 1], 0xEE, chunk + 4096);
        const uint64_t direct = (fdd >= 0 && (off & 4095) == 0) ? (want & ~4095ull) : 0;
        if (direct)
            ring->submit(&batches[c], fdd, fd, off, direct, buf[c & 1], piece);
        if (want > direct)
            ring->submit(&batches[c], fd, fd, off + direct, want - direct, buf[c & 1] + direct, piece);
    };
    uint64_t bad = 0;
    if (nchunks)
        submit(0);
    for (uint64_t c = 0; c < nchunks; c++) {
        if (c + 1 < nchunks)
            submit(c + 1);  // read ahead, like the ingest
        if (ring->wait(&batches[c])) {
            printf("error %d in chunk %llu\n", batches[c].error, (unsigned long long)c);
            return 5;
        }
        const uint64_t off = c * chunk, want = size - off < chunk ? size - off : chunk;
        if (pread(fd, ref.data(), want, (off_t)off) != (ssize_t)want)
            return 6;
        if (memcmp(ref.data(), buf[c & 1], want) != 0)
            bad++;
        for (uint64_t k = want; k < want + 64; k++)  // nothing written past what was asked for
            if (buf[c & 1][k] != 0xEE)
                bad++;
    }
    delete ring;
    printf("%s %llu bytes %llu chunks direct=%d bad=%llu\n", bad ? "MISMATCH" : "ok", (unsigned long long)size,
           (unsigned long long)nchunks, fdd >= 0, (unsigned long long)bad);
    return bad ? 1 : 0;
}
