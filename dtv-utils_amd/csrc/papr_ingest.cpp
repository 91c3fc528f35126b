// papr_ingest.cpp — the file ingest engine of the host runtime (replaces the fread loops of reference papr.c:100-101,
// 143-144, 175-176): reader threads -> pinned staging buffers -> hipMemcpyAsync -> the pass kernel on every chunk as it
// lands; resident and re-streamed shards; the one-sweep ingest and the mean estimate of a file that is not loaded yet.

#include "papr_runtime_internal.h"
#include "papr_uring.h"

using namespace papr_rt;

namespace papr_rt {

// (computed once per device and cached: bin/papr drives one thread per GPU shard through here concurrently)
static CpuSet numa_cpus_of_device_uncached(int device)
{
    CpuSet out;
    CPU_ZERO(&out.set);
    const char *env = getenv("PAPR_NUMA");
    if (env && env[0] == '0')
        return out;
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) {
        (void)hipGetLastError();
        return out;
    }
    for (char *c = bus; *c; c++)
        *c = (char)tolower(*c);
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *fp = fopen(path, "r");
    int node = -1;
    if (!fp || fscanf(fp, "%d", &node) != 1)
        node = -1;
    if (fp)
        fclose(fp);
    if (node < 0)
        return out;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    fp = fopen(path, "r");
    if (!fp)
        return out;
    char list[4096] = "";
    if (!fgets(list, sizeof(list), fp))
        list[0] = 0;
    fclose(fp);
    // "0-63,128-191" -> set, intersected with what this process may use
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0)
        return out;
    int count = 0;
    for (const char *p = list; *p;) {
        char *end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p)
            break;  // not a number: end of the list
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            if (end == p + 1)
                break;
            p = end;
        }
        for (long c = std::max<long>(a, 0); c <= b && c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &allowed)) {
                CPU_SET(c, &out.set);
                count++;
            }
        while (*p == ',' || *p == ' ' || *p == '\n')
            p++;
    }
    out.valid = count > 0 && count < CPU_COUNT(&allowed);  // nothing to gain when the node is all we have
    return out;
}

CpuSet numa_cpus_of_device(int device)
{
    static std::mutex m;
    static std::vector<std::pair<int, CpuSet>> cache;
    std::lock_guard<std::mutex> g(m);
    for (const auto &e : cache)
        if (e.first == device)
            return e.second;
    cache.emplace_back(device, numa_cpus_of_device_uncached(device));
    return cache.back().second;
}

// Fraction of the file that is in the page cache, from mincore() on 64 windows of 1 MiB spread over it.
double page_cache_fraction(int fd, uint64_t size)
{
    if (size == 0)
        return 1.0;
    const uint64_t win = std::min<uint64_t>(size, 1u << 20), nwin = std::min<uint64_t>(64, (size + win - 1) / win);
    uint64_t seen = 0, resident = 0;
    std::vector<unsigned char> vec((win + 4095) / 4096);
    for (uint64_t k = 0; k < nwin; k++) {
        const uint64_t off = nwin > 1 ? (size - win) / (nwin - 1) * k / 4096 * 4096 : 0;
        const uint64_t len = std::min<uint64_t>(win, size - off);
        void *m = mmap(nullptr, len, PROT_READ, MAP_SHARED, fd, (off_t)off);
        if (m == MAP_FAILED)
            return 1.0;
        const uint64_t pages = (len + 4095) / 4096;
        if (mincore(m, len, vec.data()) == 0) {
            seen += pages;
            for (uint64_t p = 0; p < pages; p++)
                resident += vec[p] & 1;
        }
        munmap(m, len);
    }
    return seen ? (double)resident / (double)seen : 1.0;
}

void close_file_src(FileSrc *fs)
{
    if (fs->fd >= 0)
        close(fs->fd);
    if (fs->fd_direct >= 0)
        close(fs->fd_direct);
    fs->fd = fs->fd_direct = -1;
}

// What the reference pairs a trailing lone float with (papr.c:102-103): the
// float left in the same slot of its static 16384-float buffer by the previous
// chunk (zero when there was none), with its low bytes overwritten by the
// file's 1-3 stray tail bytes (glibc fread copies a partial element).
int open_file_src(papr_hip_ctx *ctx, const char *path, FileSrc *fs)
{
    fs->fd = open(path, O_RDONLY);
    if (fs->fd < 0)
        return fail(ctx, PAPR_E_IO, "cannot open %s", path);
    struct stat sb;
    if (fstat(fs->fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {
        close(fs->fd);
        fs->fd = -1;
        return fail(ctx, PAPR_E_IO, "cannot stat %s (or not a regular file)", path);
    }
    fs->size = (uint64_t)sb.st_size;
    // O_DIRECT pays off for files that are NOT in the page cache (measured 1.7x on the test box's disk) and
    // costs 2x for files that are: PAPR_O_DIRECT=0/1 forces, otherwise decide from a residency sample
    const int direct_mode = env_int("PAPR_O_DIRECT", -1);
    const bool want_direct =
        direct_mode > 0 || (direct_mode < 0 && fs->size >= (64u << 20) && page_cache_fraction(fs->fd, fs->size) < 0.5);
    fs->fd_direct = want_direct ? open(path, O_RDONLY | O_DIRECT) : -1;  // EINVAL on tmpfs: stays -1
    fs->nfloats = fs->size / 4;
    fs->odd = (fs->nfloats & 1u) != 0;
    fs->nsamples = (fs->nfloats + 1) / 2;
    fs->partner = 0.0f;
    if (fs->odd) {
        const uint64_t chunk = 16384;  // papr.c:30
        const uint64_t nfull = fs->nfloats / chunk, rem = fs->nfloats % chunk;
        unsigned char bytes[4] = {0, 0, 0, 0};
        if (nfull >= 1) {
            const uint64_t fidx = (nfull - 1) * chunk + rem;
            if (pread(fs->fd, bytes, 4, (off_t)(fidx * 4)) != 4) {
                close_file_src(fs);
                return fail(ctx, PAPR_E_IO, "short read in %s", path);
            }
        }
        const uint64_t stray = fs->size % 4;
        if (stray && pread(fs->fd, bytes, stray, (off_t)(fs->nfloats * 4)) != (ssize_t)stray) {
            close_file_src(fs);
            return fail(ctx, PAPR_E_IO, "short read in %s", path);
        }
        memcpy(&fs->partner, bytes, 4);
    }
    return PAPR_OK;
}

// read logical samples [s0, s0 + cnt) into dst (8 bytes each)
int read_samples(const FileSrc &fs, uint64_t s0, uint64_t cnt, unsigned char *dst)
{
    const uint64_t byte0 = s0 * 8, file_bytes = fs.nfloats * 4;
    uint64_t want = cnt * 8;
    if (byte0 + want > file_bytes)
        want = file_bytes > byte0 ? file_bytes - byte0 : 0;
    uint64_t done = 0;
    // O_DIRECT (cold files: the device DMAs into the pinned buffer, no page-cache copy) needs 4 KiB-aligned
    // offset, address and length: the whole 4 KiB blocks of the request go through the direct descriptor, never a
    // byte more than was asked for (nothing is written past dst + want); the sub-block remainder, an unaligned
    // request, a short count at the end of the file or an error fall through to the buffered descriptor.
    if (fs.fd_direct >= 0 && (byte0 & 4095) == 0 && ((uintptr_t)dst & 4095) == 0) {
        const uint64_t direct_want = want & ~4095ull;
        while (done < direct_want) {
            const uint64_t ask = std::min<uint64_t>(direct_want - done, (uint64_t)1 << 30);
            const ssize_t got = pread(fs.fd_direct, dst + done, ask, (off_t)(byte0 + done));
            if (got <= 0)
                break;
            done += (uint64_t)got;
            if ((got & 4095) != 0)
                break;  // the unaligned end of the file
        }
    }
    while (done < want) {
        ssize_t got = pread(fs.fd, dst + done, want - done, (off_t)(byte0 + done));
        if (got <= 0)
            return PAPR_E_IO;
        done += (uint64_t)got;
    }
    if (fs.odd && s0 + cnt == fs.nsamples && cnt > 0)
        memcpy(dst + cnt * 8 - 4, &fs.partner, 4);
    return PAPR_OK;
}

// The pinned staging buffers of the ingest (and the pipeline's depth).  Touches nothing else of the context: papr_hip_open_ex
// runs it beside the creation of the streams when the caller says a file load follows.
hipError_t make_staging(papr_hip_ctx *ctx)
{
    if (!ctx->stage_bytes) {
        size_t mb = (size_t)std::max(1, env_int("PAPR_CHUNK_MB", 16));
        ctx->stage_bytes = (mb << 20) / (kChunkAlign * 8) * (kChunkAlign * 8);
        if (!ctx->stage_bytes)
            ctx->stage_bytes = kChunkAlign * 8;
    }
    if (!ctx->num_buf) {
        // the depth of the pipeline: num_buf pinned buffers, read_ahead of them being filled by the readers while the
        // others' copies are queued on the engine
        ctx->num_buf = std::min(kMaxBuf, std::max(2, env_int("PAPR_STAGE_BUFS", kNumBufDefault)));
        ctx->read_ahead = std::min(ctx->num_buf - 1, std::max(1, env_int("PAPR_READ_AHEAD", std::max(kReadAheadDefault, ctx->num_buf / 2))));
    }
    const CpuSet near_gpu = numa_cpus_of_device(ctx->device);
    // pinned pages are placed where they are first touched: do that on the GPU's NUMA node.  (Pinning 16 MiB takes ~3 ms
    // and the driver takes the buffers one at a time: four threads making one each were no faster.)
    cpu_set_t before;
    const bool moved = near_gpu.valid && sched_getaffinity(0, sizeof(before), &before) == 0 &&
                       sched_setaffinity(0, sizeof(near_gpu.set), &near_gpu.set) == 0;
    hipError_t e = hipSuccess;
    for (int b = 0; b < ctx->num_buf && e == hipSuccess; b++) {
        if (ctx->h_stage[b])
            continue;
        e = hipHostMalloc(&ctx->h_stage[b], ctx->stage_bytes, hipHostMallocDefault);
        if (e == hipSuccess && moved)
            memset(ctx->h_stage[b], 0, ctx->stage_bytes);
        if (e == hipSuccess && hipHostGetDevicePointer(&ctx->h_stage_dev[b], ctx->h_stage[b], 0) != hipSuccess) {
            (void)hipGetLastError();
            ctx->h_stage_dev[b] = nullptr;  // (not mapped: this buffer goes through the copy engine)
        }
    }
    if (moved)
        (void)sched_setaffinity(0, sizeof(before), &before);
    return e;
}

int ensure_ingest(papr_hip_ctx *ctx, bool need_device_stage)
{
    HIPCHK(ctx, make_staging(ctx));
    const CpuSet near_gpu = numa_cpus_of_device(ctx->device);
    for (int b = 0; b < ctx->num_buf; b++) {
        if (!ctx->ev_copy[b])
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_copy[b], hipEventDisableTiming));
        if (!ctx->ev_kernel[b])
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_kernel[b], hipEventDisableTiming));
        if (need_device_stage && !ctx->d_stage[b])
            HIPCHK(ctx, hipMalloc(&ctx->d_stage[b], ctx->stage_bytes + PAPR_TILE_SAMPLES_MAX * 8));
    }
    if (need_device_stage && !ctx->d_tail)
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_tail, PAPR_TILE_SAMPLES_MAX * 8));
    ctx->trace.mark("ei_bufs");
    if (!ctx->pool) {
        int n = env_int("PAPR_READ_THREADS", 0);
        if (n <= 0)
            n = (int)std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
        ctx->reader_threads = n;
        ctx->pool = new ReaderPool(n, near_gpu);
        ctx->ingest_numa = near_gpu.valid;
    }
    ctx->trace.mark("ei_pool");
    return PAPR_OK;
}

// Walk file samples [first, first + n) in pinned-buffer-sized chunks: parallel
// pread into a pinned buffer, hipMemcpyAsync on the copy stream, then the pass
// kernel on the compute stream as soon as that chunk has landed.  Three buffers
// keep disk/page-cache reads, PCIe copies and kernels overlapped.
int stream_file(papr_hip_ctx *ctx, StreamPass pass, const CcdfPlan *plan, size_t *nrecords_out)
{
    FileSrc fs;
    int rc = open_file_src(ctx, ctx->path.c_str(), &fs);
    if (rc)
        return rc;
    ctx->ingest.file_passes++;
    const bool to_resident = (pass == PASS_LOAD_STATS);
    const bool timed = (pass == PASS_LOAD_STATS || pass == PASS_STREAM_STATS);
    double t_mark = now_s();
    ctx->trace.mark("sf_opened");
    rc = ensure_ingest(ctx, !to_resident);
    if (rc) {
        close_file_src(&fs);
        return rc;
    }
    const uint64_t chunk_samples = ctx->stage_bytes / 8;
    const uint64_t nchunks = (ctx->n + chunk_samples - 1) / chunk_samples;
    size_t records = 0;
    if (pass == PASS_LOAD_STATS || pass == PASS_STREAM_STATS) {
        const int per_chunk = ctx->ingest_run ? ctx->ingest_run->blocks : blocks_of(ctx, PASS1);
        rc = ensure_partials(ctx, (size_t)nchunks * per_chunk + 1);
        if (rc) {
            close_file_src(&fs);
            return rc;
        }
    }
    if (timed) {
        ctx->ingest.setup_s += now_s() - t_mark;
        ctx->ingest.chunks = nchunks;
        ctx->ingest.reader_threads = ctx->reader_threads;
        ctx->ingest.o_direct = fs.fd_direct >= 0;
        ctx->ingest.numa_bound = ctx->ingest_numa ? 1 : 0;
    }
    // A file read with O_DIRECT goes through one io_uring (papr_uring.h) where the kernel offers it: the reads need no
    // CPU, only enough of them in flight.  PAPR_IO_URING=0 keeps the reader threads.
    if (fs.fd_direct >= 0 && !ctx->uring_tried) {
        ctx->uring_tried = true;
        if (env_int("PAPR_IO_URING", 1) != 0)
            ctx->uring = UringReader::create(256);
    }
    if (ctx->uring && ctx->uring->poisoned())
        return fail(ctx, PAPR_E_IO, "the io_uring reader was given up with reads the kernel did not give back: its staging buffers are not safe to reuse");
    UringReader *ring = fs.fd_direct >= 0 && ctx->uring && !ctx->uring->dead() ? ctx->uring : nullptr;  // (a ring that failed once: threads)
    if (timed)
        ctx->ingest.io_uring = ring ? 1 : 0;
    // queue the slices of chunk c for the reader threads (buffer c % num_buf must be free)
    std::vector<ReadBatch> batches(nchunks);
    const FileSrc *fsp = &fs;
    const uint64_t file_first = ctx->file_first, shard_n = ctx->n;
    auto submit_chunk = [&](uint64_t c) {
        const int b = (int)(c % ctx->num_buf);
        const uint64_t s0 = c * chunk_samples;
        const uint64_t cnt = std::min(chunk_samples, shard_n - s0);
        unsigned char *hbuf = (unsigned char *)ctx->h_stage[b];
        if (ring) {
            // the whole 4 KiB blocks of the chunk through the direct descriptor in 1 MiB requests, the sub-block rest
            // (only the end of the file has one) through the buffered one; never a byte more than was asked for
            const uint64_t byte0 = (file_first + s0) * 8, file_bytes = fs.nfloats * 4;
            const uint64_t want = byte0 < file_bytes ? std::min<uint64_t>(cnt * 8, file_bytes - byte0) : 0;
            const bool aligned = (byte0 & 4095) == 0 && ((uintptr_t)hbuf & 4095) == 0;
            const uint64_t direct = aligned ? (want & ~4095ull) : 0;
            if (direct)
                ring->submit(&batches[c], fs.fd_direct, fs.fd, byte0, direct, hbuf, (uint64_t)1 << 20);
            if (want > direct)
                ring->submit(&batches[c], fs.fd, fs.fd, byte0 + direct, want - direct, hbuf + direct, (uint64_t)1 << 20);
            return;
        }
        const int nthr = ctx->reader_threads;
        const uint64_t per = ((cnt + nthr - 1) / nthr + 511) & ~511ull;
        for (int t = 0; t < nthr; t++) {
            const uint64_t a = std::min<uint64_t>((uint64_t)t * per, cnt), e = std::min<uint64_t>(a + per, cnt);
            if (e > a)
                ctx->pool->submit(&batches[c], [fsp, file_first, s0, a, e, hbuf] {
                    return read_samples(*fsp, file_first + s0 + a, e - a, hbuf + a * 8);
                });
        }
    };
    bool two_copy_streams = env_int("PAPR_COPY_STREAMS", 1) >= 2;
    if (two_copy_streams && !ctx->copy_stream2 &&
        hipStreamCreateWithFlags(&ctx->copy_stream2, hipStreamNonBlocking) != hipSuccess) {
        ctx->copy_stream2 = nullptr;
        two_copy_streams = false;  // (one stream it is)
    }
    // copy chunk c (already read into its pinned buffer) to the device and run the pass kernel on it
    auto process_chunk = [&](uint64_t c) -> int {
        const int b = (int)(c % ctx->num_buf);
        const uint64_t s0 = c * chunk_samples;
        const uint64_t cnt = std::min(chunk_samples, ctx->n - s0);
        unsigned char *hbuf = (unsigned char *)ctx->h_stage[b];
        float *dst = to_resident ? ctx->d_iq + 2 * s0 : (float *)ctx->d_stage[b];
        hipStream_t cs = two_copy_streams && (c & 1) ? ctx->copy_stream2 : ctx->copy_stream;
        if (!to_resident && c >= (uint64_t)ctx->num_buf)
            HIPCHK(ctx, hipStreamWaitEvent(cs, ctx->ev_kernel[b], 0));
        // the H2D leg: a kernel pulls the staging buffer over the link (PAPR_H2D=copy: the copy engine does)
        if (ctx->h2d_pull < 0) {
            const char *form = getenv("PAPR_H2D");
            ctx->h2d_pull = form && !strcmp(form, "copy") ? 0 : 1;
        }
        if (ctx->h2d_pull && ctx->h_stage_dev[b] && (cnt * 8) % 8 == 0 && ((uintptr_t)dst & 15) == 0) {
            papr_launch_pull(cs, ctx->h_stage_dev[b], dst, cnt * 8);
            HIPCHK(ctx, hipGetLastError());
        } else {
            HIPCHK(ctx, hipMemcpyAsync(dst, hbuf, cnt * 8, hipMemcpyHostToDevice, cs));
        }
        if (c == 0)
            ctx->trace.mark("c0_copy_queued");
        HIPCHK(ctx, hipEventRecord(ctx->ev_copy[b], cs));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_copy[b], 0));
        if (c == 0)
            ctx->trace.mark("c0_events");
        const bool last = (c + 1 == nchunks);
        int prc = PAPR_OK;
        switch (pass) {
        case PASS_LOAD_STATS:
        case PASS_STREAM_STATS: {
            int nrec = 0;
            if (ctx->ingest_run)  // one-sweep ingest: pass 1 + banded pass 2 on the chunk
                prc = sweep_launch(ctx, *ctx->ingest_run, dst, cnt, ctx->base + s0, records, &nrec);
            else
                prc = launch_stats_range(ctx, dst, cnt, ctx->base + s0, records, &nrec);
            records += (size_t)nrec;
            if (prc == PAPR_OK && last && pass == PASS_STREAM_STATS) {
                const uint64_t tile = ctx->ingest_run ? ctx->ingest_run->tile : tile_samples(ctx, PASS1);
                const uint64_t full = cnt / tile * tile;
                if (cnt > full)
                    HIPCHK(ctx, hipMemcpyAsync(ctx->d_tail, dst + 2 * full, (cnt - full) * 8, hipMemcpyDeviceToDevice,
                                               ctx->stream));
            }
            break;
        }
        case PASS_STREAM_CCDF:
            prc = launch_ccdf_range(ctx, *plan, dst, cnt);
            break;
        case PASS_STREAM_CCDF_EXACT:
            prc = launch_fused_chunk(ctx, *plan, dst, s0, cnt, last);
            break;
        case PASS_STREAM_NAN:
            papr_launch_first_nan(ctx->stream, 1024, dst, cnt, ctx->base + s0, ctx->d_nan_key);
            HIPCHK(ctx, hipGetLastError());
            break;
        }
        if (prc)
            return prc;
        HIPCHK(ctx, hipEventRecord(ctx->ev_kernel[b], ctx->stream));
        return PAPR_OK;
    };

    uint64_t submitted = 0;
    ctx->trace.mark("sf_loop");
    for (; submitted < std::min<uint64_t>((uint64_t)ctx->read_ahead, nchunks); submitted++)
        submit_chunk(submitted);
    for (uint64_t c = 0; c < nchunks && rc == PAPR_OK; c++) {
        t_mark = now_s();
        if (ring) {
            if (ring->wait(&batches[c]))
                rc = fail(ctx, PAPR_E_IO, "read error in %s (io_uring: errno %d)", ctx->path.c_str(), batches[c].error);
            // (the reader threads patch the phantom sample's partner in read_samples: here it is done once per chunk)
            const uint64_t s0 = c * chunk_samples, cnt = std::min(chunk_samples, shard_n - s0);
            if (!rc && fs.odd && file_first + s0 + cnt == fs.nsamples && cnt > 0)
                memcpy((unsigned char *)ctx->h_stage[c % ctx->num_buf] + cnt * 8 - 4, &fs.partner, 4);
        } else if (ctx->pool->wait(&batches[c])) {
            rc = fail(ctx, PAPR_E_IO, "read error in %s", ctx->path.c_str());
        }
        if (timed)
            ctx->ingest.read_s += now_s() - t_mark;
        if (rc)
            break;
        if (c == 0)
            ctx->trace.mark("sf_chunk0_read");
        t_mark = now_s();
        rc = process_chunk(c);
        if (timed)
            ctx->ingest.issue_s += now_s() - t_mark;
        if (c < 2)
            ctx->trace.mark(c ? "sf_chunk1_issued" : "sf_chunk0_issued");
        // read ahead: the next unread chunk goes into the buffer used num_buf chunks earlier, which is
        // free once that chunk's H2D copy has completed
        if (rc == PAPR_OK && submitted < nchunks) {
            t_mark = now_s();
            if (submitted >= (uint64_t)ctx->num_buf &&
                hipEventSynchronize(ctx->ev_copy[submitted % ctx->num_buf]) != hipSuccess)
                rc = fail(ctx, PAPR_E_HIP, "hipEventSynchronize failed while recycling a staging buffer");
            if (timed)
                ctx->ingest.buffer_wait_s += now_s() - t_mark;
            if (rc == PAPR_OK)
                submit_chunk(submitted++);
        }
    }
    ctx->trace.mark("sf_loop_end");
    // on any failure let the reads already queued finish before `fs` and the batches go away
    for (uint64_t k = 0; k < submitted; k++) {
        if (ring)
            (void)ring->wait(&batches[k]);
        else
            (void)ctx->pool->wait(&batches[k]);
    }
    close_file_src(&fs);
    if (nrecords_out)
        *nrecords_out = records;
    return rc;
}

// papr_hip_load_file, optionally as a one-sweep ingest (guess != nullptr): the per-chunk kernel then also bins
// against the guessed bands and stashes, so that papr_hip_ccdf needs no second read of the shard — or of the file
int load_file_impl(papr_hip_ctx *ctx, const char *path, uint64_t first_sample, uint64_t nsamples, const float *guess,
                   int nguess)
{
    if (!ctx || !path)
        return PAPR_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const double t_begin = now_s();
    ctx->trace.mark("load_enter");
    memset(&ctx->ingest, 0, sizeof(ctx->ingest));
    papr_hip_sweep_info &info = ctx->sweep_info;
    info.swept = info.resolved = 0;
    info.stash_samples = 0;
    info.reason = PAPR_SWEEP_NONE;
    ctx->ingest_run = nullptr;
    FileSrc fs;
    int rc = open_file_src(ctx, path, &fs);
    if (rc)
        return rc;
    close_file_src(&fs);
    if (first_sample > fs.nsamples)
        return fail(ctx, PAPR_E_ARG, "first_sample %llu is past the end of %s (%llu samples)",
                    (unsigned long long)first_sample, path, (unsigned long long)fs.nsamples);
    if (nsamples > fs.nsamples - first_sample)  // (also UINT64_MAX = "to the end"; written so that it cannot wrap)
        nsamples = fs.nsamples - first_sample;

    const bool fits = (nsamples + PAPR_TILE_SAMPLES_MAX) * 8 <= ctx->hbm_budget;
    if (!ctx->owns_iq || !fits)
        release_shard(ctx);
    if (fits) {
        rc = ensure_owned_capacity(ctx, nsamples);
        if (rc)
            return rc;
    }
    ctx->trace.mark("shard_alloc");
    ctx->path = path;
    ctx->file_first = first_sample;
    ctx->n = nsamples;
    ctx->base = first_sample;
    ctx->resident = fits;
    ctx->loaded = true;
    ctx->have_file_stats = false;
    ctx->exact_valid = false;
    ctx->peer_epoch += 0x9E3779B97F4A7C15ull;  // (the ranks agree again on the single-wait step: stats_sweep_fused)
    ctx->sweep_valid = false;
    ctx->est_groups_valid = ctx->exact_swept = ctx->exact_program_launched = false;
    ctx->shard_flags = (fs.odd && first_sample + nsamples == fs.nsamples && nsamples > 0) ? PAPR_FLAG_ODD_TAIL : 0;

    SweepRun run;
    if (guess) {
        int reason = PAPR_SWEEP_MODE;
        // exact-sum mode rides along too when papr_hip_estimate_file left the per-tile sums of THIS range behind
        // (the sweep then also builds every segment's rounding-function pair for a speculated binade, section 5)
        const bool exact_ok = ctx->exact && ctx->est_file_valid && ctx->est_file_first == first_sample &&
                              ctx->est_file_n == nsamples;
        if ((!ctx->exact || exact_ok) && nsamples) {
            ctx->est_groups_valid = exact_ok;
            rc = ensure_ingest(ctx, !fits);  // fixes the chunk size
            if (rc == PAPR_OK)
                rc = sweep_prepare(ctx, guess, nguess, nsamples, ctx->stage_bytes / 8, &run, &reason);
            if (rc == PAPR_OK && reason == PAPR_SWEEP_OK && run.exact) {
                rc = ensure_exact_buffers(ctx);
                if (rc == PAPR_OK) {
                    papr_launch_exact_spec(ctx->stream, ctx->d_est_groups, ctx->est_ngroups, (uint32_t)ctx->est_ratio,
                                           (double)ctx->est_ratio, ctx->exact_before_hint,
                                           ctx->d_est_groups + 4 * ctx->est_groups_cap, nsamples / PAPR_EXACT_TILE_SAMPLES,
                                           ctx->d_tile_E_spec);
                    if (hipGetLastError() != hipSuccess)
                        rc = fail(ctx, PAPR_E_HIP, "the binade speculation failed to launch");
                }
            }
            if (rc) {
                ctx->loaded = false;
                return rc;
            }
        }
        info.reason = reason;
        if (reason == PAPR_SWEEP_OK)
            ctx->ingest_run = &run;
    }
    ctx->ingest.setup_s = now_s() - t_begin;
    ctx->ingest.bytes = nsamples * 8;
    ctx->ingest.resident = fits ? 1 : 0;
    // pass 1 (or the whole sweep) rides along with the ingest
    size_t records = 0;
    rc = stream_file(ctx, fits ? PASS_LOAD_STATS : PASS_STREAM_STATS, nullptr, &records);
    ctx->trace.mark("streamed");
    const bool swept = ctx->ingest_run != nullptr;
    ctx->ingest_run = nullptr;
    if (rc) {
        ctx->loaded = false;
        return rc;
    }
    const uint64_t chunk_samples = ctx->stage_bytes / 8;
    const uint64_t last_cnt = nsamples ? nsamples - (nsamples - 1) / chunk_samples * chunk_samples : 0;
    const uint32_t tail = (uint32_t)(last_cnt % (swept ? run.tile : tile_samples(ctx, PASS1)));
    if (swept) {
        rc = sweep_fetch(ctx, run);
        if (rc) {
            ctx->loaded = false;
            return rc;
        }
    }
    const float *tail_ptr = fits ? ctx->d_iq + 2 * (nsamples - tail) : ctx->d_tail;
    papr_stats st;
    const double t_drain = now_s();
    rc = finish_stats(ctx, records, tail_ptr, tail, ctx->base + nsamples - tail, &st);
    if (rc) {
        ctx->loaded = false;
        return rc;
    }
    ctx->ingest.drain_s = now_s() - t_drain;
    if (swept && std::isnan(st.sum)) {
        // NaN in the data: the sweep's integer-max trackers do not apply — take the file in again the plain way
        rc = load_file_impl(ctx, path, first_sample, nsamples, nullptr, 0);
        info.reason = PAPR_SWEEP_NO_BANDS;
        return rc;
    }
    if (swept) {
        rc = sweep_collect(ctx, run);
        if (rc) {
            ctx->loaded = false;
            return rc;
        }
        if (run.exact) {  // d_seg_D holds every segment's sum and its pair for the speculated binade
            ctx->exact_swept = true;
            ctx->exact_valid = true;
        }
    }
    if (std::isnan(st.sum)) {
        unsigned long long key = ~0ull;
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_nan_key, &key, 8, hipMemcpyHostToDevice, ctx->stream));
        if (fits) {
            papr_launch_first_nan(ctx->stream, 1024, ctx->d_iq, ctx->n, ctx->base, ctx->d_nan_key);
            HIPCHK(ctx, hipGetLastError());
        } else {
            rc = stream_file(ctx, PASS_STREAM_NAN, nullptr, nullptr);
            if (rc)
                return rc;
        }
        HIPCHK(ctx, hipMemcpyAsync(&key, ctx->d_nan_key, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        apply_nan_key(&st, key);
    }
    ctx->file_stats = st;
    ctx->have_file_stats = true;
    ctx->exact_valid = ctx->exact;  // the per-tile sums are on the device, resident shard or not
    ctx->ingest.total_s = now_s() - t_begin;
    ctx->trace.mark("load_leave");
    ctx->trace.dump();
    return PAPR_OK;
}

}  // namespace papr_rt

extern "C" {

static int papr_hip_load_file_impl(papr_hip_ctx *ctx, const char *path, uint64_t first_sample, uint64_t nsamples)
{
    return load_file_impl(ctx, path, first_sample, nsamples, nullptr, 0);
}

int papr_hip_shard_fits(const papr_hip_ctx *ctx, uint64_t nsamples)
{
    if (!ctx)
        return PAPR_E_ARG;
    return (nsamples + PAPR_TILE_SAMPLES_MAX) * 8 <= ctx->hbm_budget ? 1 : 0;
}

static int papr_hip_load_file_sweep_impl(papr_hip_ctx *ctx, const char *path, uint64_t first_sample, uint64_t nsamples,
                             const float *guess_levels, int nlevels)
{
    if (nlevels < 0 || (nlevels && !guess_levels))
        return PAPR_E_ARG;
    static const float none = 0.0f;
    return load_file_impl(ctx, path, first_sample, nsamples, guess_levels ? guess_levels : &none, nlevels);
}

// Mean estimate of a file range without loading it: the same 1-in-`ratio` tile sample as papr_hip_estimate, read
// by the ingest's reader threads into the staging buffers and summed on the device.
static int papr_hip_estimate_file_impl(papr_hip_ctx *ctx, const char *path, uint64_t first_sample, uint64_t nsamples, papr_stats *est)
{
    if (!ctx || !path || !est)
        return PAPR_E_ARG;
    papr_stats_init(est);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    FileSrc fs;
    int rc = open_file_src(ctx, path, &fs);
    if (rc)
        return rc;
    if (first_sample > fs.nsamples) {
        close_file_src(&fs);
        return fail(ctx, PAPR_E_ARG, "first_sample %llu is past the end of %s (%llu samples)",
                    (unsigned long long)first_sample, path, (unsigned long long)fs.nsamples);
    }
    if (nsamples > fs.nsamples - first_sample)
        nsamples = fs.nsamples - first_sample;
    ctx->sweep_info.estimate_samples = 0;
    const uint64_t ntiles = nsamples / PAPR_ESTIMATE_TILE_SAMPLES;
    if (ntiles == 0) {
        close_file_src(&fs);
        return PAPR_OK;  // n = 0: no estimate
    }
    rc = ensure_ingest(ctx, true);
    if (rc) {
        close_file_src(&fs);
        return rc;
    }
    uint64_t ratio = ctx->tune.estimate_ratio > 0 ? (uint64_t)ctx->tune.estimate_ratio : (uint64_t)kEstimateRatio;
    ratio = std::max<uint64_t>(1, std::min<uint64_t>(ratio, ntiles / kEstimateMinTiles));
    const uint64_t ngroups = ntiles / ratio;
    constexpr uint64_t kTileBytes = (uint64_t)PAPR_ESTIMATE_TILE_SAMPLES * 8;
    const uint64_t per_batch = ctx->stage_bytes / kTileBytes;
    const uint64_t nbatches = (ngroups + per_batch - 1) / per_batch;
    const int blocks_max = (int)std::min<uint64_t>(per_batch, (uint64_t)ctx->num_cus * 8);
    rc = ensure_partials(ctx, (size_t)nbatches * blocks_max + 1);
    // per-workgroup sums of squared piece sums, for the estimate's standard error (as papr_hip_estimate)
    const size_t sq_count = (size_t)nbatches * (size_t)blocks_max;
    double *d_sq = nullptr;
    std::vector<double> h_sq;
    if (rc == PAPR_OK) {
        try {
            h_sq.resize(sq_count);
        } catch (...) {
            rc = fail(ctx, PAPR_E_NOMEM, "out of host memory");
        }
        if (rc == PAPR_OK && hipMalloc((void **)&d_sq, sq_count * sizeof(double)) != hipSuccess)
            rc = fail(ctx, PAPR_E_NOMEM, "hipMalloc for the estimate failed");
    }
    // exact-sum mode: keep the per-tile sampled sums on the device (as papr_hip_estimate does) — a one-sweep ingest of
    // the same range speculates every tile's running-sum binade from them
    ctx->est_file_valid = false;
    double *group_sums = nullptr;
    if (rc == PAPR_OK && ctx->exact) {
        if (ctx->est_groups_cap < ngroups) {
            if (ctx->d_est_groups) (void)hipFree(ctx->d_est_groups);
            ctx->d_est_groups = nullptr;
            ctx->est_groups_cap = 0;
            const uint64_t cap = std::max<uint64_t>(ngroups, 4096);
            if (hipMalloc((void **)&ctx->d_est_groups, cap * 5 * sizeof(double)) != hipSuccess)
                rc = fail(ctx, PAPR_E_NOMEM, "hipMalloc for the estimate failed");
            else
                ctx->est_groups_cap = cap;
        }
        group_sums = ctx->d_est_groups;
    }
    std::vector<ReadBatch> batches(nbatches);
    const FileSrc *fsp = &fs;
    auto submit = [&](uint64_t bi) {
        const uint64_t g0 = bi * per_batch, g1 = std::min(ngroups, g0 + per_batch);
        unsigned char *hbuf = (unsigned char *)ctx->h_stage[bi % ctx->num_buf];
        const int nthr = ctx->reader_threads;
        const uint64_t per = (g1 - g0 + nthr - 1) / nthr;
        for (int t = 0; t < nthr; t++) {
            const uint64_t a = std::min(g1, g0 + (uint64_t)t * per), e = std::min(g1, a + per);
            if (e > a)
                ctx->pool->submit(&batches[bi], [fsp, first_sample, ratio, g0, a, e, hbuf] {
                    for (uint64_t g = a; g < e; g++) {
                        // one tile of group g, picked by a hash of g (no aliasing with periodic structure in the capture)
                        const uint64_t tile = g * ratio + ((g + 1) * 0x9E3779B97F4A7C15ull >> 40) % ratio;
                        const int r = read_samples(*fsp, first_sample + tile * PAPR_ESTIMATE_TILE_SAMPLES,
                                                   PAPR_ESTIMATE_TILE_SAMPLES, hbuf + (g - g0) * kTileBytes);
                        if (r)
                            return r;
                    }
                    return (int)PAPR_OK;
                });
        }
    };
    size_t records = 0;
    uint64_t submitted = 0;
    for (; rc == PAPR_OK && submitted < std::min<uint64_t>(2, nbatches); submitted++)
        submit(submitted);
    for (uint64_t bi = 0; bi < nbatches && rc == PAPR_OK; bi++) {
        if (ctx->pool->wait(&batches[bi])) {
            rc = fail(ctx, PAPR_E_IO, "read error in %s", path);
            break;
        }
        const int b = (int)(bi % ctx->num_buf);
        const uint64_t cnt = std::min(ngroups, (bi + 1) * per_batch) - bi * per_batch;
        if (hipMemcpyAsync(ctx->d_stage[b], ctx->h_stage[b], cnt * kTileBytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            rc = fail(ctx, PAPR_E_HIP, "hipMemcpyAsync of the estimate sample failed");
            break;
        }
        const int blocks = (int)std::min<uint64_t>(cnt, (uint64_t)blocks_max);
        time_begin(ctx, 4, cnt * kTileBytes);
        papr_launch_estimate(ctx->stream, blocks, ctx->d_stage[b], cnt, 1, ctx->d_partials + records,
                             group_sums ? group_sums + 4 * bi * per_batch : nullptr, d_sq + records);
        time_end(ctx);
        records += (size_t)blocks;
        if (hipEventRecord(ctx->ev_copy[b], ctx->stream) != hipSuccess)
            rc = fail(ctx, PAPR_E_HIP, "hipEventRecord failed");
        if (rc == PAPR_OK && submitted < nbatches) {
            // the buffer about to be refilled was consumed num_buf batches ago
            if (submitted >= (uint64_t)ctx->num_buf && hipEventSynchronize(ctx->ev_copy[submitted % ctx->num_buf]) != hipSuccess)
                rc = fail(ctx, PAPR_E_HIP, "hipEventSynchronize failed while recycling a staging buffer");
            if (rc == PAPR_OK)
                submit(submitted++);
        }
    }
    for (uint64_t k = 0; k < submitted; k++)
        (void)ctx->pool->wait(&batches[k]);
    close_file_src(&fs);
    if (rc == PAPR_OK) {
        papr_launch_stats_finalize(ctx->stream, nullptr, 0, 0, ctx->d_partials, (uint32_t)records, ctx->h_result_dev);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(h_sq.data(), d_sq, records * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = fail(ctx, PAPR_E_HIP, "the estimate's reduction failed");
    }
    if (d_sq)
        (void)hipFree(d_sq);
    if (rc)
        return rc;
    // as papr_hip_estimate: the record describes the whole range (sampled sum scaled to all of its samples), and
    // est->peak carries the estimate's relative standard error
    const uint64_t sampled = ngroups * PAPR_ESTIMATE_TILE_SAMPLES;
    double sq = 0.0;
    for (size_t k = 0; k < records; k++)
        sq += h_sq[k];
    const double S = ctx->h_result->sum, P = 4.0 * (double)ngroups;
    const double var_total = P > 1.0 ? P / (P - 1.0) * std::max(0.0, sq - S * S / P) : 0.0;
    est->peak = (S > 0.0 && ratio > 1) ? (float)(std::sqrt(var_total) / S) : 0.0f;
    est->sum = S * ((double)nsamples / (double)sampled);
    est->n = nsamples;
    ctx->sweep_info.estimate_samples = sampled;
    if (group_sums) {
        ctx->est_ngroups = ngroups;
        ctx->est_ratio = ratio;
        ctx->est_file_valid = true;
        ctx->est_file_first = first_sample;
        ctx->est_file_n = nsamples;
    }
    return PAPR_OK;
}


// A pipe holds 64 KiB by default: a reader that empties it 16 times per MiB spends its time being woken.  As much as an
// unprivileged process may ask for (fs.pipe-max-size, 1 MiB by default); not a pipe, or refused: as it was.
static void widen_pipe(int fd)
{
#ifdef F_SETPIPE_SZ
    long want = 1 << 20;
    if (FILE *f = fopen("/proc/sys/fs/pipe-max-size", "r")) {
        long v = 0;
        if (fscanf(f, "%ld", &v) == 1 && v >= 65536)
            want = std::min<long>(v, 16 << 20);
        fclose(f);
    }
    (void)fcntl(fd, F_SETPIPE_SZ, (int)want);
#else
    (void)fd;
#endif
}

// ---- a stream that cannot be positioned (a FIFO, a pipe, a socket) ---------------------------------------------------
// The reference fopen()s whatever it is given (papr.c:62, 93) and reads a pipe like a file, 64 KiB at a time, to its end:
// pass 1 sees every sample.  Its length is only known at the end, so the shard grows as the bytes arrive: ONE reader
// (this thread: read() fills a pinned staging buffer while the previous buffers' copies are on the link), the bytes land
// in HBM where the file's would, and the tail rules (papr.c:102-103: a lone last float is paired with what the static
// buffer still holds in its slot, stray bytes overwrite that float's low bytes) are applied from the bytes in HBM once
// the length is known.  What the reference does NOT get from a pipe is its pass 2 (fseeko fails, EOF stays set:
// papr.c:142-143 / 174-175): that is the caller's to reproduce (bin/papr prints zero counts).
static int papr_hip_load_stream_impl(papr_hip_ctx *ctx, int fd, uint64_t *nsamples_out)
{
    if (!ctx || fd < 0)
        return PAPR_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    memset(&ctx->ingest, 0, sizeof(ctx->ingest));
    const double t_begin = now_s();
    HIPCHK(ctx, make_staging(ctx));
    release_shard(ctx);
    widen_pipe(fd);
    const size_t slack = (size_t)PAPR_TILE_SAMPLES_MAX * 8;
    unsigned char *dev = nullptr;
    size_t cap = 0, total = 0;
    std::vector<hipEvent_t> done((size_t)ctx->num_buf, nullptr);
    auto cleanup = [&](int rc) {
        (void)hipStreamSynchronize(ctx->stream);
        for (hipEvent_t e : done)
            if (e)
                (void)hipEventDestroy(e);
        if (rc != PAPR_OK && dev)
            (void)hipFree(dev);
        return rc;
    };
    auto grow = [&](size_t need) -> int {  // HBM for `need` bytes of stream (+ slack): doubled, what is there moves over
        if (need + slack <= cap)
            return PAPR_OK;
        if (need + slack > ctx->hbm_budget)
            return fail(ctx, PAPR_E_NOMEM, "the stream is longer than the HBM budget (%zu MiB): a stream that cannot be read "
                        "twice has to stay resident", ctx->hbm_budget >> 20);
        size_t want = std::max<size_t>(cap * 2, (size_t)256 << 20);
        while (want < need + slack)
            want *= 2;
        want = std::min(want, ctx->hbm_budget);
        unsigned char *bigger = nullptr;
        if (hipMalloc((void **)&bigger, want) != hipSuccess) {
            (void)hipGetLastError();
            want = need + slack;  // (no room for the doubled size: exactly what is needed now)
            if (hipMalloc((void **)&bigger, want) != hipSuccess) {
                (void)hipGetLastError();
                return fail(ctx, PAPR_E_NOMEM, "hipMalloc(%zu bytes) for the growing shard failed", want);
            }
        }
        if (dev) {
            if (hipMemcpyAsync(bigger, dev, total, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) {
                (void)hipFree(bigger);
                return fail(ctx, PAPR_E_HIP, "moving the growing shard failed");
            }
            (void)hipFree(dev);
        }
        dev = bigger;
        cap = want;
        return PAPR_OK;
    };
    bool eof = false;
    for (uint64_t k = 0; !eof; k++) {
        const int b = (int)(k % (uint64_t)ctx->num_buf);
        if (done[(size_t)b] && hipEventSynchronize(done[(size_t)b]) != hipSuccess)  // (this buffer's last copy has left it)
            return cleanup(fail(ctx, PAPR_E_HIP, "hipEventSynchronize failed"));
        unsigned char *host = (unsigned char *)ctx->h_stage[b];
        size_t got = 0;
        const double t_read = now_s();
        while (got < ctx->stage_bytes) {
            const ssize_t r = read(fd, host + got, ctx->stage_bytes - got);
            if (r < 0 && errno == EINTR)
                continue;
            if (r < 0)
                return cleanup(fail(ctx, PAPR_E_IO, "read from the stream failed: %s", strerror(errno)));
            if (r == 0) {
                eof = true;
                break;
            }
            got += (size_t)r;
        }
        ctx->ingest.read_s += now_s() - t_read;
        if (!got)
            break;
        int rc = grow(total + got);
        if (rc)
            return cleanup(rc);
        if (hipMemcpyAsync(dev + total, host, got, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            return cleanup(fail(ctx, PAPR_E_HIP, "hipMemcpyAsync of a stream chunk failed"));
        if (!done[(size_t)b] && hipEventCreateWithFlags(&done[(size_t)b], hipEventDisableTiming) != hipSuccess)
            return cleanup(fail(ctx, PAPR_E_HIP, "hipEventCreate failed"));
        if (hipEventRecord(done[(size_t)b], ctx->stream) != hipSuccess)
            return cleanup(fail(ctx, PAPR_E_HIP, "hipEventRecord failed"));
        total += got;
        ctx->ingest.chunks++;
    }
    int rc = grow(total);  // (an empty stream still gets its slack: the kernels' lane loads stay in bounds)
    if (rc)
        return cleanup(rc);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess)
        return cleanup(fail(ctx, PAPR_E_HIP, "hipStreamSynchronize failed"));
    // the reference's sample count and tail (as open_file_src derives them for a file of this length)
    const uint64_t nfloats = total / 4, stray = total % 4;
    const bool odd = (nfloats & 1u) != 0;
    const uint64_t nsamples = (nfloats + 1) / 2;
    if (odd) {
        const uint64_t chunk = 16384, nfull = nfloats / chunk, rem = nfloats % chunk;
        unsigned char bytes[4] = {0, 0, 0, 0};
        if ((nfull >= 1 && hipMemcpy(bytes, dev + ((nfull - 1) * chunk + rem) * 4, 4, hipMemcpyDeviceToHost) != hipSuccess) ||
            (stray && hipMemcpy(bytes, dev + nfloats * 4, stray, hipMemcpyDeviceToHost) != hipSuccess) ||
            hipMemcpy(dev + nfloats * 4, bytes, 4, hipMemcpyHostToDevice) != hipSuccess)
            return cleanup(fail(ctx, PAPR_E_HIP, "building the phantom sample failed"));
    }
    cleanup(PAPR_OK);
    ctx->d_iq = (float *)dev;
    ctx->owns_iq = true;
    ctx->cap = (cap - slack) / 8;
    ctx->n = nsamples;
    ctx->base = 0;
    ctx->loaded = ctx->resident = true;
    ctx->shard_flags = odd ? PAPR_FLAG_ODD_TAIL : 0;
    ctx->ingest.resident = 1;
    ctx->ingest.reader_threads = 1;
    ctx->ingest.file_passes = 1;
    ctx->ingest.setup_s = 0;
    ctx->ingest.drain_s = now_s() - t_begin - ctx->ingest.read_s;
    if (nsamples_out)
        *nsamples_out = nsamples;
    return PAPR_OK;
}

// ---- the same stream, of ANY length -----------------------------------------------------------------------------------
// papr.c:100-129 reads a FIFO of any length in 64 KiB of memory; what it keeps is pass 1's few scalars.  So does this: the
// stream crosses ONE window of HBM (at most the budget, 256 MiB by default).  A full window is a resident shard for as long
// as it takes to reduce it — papr_hip_stats, and in exact-sum mode the window's sum program built from the EXACT running
// sum in front of it (known: the windows are reduced in stream order) and replayed at once (papr_exact_chain_continue),
// so the sum stays the reference's accumulator bit for bit — then the next bytes land on the same memory.  The 64 KiB in
// front of the window keep the previous window's last reference chunk: the phantom sample of an odd float count takes its
// partner from 16384 floats before the stream's end (papr.c:102-103 with the static buffer), wherever a window ended.
// Pass 2 over such an input counts nothing in the reference (fseeko fails, EOF stays set: papr.c:142-143 / 174-175), so no
// sample is needed a second time.
static int papr_hip_stream_stats_impl(papr_hip_ctx *ctx, int fd, papr_stats *total_out, int *exact_out, uint64_t *windows_out)
{
    if (!ctx || fd < 0 || !total_out)
        return PAPR_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    memset(&ctx->ingest, 0, sizeof(ctx->ingest));
    const double t_begin = now_s();
    HIPCHK(ctx, make_staging(ctx));
    release_shard(ctx);
    widen_pipe(fd);
    const size_t slack = (size_t)PAPR_TILE_SAMPLES_MAX * 8, keep = 65536, group = (size_t)2 << 20;
    if (ctx->hbm_budget < keep + slack + keep)
        return fail(ctx, PAPR_E_NOMEM, "an HBM budget of %zu bytes holds no window of the stream", ctx->hbm_budget);
    size_t window = std::min<size_t>((size_t)std::max(1, env_int("PAPR_STREAM_WINDOW_MB", 256)) << 20, ctx->hbm_budget - keep - slack);
    window = window >= group ? window / group * group : window / keep * keep;  // whole groups of the sum program, or whole chunks
    unsigned char *buf = nullptr;
    if (hipMalloc((void **)&buf, keep + window + slack) != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, PAPR_E_NOMEM, "hipMalloc(%zu bytes) for the stream's window failed", keep + window + slack);
    }
    unsigned char *dev = buf + keep;
    std::vector<hipEvent_t> done((size_t)ctx->num_buf, nullptr);
    auto cleanup = [&](int rc) {
        (void)hipStreamSynchronize(ctx->stream);
        for (hipEvent_t e : done)
            if (e)
                (void)hipEventDestroy(e);
        ctx->d_iq = nullptr;  // (the window was never the context's to free)
        release_shard(ctx);
        (void)hipFree(buf);
        return rc;
    };
    if (hipMemsetAsync(buf, 0, keep, ctx->stream) != hipSuccess)
        return cleanup(fail(ctx, PAPR_E_HIP, "hipMemsetAsync failed"));
    papr_stats total;
    papr_stats_init(&total);
    double S = 0.0;              // the reference's accumulator after the windows reduced so far
    bool exact_ok = ctx->exact;  // (until a window's sum is not finite or its program cannot be built: the tree sum stands in)
    uint64_t base = 0, windows = 0;
    size_t fill = 0, bytes_total = 0;
    // one window of `nsamples` samples (the bytes are in HBM): stats, and the exact running sum carried across it
    auto reduce_window = [&](uint64_t nsamples, bool odd) -> int {
        ctx->d_iq = (float *)dev;
        ctx->owns_iq = false;
        ctx->cap = window / 8;
        ctx->n = nsamples;
        ctx->base = base;
        ctx->loaded = ctx->resident = true;
        ctx->have_file_stats = false;
        ctx->exact_valid = ctx->sweep_valid = ctx->exact_swept = ctx->est_groups_valid = ctx->exact_program_launched = false;
        ctx->shard_flags = odd ? PAPR_FLAG_ODD_TAIL : 0;
        papr_stats st;
        int rc = papr_hip_stats(ctx, &st);
        if (rc)
            return rc;
        if (exact_ok && !(std::isfinite(st.sum) && std::isfinite(S)))
            exact_ok = false;  // (NaN / Inf in the stream: the sum is NaN / Inf whatever the order)
        if (exact_ok) {
            const void *prog = nullptr;
            size_t nbytes = 0;
            rc = papr_hip_exact_program(ctx, S, base + nsamples, &prog, &nbytes);
            if (rc == PAPR_OK) {
                const void *progs[1] = {prog};
                rc = papr_exact_chain_continue(&S, progs, &nbytes, 1);
            }
            if (rc == PAPR_E_HIP || rc == PAPR_E_NOMEM)
                return rc;
            if (rc != PAPR_OK)
                exact_ok = false;  // an invariant of the emulation did not hold (never expected): the parallel sum stands in
        }
        if (windows == 0)
            total = st;
        else
            papr_stats_merge(&total, &st);
        windows++;
        base += nsamples;
        return PAPR_OK;
    };
    bool eof = false;
    for (uint64_t k = 0; !eof; k++) {
        const int b = (int)(k % (uint64_t)ctx->num_buf);
        if (done[(size_t)b] && hipEventSynchronize(done[(size_t)b]) != hipSuccess)  // (this buffer's last copy has left it)
            return cleanup(fail(ctx, PAPR_E_HIP, "hipEventSynchronize failed"));
        unsigned char *host = (unsigned char *)ctx->h_stage[b];
        size_t got = 0;
        const double t_read = now_s();
        while (got < ctx->stage_bytes) {
            const ssize_t r = read(fd, host + got, ctx->stage_bytes - got);
            if (r < 0 && errno == EINTR)
                continue;
            if (r < 0)
                return cleanup(fail(ctx, PAPR_E_IO, "read from the stream failed: %s", strerror(errno)));
            if (r == 0) {
                eof = true;
                break;
            }
            got += (size_t)r;
        }
        ctx->ingest.read_s += now_s() - t_read;
        for (size_t off = 0; off < got;) {  // (a staging buffer may end one window and begin the next)
            const size_t part = std::min(got - off, window - fill);
            if (hipMemcpyAsync(dev + fill, host + off, part, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                return cleanup(fail(ctx, PAPR_E_HIP, "hipMemcpyAsync of a stream chunk failed"));
            fill += part;
            off += part;
            bytes_total += part;
            if (fill == window) {
                // a full window in the middle of the stream: whole samples, no tail rule
                int rc = reduce_window(window / 8, false);
                if (rc)
                    return cleanup(rc);
                if (hipMemcpyAsync(buf, dev + window - keep, keep, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                    return cleanup(fail(ctx, PAPR_E_HIP, "keeping the window's last chunk failed"));
                fill = 0;
            }
        }
        if (got) {
            if (!done[(size_t)b] && hipEventCreateWithFlags(&done[(size_t)b], hipEventDisableTiming) != hipSuccess)
                return cleanup(fail(ctx, PAPR_E_HIP, "hipEventCreate failed"));
            if (hipEventRecord(done[(size_t)b], ctx->stream) != hipSuccess)
                return cleanup(fail(ctx, PAPR_E_HIP, "hipEventRecord failed"));
            ctx->ingest.chunks++;
        }
    }
    // the stream's end: the reference's sample count and tail, from the stream's whole length (windows are multiples of 8
    // bytes, so the last window's floats have the stream's parity)
    if (fill || windows == 0) {
        if (hipStreamSynchronize(ctx->stream) != hipSuccess)
            return cleanup(fail(ctx, PAPR_E_HIP, "hipStreamSynchronize failed"));
        const uint64_t nfloats = bytes_total / 4, stray = bytes_total % 4, lfloats = fill / 4;
        const bool odd = (nfloats & 1u) != 0;
        if (odd) {
            unsigned char bytes[4] = {0, 0, 0, 0};
            // partner = the stream's float 16384 before its end, if the stream has that many: in the window, or in the chunk kept
            // in front of it
            if ((nfloats >= 16384 && hipMemcpy(bytes, dev + (ptrdiff_t)lfloats * 4 - 65536, 4, hipMemcpyDeviceToHost) != hipSuccess) ||
                (stray && hipMemcpy(bytes, dev + lfloats * 4, stray, hipMemcpyDeviceToHost) != hipSuccess) ||
                hipMemcpy(dev + lfloats * 4, bytes, 4, hipMemcpyHostToDevice) != hipSuccess)
                return cleanup(fail(ctx, PAPR_E_HIP, "building the phantom sample failed"));
        }
        int rc = reduce_window((lfloats + 1) / 2, odd);
        if (rc)
            return cleanup(rc);
    }
    if (exact_ok)
        total.sum = S;
    *total_out = total;
    if (exact_out)
        *exact_out = exact_ok ? 1 : 0;
    if (windows_out)
        *windows_out = windows;
    const double read_s = ctx->ingest.read_s;
    const uint64_t chunks = ctx->ingest.chunks;
    cleanup(PAPR_OK);
    ctx->ingest.read_s = read_s;
    ctx->ingest.chunks = chunks;
    ctx->ingest.resident = 0;
    ctx->ingest.reader_threads = 1;
    ctx->ingest.file_passes = 1;
    ctx->ingest.drain_s = now_s() - t_begin - read_s;
    return PAPR_OK;
}


int papr_hip_get_ingest_timing(const papr_hip_ctx *ctx, papr_hip_ingest_timing *out)
{
    if (!ctx || !out)
        return PAPR_E_ARG;
    *out = ctx->ingest;
    return PAPR_OK;
}

int papr_hip_estimate_file(papr_hip_ctx *ctx, const char *path, uint64_t first_sample, uint64_t nsamples, papr_stats *est)
{
    return guarded(ctx, [&] { return papr_hip_estimate_file_impl(ctx, path, first_sample, nsamples, est); });
}

int papr_hip_load_file(papr_hip_ctx *ctx, const char *path, uint64_t first_sample, uint64_t nsamples)
{
    return guarded(ctx, [&] { return papr_hip_load_file_impl(ctx, path, first_sample, nsamples); });
}

int papr_hip_load_stream(papr_hip_ctx *ctx, int fd, uint64_t *nsamples)
{
    return guarded(ctx, [&] { return papr_hip_load_stream_impl(ctx, fd, nsamples); });
}

int papr_hip_stream_stats(papr_hip_ctx *ctx, int fd, papr_stats *total, int *exact_sum, uint64_t *windows)
{
    return guarded(ctx, [&] { return papr_hip_stream_stats_impl(ctx, fd, total, exact_sum, windows); });
}

int papr_hip_load_file_sweep(papr_hip_ctx *ctx, const char *path, uint64_t first_sample, uint64_t nsamples,
                             const float *guess_levels, int nlevels)
{
    return guarded(ctx, [&] { return papr_hip_load_file_sweep_impl(ctx, path, first_sample, nsamples, guess_levels, nlevels); });
}

}  // extern "C"
