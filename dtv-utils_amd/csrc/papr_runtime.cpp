// papr_runtime.cpp — host runtime behind the C ABI of include/papr_hip.h:
// context/shard management, the file ingest engine (replaces the fread loops
// of reference papr.c:100-101, 143-144, 175-176), launch sequencing for the two
// passes and the exact threshold-table construction for pass 2.
//
// No CPU compute path: every sample is reduced on the GPU; the host only reads
// file bytes into pinned buffers, builds the <= 16 K-entry level tables and
// folds a handful of scalars.

#include "papr_hip.h"
#include "papr_exact_format.h"
#include "papr_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr uint64_t kChunkAlign = PAPR_TILE_SAMPLES_MAX;  // chunk boundaries stay tile aligned for every variant
constexpr int kNumBuf = 4;      // pinned staging buffers
constexpr int kReadAhead = 2;   // chunks being read ahead of the one being copied
constexpr int kMaxTimed = 4096;

char g_open_error[256] = "";

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- a tiny pool of file-reader threads -------------------------------------
// Jobs are grouped in batches (one batch = the slices of one chunk); the
// submitter can queue the next chunk's batch before waiting for the current
// one, so the readers never go idle between chunks.
struct ReadBatch {
    int pending = 0;
    int error = 0;
};

// CPUs of the NUMA node the GPU hangs off (its PCIe root): the ingest's reader threads run there and the pinned
// staging buffers are first touched there, so that the H2D DMA never crosses the socket interconnect.
// Empty set = unknown / single node / PAPR_NUMA=0.
struct CpuSet {
    cpu_set_t set;
    bool valid = false;
};

CpuSet numa_cpus_of_device(int device)
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
    for (char *tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a = 0, b = 0;
        const int k = sscanf(tok, "%d-%d", &a, &b);
        if (k == 1)
            b = a;
        if (k < 1)
            continue;
        for (int c = a; c <= b && c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &allowed)) {
                CPU_SET(c, &out.set);
                count++;
            }
    }
    out.valid = count > 0 && count < CPU_COUNT(&allowed);  // nothing to gain when the node is all we have
    return out;
}

class ReaderPool {
  public:
    explicit ReaderPool(int n, const CpuSet &cpus = CpuSet())
    {
        for (int i = 0; i < n; i++)
            threads_.emplace_back([this, cpus] {
                if (cpus.valid)
                    (void)sched_setaffinity(0, sizeof(cpus.set), &cpus.set);
                run();
            });
    }
    ~ReaderPool()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_)
            t.join();
    }
    // job returns 0 or an error code, recorded in the batch
    void submit(ReadBatch *batch, std::function<int()> job)
    {
        {
            std::lock_guard<std::mutex> g(m_);
            batch->pending++;
            jobs_.push_back({batch, std::move(job)});
        }
        cv_.notify_one();
    }
    int wait(ReadBatch *batch)
    {
        std::unique_lock<std::mutex> g(m_);
        done_cv_.wait(g, [batch] { return batch->pending == 0; });
        return batch->error;
    }

  private:
    struct Job {
        ReadBatch *batch;
        std::function<int()> fn;
    };
    void run()
    {
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return stop_ || !jobs_.empty(); });
                if (stop_ && jobs_.empty())
                    return;
                job = std::move(jobs_.front());
                jobs_.pop_front();
            }
            const int rc = job.fn();
            {
                std::lock_guard<std::mutex> g(m_);
                if (rc)
                    job.batch->error = rc;
                if (--job.batch->pending == 0)
                    done_cv_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::deque<Job> jobs_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    bool stop_ = false;
};

struct TimedLaunch {
    hipEvent_t a, b;
    int kind;  // 0 stats, 1 ccdf, 2 exact-sum kernels, 3 one-sweep kernel, 4 estimate / stash recount
    uint64_t bytes;
};

}  // namespace

namespace {
struct SweepRun;
}

struct papr_hip_ctx {
    int device = -1;
    hipStream_t stream = nullptr;     // compute
    hipStream_t copy_stream = nullptr;
    char name[128] = "";
    char err[256] = "";
    int num_cus = 256;
    size_t hbm_budget = 0;

    // shard
    float *d_iq = nullptr;   // resident samples (owned or adopted)
    bool owns_iq = false;
    uint64_t cap = 0;        // capacity in samples
    uint64_t n = 0;          // samples in the shard
    uint64_t base = 0;       // global index of sample 0 of the shard
    bool loaded = false;
    bool resident = false;
    uint32_t shard_flags = 0;

    // file source (kept for re-streaming shards that exceed the HBM budget)
    std::string path;
    uint64_t file_first = 0;  // first sample of the range within the file
    bool have_file_stats = false;
    papr_stats file_stats;

    // work buffers
    papr_partial *d_partials = nullptr;
    size_t partials_cap = 0;
    papr_partial *h_result = nullptr;  // pinned, written by the finalize kernel
    papr_partial *h_result_dev = nullptr;
    unsigned long long *d_hist = nullptr;
    unsigned long long *h_hist = nullptr;  // pinned
    uint32_t *d_table = nullptr;
    uint32_t *h_table = nullptr;           // pinned
    size_t table_cap_words = 0;
    unsigned long long *d_nan_key = nullptr;
    float *d_tail = nullptr;               // streaming mode: the last chunk's sub-tile tail

    // ingest
    void *h_stage[kNumBuf] = {};
    void *d_stage[kNumBuf] = {};
    hipEvent_t ev_copy[kNumBuf] = {};
    hipEvent_t ev_kernel[kNumBuf] = {};
    size_t stage_bytes = 0;
    ReaderPool *pool = nullptr;
    int reader_threads = 0;
    bool ingest_numa = false;   // reader threads and staging buffers are bound to the GPU's NUMA node

    // exact-sum mode (papr_exact.hip)
    bool exact = false;
    bool exact_valid = false;        // tile sums of the current shard are on the device
    uint64_t exact_tiles_cap = 0;
    double *d_tile_sums = nullptr;   // ntiles x 4 per-wave sums
    double *d_block_sums = nullptr;
    int32_t *d_tile_E = nullptr;
    double *d_seg_D = nullptr;       // 2 x ntiles pairs
    papr_exact_group *d_groups = nullptr;
    unsigned char *h_program = nullptr;  // pinned + mapped: the pack kernel writes the program straight into it
    size_t h_program_cap = 0;
    uint32_t *d_mixed_list = nullptr, *d_raw_list = nullptr;
    papr_exact_plan *d_plan = nullptr;
    uint32_t *d_ambig = nullptr;   // re-streamed shards: [0, cap) unordered list, [cap, 2 cap) sorted list, [2 cap] count
    float *d_raw_store = nullptr;  // ... and the captured raw tiles

    // one-sweep mode (papr_sweep.hip)
    unsigned long long *d_sweep_hist = nullptr;  // 2 L + 2 bins, then one stash-segment length per workgroup
    unsigned long long *h_sweep_hist = nullptr;  // pinned
    float *d_stash = nullptr;                    // in-band powers of the last sweep
    uint64_t stash_cap = 0;
    bool sweep_valid = false;                    // the fields below describe the CURRENT shard
    uint32_t sweep_half = 0;                     // half-width of a band, in bit patterns
    std::vector<uint32_t> sweep_keys;            // unique guessed keys (band centres), ascending
    std::vector<uint64_t> sweep_even_above;      // per guessed key j: samples in even bins >= 2 j + 2
    uint64_t sweep_stash_count = 0;
    uint64_t sweep_seg_cap = 0;                  // floats per stash segment
    uint32_t sweep_nsegs = 0, sweep_nbins = 0;
    bool sweep_overflow = false;
    papr_hip_sweep_info sweep_info{};
    const SweepRun *ingest_run = nullptr;        // set while papr_hip_load_file_sweep streams the file in

    papr_hip_ingest_timing ingest{};
    papr_hip_tuning tune{};
    bool timing = false;
    std::vector<TimedLaunch> timed;
    size_t timed_used = 0;
};

namespace {

int fail(papr_hip_ctx *ctx, int code, const char *fmt, ...)
{
    char *dst = ctx ? ctx->err : g_open_error;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 256, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(ctx, call)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, PAPR_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_));      \
    } while (0)

int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

void parse_tune_env(papr_hip_tuning *t)
{
    const char *v = getenv("PAPR_HIP_TUNE");
    if (!v)
        return;
    std::string s(v);
    size_t pos = 0;
    while (pos < s.size()) {
        size_t end = s.find(',', pos);
        if (end == std::string::npos)
            end = s.size();
        std::string kv = s.substr(pos, end - pos);
        size_t eq = kv.find('=');
        if (eq != std::string::npos) {
            std::string k = kv.substr(0, eq);
            int val = atoi(kv.c_str() + eq + 1);
            if (k == "blocks") t->stats_blocks = t->ccdf_blocks = val;
            else if (k == "variant") t->stats_variant = t->ccdf_variant = val + 1;
            else if (k == "sblocks") t->stats_blocks = val;
            else if (k == "svariant") t->stats_variant = val + 1;
            else if (k == "cblocks") t->ccdf_blocks = val;
            else if (k == "cvariant") t->ccdf_variant = val + 1;
            else if (k == "map") t->stats_map = t->ccdf_map = val + 1;
            else if (k == "smap") t->stats_map = val + 1;
            else if (k == "cmap") t->ccdf_map = val + 1;
            else if (k == "nt") t->nontemporal = val ? 1 : 2;
            else if (k == "copies") t->hist_copies = val;
            else if (k == "search") t->flags = val ? (t->flags | 1) : (t->flags & ~1);
            else if (k == "wblocks") t->sweep_blocks = val;
            else if (k == "wvariant") t->sweep_variant = val + 1;
            else if (k == "wmap") t->sweep_map = val + 1;
            else if (k == "band") t->sweep_band_log2 = val;
            else if (k == "ratio") t->estimate_ratio = val;
        }
        pos = end + 1;
    }
}

// Built-in launch geometry, from the 10 GiB sweeps on MI355X:
//   pass 1: 256-thread workgroups, 4 loads per lane, next-tile prefetch, 2 workgroups per CU (8 waves/CU),
//           grid-stride tiles                                                -> 7.2-7.3 TB/s
//   pass 2: 512-thread workgroups, 4 loads per lane, 2 workgroups per CU (16 waves/CU), grid-stride tiles
//                                                                            -> 7.20 TB/s
// (one contiguous eighth of the shard per XCD is 1 % faster for pass 1 in most processes and 8 % slower in about
// one process in four — it depends on where the allocation landed — so it is not the default)
constexpr int kStatsVariant = 1, kStatsPerCU = 2, kStatsMap = PAPR_MAP_GRID_STRIDE;
constexpr int kCcdfVariant = 13, kCcdfPerCU = 2, kCcdfMap = PAPR_MAP_GRID_STRIDE;

// one-sweep kernel (pass 1 + banded pass 2 in one read)
constexpr int kSweepVariant = 13, kSweepPerCU = 2, kSweepMap = PAPR_MAP_GRID_STRIDE;
constexpr int kSweepBandLog2 = 14, kEstimateRatio = 64;
constexpr uint64_t kEstimateMinTiles = 8192;  // sample at least 16 Mi samples (or everything)

enum Pass { PASS1 = 0, PASS2 = 1, SWEEP = 2 };

int variant_of(const papr_hip_ctx *ctx, Pass p)
{
    if (p == PASS1 && ctx->exact)
        return 1;  // 256 x 4 pipelined: the geometry papr_launch_stats_tilesums is built for
    if (p == SWEEP) {
        const int v = papr_sweep_variant(ctx->tune.sweep_variant - 1);
        return v >= 0 ? v : kSweepVariant;
    }
    const int v = (p == PASS1 ? ctx->tune.stats_variant : ctx->tune.ccdf_variant) - 1;
    int b, u;
    if (v >= 0 && papr_variant_geometry(v, &b, &u) == 0)
        return v;
    return p == PASS1 ? kStatsVariant : kCcdfVariant;
}

int blocks_of(const papr_hip_ctx *ctx, Pass p)
{
    const int b = p == PASS1 ? ctx->tune.stats_blocks : p == PASS2 ? ctx->tune.ccdf_blocks : ctx->tune.sweep_blocks;
    return b > 0 ? b : ctx->num_cus * (p == PASS1 ? kStatsPerCU : p == PASS2 ? kCcdfPerCU : kSweepPerCU);
}

// samples one workgroup consumes per loop iteration under the pass's kernel variant
uint64_t tile_samples(const papr_hip_ctx *ctx, Pass p)
{
    int block = 256, unroll = 8;
    (void)papr_variant_geometry(variant_of(ctx, p), &block, &unroll);
    return 2ull * (uint64_t)block * (uint64_t)unroll;
}

int map_of(const papr_hip_ctx *ctx, Pass p)
{
    const int m = (p == PASS1 ? ctx->tune.stats_map : p == PASS2 ? ctx->tune.ccdf_map : ctx->tune.sweep_map) - 1;
    if (m >= 0 && m <= 2)
        return m;
    if (p == SWEEP)
        return kSweepMap;
    // exact-sum mode: the per-tile sums of neighbouring tiles are then written by workgroups of the same
    // XCD (1.70 ms vs 1.83 ms per 10 GiB with grid-stride; its slow mode costs no more than that)
    if (p == PASS1 && ctx->exact)
        return PAPR_MAP_XCD_SPAN;
    return p == PASS1 ? kStatsMap : kCcdfMap;
}

int pick_blocks(const papr_hip_ctx *ctx, Pass p, uint64_t ntiles)
{
    int blocks = blocks_of(ctx, p);
    if ((uint64_t)blocks > ntiles)
        blocks = (int)std::max<uint64_t>(ntiles, 1);
    if (map_of(ctx, p) == PAPR_MAP_XCD_SPAN && blocks >= 8)
        blocks &= ~7;
    return blocks;
}

int effective_map(const papr_hip_ctx *ctx, Pass p, int blocks)
{
    int map = map_of(ctx, p);
    if (map == PAPR_MAP_XCD_SPAN && (blocks % 8) != 0)
        map = PAPR_MAP_GRID_STRIDE;
    return map;
}

bool use_nt(const papr_hip_ctx *ctx)
{
    return ctx->tune.nontemporal != 2;
}

int ensure_partials(papr_hip_ctx *ctx, size_t count)
{
    if (count <= ctx->partials_cap)
        return PAPR_OK;
    if (ctx->d_partials)
        HIPCHK(ctx, hipFree(ctx->d_partials));
    ctx->d_partials = nullptr;
    ctx->partials_cap = 0;
    size_t cap = std::max<size_t>(count, 4096);
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_partials, cap * sizeof(papr_partial)));
    ctx->partials_cap = cap;
    return PAPR_OK;
}

int ensure_table(papr_hip_ctx *ctx, size_t words)
{
    if (words <= ctx->table_cap_words)
        return PAPR_OK;
    if (ctx->d_table) HIPCHK(ctx, hipFree(ctx->d_table));
    if (ctx->h_table) HIPCHK(ctx, hipHostFree(ctx->h_table));
    ctx->d_table = nullptr;
    ctx->h_table = nullptr;
    ctx->table_cap_words = 0;
    size_t cap = std::max<size_t>(words, 16384);
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_table, cap * sizeof(uint32_t)));
    HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_table, cap * sizeof(uint32_t), hipHostMallocDefault));
    ctx->table_cap_words = cap;
    return PAPR_OK;
}

void release_shard(papr_hip_ctx *ctx)
{
    if (ctx->owns_iq && ctx->d_iq)
        (void)hipFree(ctx->d_iq);
    ctx->d_iq = nullptr;
    ctx->owns_iq = false;
    ctx->cap = ctx->n = ctx->base = 0;
    ctx->loaded = ctx->resident = false;
    ctx->have_file_stats = false;
    ctx->exact_valid = false;
    ctx->sweep_valid = false;
    ctx->shard_flags = 0;
    ctx->path.clear();
}

int ensure_owned_capacity(papr_hip_ctx *ctx, uint64_t nsamples)
{
    if (ctx->d_iq && ctx->cap >= nsamples)
        return PAPR_OK;
    release_shard(ctx);
    // one extra tile of slack keeps every 16-byte lane load in bounds
    size_t bytes = (size_t)(nsamples + PAPR_TILE_SAMPLES_MAX) * 8;
    hipError_t e = hipMalloc((void **)&ctx->d_iq, bytes);
    if (e != hipSuccess) {
        ctx->d_iq = nullptr;
        return fail(ctx, PAPR_E_NOMEM, "hipMalloc(%zu bytes) for the shard failed: %s", bytes, hipGetErrorString(e));
    }
    ctx->owns_iq = true;
    ctx->cap = nsamples;
    return PAPR_OK;
}

void time_begin(papr_hip_ctx *ctx, int kind, uint64_t bytes)
{
    if (!ctx->timing || ctx->timed_used >= (size_t)kMaxTimed)
        return;
    if (ctx->timed_used == ctx->timed.size()) {
        TimedLaunch t{};
        if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess)
            return;
        ctx->timed.push_back(t);
    }
    TimedLaunch &t = ctx->timed[ctx->timed_used];
    t.kind = kind;
    t.bytes = bytes;
    (void)hipEventRecord(t.a, ctx->stream);
}

void time_end(papr_hip_ctx *ctx)
{
    if (!ctx->timing || ctx->timed_used >= ctx->timed.size() || ctx->timed_used >= (size_t)kMaxTimed)
        return;
    (void)hipEventRecord(ctx->timed[ctx->timed_used].b, ctx->stream);
    ctx->timed_used++;
}

// exact-sum mode: device buffers sized for the current shard
int ensure_exact_buffers(papr_hip_ctx *ctx)
{
    const uint64_t ntiles = ctx->n / PAPR_EXACT_TILE_SAMPLES;
    if (ntiles <= ctx->exact_tiles_cap && ctx->d_tile_sums)
        return PAPR_OK;
    if (ctx->d_tile_sums) (void)hipFree(ctx->d_tile_sums);
    if (ctx->d_block_sums) (void)hipFree(ctx->d_block_sums);
    if (ctx->d_tile_E) (void)hipFree(ctx->d_tile_E);
    if (ctx->d_seg_D) (void)hipFree(ctx->d_seg_D);
    if (ctx->d_groups) (void)hipFree(ctx->d_groups);
    ctx->d_tile_sums = ctx->d_block_sums = ctx->d_seg_D = nullptr;
    ctx->d_tile_E = nullptr;
    ctx->d_groups = nullptr;
    ctx->exact_tiles_cap = 0;
    const uint64_t cap = std::max<uint64_t>(ntiles, 1024);
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_tile_sums, cap * PAPR_EXACT_TILE_WAVES * sizeof(double)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_block_sums, (cap / 1024 + 2) * sizeof(double)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_tile_E, cap * sizeof(int32_t)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_seg_D, cap * 2 * 2 * sizeof(double)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_groups, (cap / PAPR_EXACT_GROUP_TILES + 2) * sizeof(papr_exact_group)));
    ctx->exact_tiles_cap = cap;
    return PAPR_OK;
}

// ---- pass 1 over device-resident samples -------------------------------------
// Launches the streaming kernel over the full tiles of [data, data + n) and
// returns how many partial records it appended at d_partials + slot.
int launch_stats_range(papr_hip_ctx *ctx, const float *data, uint64_t n, uint64_t base_index, size_t slot,
                       int *nrecords)
{
    const uint64_t tile = tile_samples(ctx, PASS1);
    const uint64_t ntiles = n / tile;
    *nrecords = 0;
    if (ntiles == 0)
        return PAPR_OK;
    int blocks = pick_blocks(ctx, PASS1, ntiles);
    const int map = effective_map(ctx, PASS1, blocks);
    if (ctx->exact) {
        int rc = ensure_exact_buffers(ctx);
        if (rc)
            return rc;
    }
    time_begin(ctx, 0, ntiles * tile * 8);
    if (ctx->exact) {
        papr_launch_stats_tilesums(ctx->stream, blocks, data, ntiles, base_index, map, ctx->d_partials + slot,
                                   ctx->d_tile_sums, (base_index - ctx->base) / PAPR_EXACT_TILE_SAMPLES);
    } else {
        papr_launch_stats(ctx->stream, variant_of(ctx, PASS1), blocks, use_nt(ctx), data, ntiles, base_index, map,
                          ctx->d_partials + slot);
    }
    time_end(ctx);
    HIPCHK(ctx, hipGetLastError());
    *nrecords = blocks;
    return PAPR_OK;
}

void partial_to_stats(const papr_partial &r, uint64_t n, papr_stats *out)
{
    papr_stats_init(out);
    out->sum = r.sum;
    out->n = n;
    out->peak = r.val[0];    out->peak_idx = r.idx[0];
    out->re_pos = r.val[1];  out->re_pos_idx = r.idx[1];
    out->re_neg = r.val[2];  out->re_neg_idx = r.idx[2];
    out->im_pos = r.val[3];  out->im_pos_idx = r.idx[3];
    out->im_neg = r.val[4];  out->im_neg_idx = r.idx[4];
}

void apply_nan_key(papr_stats *out, unsigned long long key)
{
    if (key == ~0ull)
        return;
    out->flags |= PAPR_FLAG_NAN;
    out->nan_first_idx = key >> 1;
    out->nan_first_neg = (uint32_t)(key & 1u);
    // papr.c:104 — the running double sum takes the first NaN power (with the
    // sign x86 propagates) and keeps it
    out->sum = out->nan_first_neg ? -(double)NAN : (double)NAN;
}

// ---- pass 2 table construction ----------------------------------------------
constexpr uint32_t kNever = 0xFFFFFFFFu;

// smallest bit pattern of a non-negative float that is > t: papr_level_key (papr_host.c)
inline uint32_t level_key(float t)
{
    return papr_level_key(t);
}

struct CcdfPlan {
    std::vector<uint32_t> keys;      // unique, ascending
    std::vector<int> pos;            // per level: index into keys, or -1
    papr_ccdf_params P{};
    bool lut = false;
    size_t lds_bytes = 0;
};

// LUT / search form and LDS layout for plan->keys (unique, ascending); `vblock` = threads of the workgroup that
// will use it, `extra_lds` = what else that workgroup keeps in LDS
int finish_plan(papr_hip_ctx *ctx, CcdfPlan *plan, int vblock, size_t extra_lds)
{
    const uint32_t m = (uint32_t)plan->keys.size();
    papr_ccdf_params &P = plan->P;
    memset(&P, 0, sizeof(P));
    P.nkeys = m;
    plan->lut = false;
    plan->lds_bytes = 0;
    if (m == 0)
        return PAPR_OK;
    const uint32_t nbins = m + 1;
    const size_t lds_cap = (size_t)papr_ccdf_max_dynamic_lds();
    const int waves = vblock / 64;
    const int want_copies = ctx->tune.hist_copies > 0 ? std::min(ctx->tune.hist_copies, waves) : std::min(waves, 4);

    // LUT: the coarsest cell size that still isolates every key in its own cell
    if (plan->keys.front() >= 0x00800000u && !(ctx->tune.flags & 1)) {  // keys in the normal-float range
        for (int shift = 23; shift >= 8; shift--) {
            const uint32_t c0 = plan->keys.front() >> shift, c1 = plan->keys.back() >> shift;
            const uint64_t ncells = (uint64_t)c1 - c0 + 1;
            if (ncells * 8 > 40 * 1024)
                break;  // finer cells only get bigger
            bool unique_cells = true;
            for (uint32_t k = 1; k < m && unique_cells; k++)
                unique_cells = (plan->keys[k] >> shift) != (plan->keys[k - 1] >> shift);
            if (!unique_cells)
                continue;
            P.shift = (uint32_t)shift;
            P.cell_lo = c0;
            P.ncells = (uint32_t)ncells;
            P.above_lo = (uint32_t)(((uint64_t)c1 + 1) << shift);
            P.above_span = 0x7F800000u - P.above_lo;
            P.table_words = 2 * P.ncells;
            plan->lut = true;
            break;
        }
    }
    if (!plan->lut) {
        P.table_words = m;
        uint32_t step = 1;
        while (step * 2 <= m)
            step *= 2;
        P.search_step = step;
    }
    // histogram copies: one per wave when it is cheap, fewer for huge tables
    int copies = want_copies;
    const size_t soft_cap = (size_t)vblock * 80;  // the workgroup's share of 160 KiB when the CU is full of threads
    while (copies > 1 && (size_t)P.table_words * 4 + (size_t)copies * nbins * 4 + extra_lds > soft_cap)
        copies--;
    P.copies = (uint32_t)copies;
    plan->lds_bytes = (size_t)P.table_words * 4 + (size_t)copies * nbins * 4;
    if (plan->lds_bytes + extra_lds > lds_cap)
        return fail(ctx, PAPR_E_LIMIT, "level table needs %zu bytes of LDS (limit %zu)", plan->lds_bytes + extra_lds,
                    lds_cap);
    return PAPR_OK;
}

int plan_ccdf(papr_hip_ctx *ctx, const float *levels, int nlevels, CcdfPlan *plan)
{
    plan->keys.clear();
    plan->pos.assign(nlevels, -1);
    std::vector<uint32_t> all(nlevels);
    for (int j = 0; j < nlevels; j++) {
        all[j] = level_key(levels[j]);
        if (all[j] != kNever)
            plan->keys.push_back(all[j]);
    }
    std::sort(plan->keys.begin(), plan->keys.end());
    plan->keys.erase(std::unique(plan->keys.begin(), plan->keys.end()), plan->keys.end());
    for (int j = 0; j < nlevels; j++)
        if (all[j] != kNever)
            plan->pos[j] = (int)(std::lower_bound(plan->keys.begin(), plan->keys.end(), all[j]) - plan->keys.begin());
    int vblock = 256, vunroll = 8;
    (void)papr_variant_geometry(variant_of(ctx, PASS2), &vblock, &vunroll);
    return finish_plan(ctx, plan, vblock, 0);
}

int upload_ccdf_table(papr_hip_ctx *ctx, const CcdfPlan &plan)
{
    const papr_ccdf_params &P = plan.P;
    int rc = ensure_table(ctx, P.table_words);
    if (rc)
        return rc;
    if (plan.lut) {
        // lut[cell] = {keys strictly below this cell, key inside this cell or never}
        uint32_t k = 0;
        for (uint32_t c = 0; c < P.ncells; c++) {
            uint32_t in_cell = kNever;
            const uint32_t below = k;
            if (k < P.nkeys && (plan.keys[k] >> P.shift) == P.cell_lo + c)
                in_cell = plan.keys[k++];
            ctx->h_table[2 * c] = below;
            ctx->h_table[2 * c + 1] = in_cell;
        }
    } else {
        memcpy(ctx->h_table, plan.keys.data(), (size_t)P.nkeys * 4);
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_table, ctx->h_table, (size_t)P.table_words * 4, hipMemcpyHostToDevice, ctx->stream));
    return PAPR_OK;
}

int launch_ccdf_range(papr_hip_ctx *ctx, const CcdfPlan &plan, const float *data, uint64_t n)
{
    const uint64_t tile = tile_samples(ctx, PASS2);
    const uint64_t ntiles = n / tile;
    const uint32_t tail = (uint32_t)(n - ntiles * tile);
    int blocks = pick_blocks(ctx, PASS2, ntiles);
    const int map = effective_map(ctx, PASS2, blocks);
    time_begin(ctx, 1, n * 8);
    papr_launch_ccdf(ctx->stream, variant_of(ctx, PASS2), blocks, use_nt(ctx), plan.lut, plan.lds_bytes, data, ntiles, map,
                     data + 2 * ntiles * tile, tail, ctx->d_table, plan.P, ctx->d_hist);
    time_end(ctx);
    HIPCHK(ctx, hipGetLastError());
    return PAPR_OK;
}

// ---- file source ---------------------------------------------------------------
struct FileSrc {
    int fd = -1;
    int fd_direct = -1;  // O_DIRECT view of the same file (PAPR_O_DIRECT=1), -1 when not usable
    uint64_t size = 0, nfloats = 0, nsamples = 0;
    bool odd = false;
    float partner = 0.0f;  // Q of the phantom sample
};

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
    // offset, address and length; slices are cut that way, the request is rounded up and a short count
    // at end of file is expected.  Anything that does not fit falls through to the buffered descriptor.
    if (fs.fd_direct >= 0 && (byte0 & 4095) == 0 && ((uintptr_t)dst & 4095) == 0) {
        while (done < want) {
            const uint64_t ask = std::min<uint64_t>((want - done + 4095) & ~4095ull, (uint64_t)1 << 30);
            ssize_t got = pread(fs.fd_direct, dst + done, ask, (off_t)(byte0 + done));
            if (got <= 0 || (got & 4095) != 0) {
                if (got > 0)
                    done += std::min<uint64_t>((uint64_t)got, want - done);
                break;  // error, or the unaligned end of the file: the buffered path finishes the job
            }
            done += std::min<uint64_t>((uint64_t)got, want - done);
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

int ensure_ingest(papr_hip_ctx *ctx, bool need_device_stage)
{
    if (!ctx->stage_bytes) {
        size_t mb = (size_t)std::max(1, env_int("PAPR_CHUNK_MB", 16));
        ctx->stage_bytes = (mb << 20) / (kChunkAlign * 8) * (kChunkAlign * 8);
        if (!ctx->stage_bytes)
            ctx->stage_bytes = kChunkAlign * 8;
    }
    const CpuSet near_gpu = numa_cpus_of_device(ctx->device);
    for (int b = 0; b < kNumBuf; b++) {
        if (!ctx->h_stage[b]) {
            // pinned pages are placed where they are first touched: do that on the GPU's NUMA node
            cpu_set_t before;
            const bool moved = near_gpu.valid && sched_getaffinity(0, sizeof(before), &before) == 0 &&
                               sched_setaffinity(0, sizeof(near_gpu.set), &near_gpu.set) == 0;
            const hipError_t e = hipHostMalloc(&ctx->h_stage[b], ctx->stage_bytes, hipHostMallocDefault);
            if (e == hipSuccess && moved)
                memset(ctx->h_stage[b], 0, ctx->stage_bytes);
            if (moved)
                (void)sched_setaffinity(0, sizeof(before), &before);
            HIPCHK(ctx, e);
        }
        if (!ctx->ev_copy[b])
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_copy[b], hipEventDisableTiming));
        if (!ctx->ev_kernel[b])
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_kernel[b], hipEventDisableTiming));
        if (need_device_stage && !ctx->d_stage[b])
            HIPCHK(ctx, hipMalloc(&ctx->d_stage[b], ctx->stage_bytes + PAPR_TILE_SAMPLES_MAX * 8));
    }
    if (need_device_stage && !ctx->d_tail)
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_tail, PAPR_TILE_SAMPLES_MAX * 8));
    if (!ctx->pool) {
        int n = env_int("PAPR_READ_THREADS", 0);
        if (n <= 0)
            n = (int)std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
        ctx->reader_threads = n;
        ctx->pool = new ReaderPool(n, near_gpu);
        ctx->ingest_numa = near_gpu.valid;
    }
    return PAPR_OK;
}

constexpr uint32_t kCapMixed = 256, kCapRaw = 512;  // beyond this the program is assembled by the host path

// exact-sum mode on a re-streamed shard: the fused sweep (rounding functions + pass 2) over one staged chunk,
// and the unprovable tiles of that chunk kept for the sum program
int launch_fused_chunk(papr_hip_ctx *ctx, const CcdfPlan &plan, const float *chunk, uint64_t s0, uint64_t cnt, bool last)
{
    const uint64_t tile0 = s0 / PAPR_EXACT_TILE_SAMPLES, ntiles = cnt / PAPR_EXACT_TILE_SAMPLES;
    const uint32_t tail = (uint32_t)(cnt - ntiles * PAPR_EXACT_TILE_SAMPLES);  // only the last chunk has one
    const uint64_t nsegs = 2 * ntiles;
    const uint64_t wg_waves = (uint64_t)papr_exact_fused_waves();
    const int per_cu = std::max(1, env_int("PAPR_EXACT_WG_PER_CU", 2));
    const int blocks =
        (int)std::max<uint64_t>(1, std::min<uint64_t>((nsegs + wg_waves - 1) / wg_waves, (uint64_t)ctx->num_cus * per_cu));
    time_begin(ctx, 2, cnt * 8);
    papr_launch_exact_segments_ccdf(ctx->stream, blocks, chunk, nsegs, ctx->d_tile_E + tile0, ctx->d_seg_D + 4 * tile0,
                                    chunk + 2 * ntiles * PAPR_EXACT_TILE_SAMPLES, tail, ctx->d_table, plan.P,
                                    plan.lds_bytes, ctx->d_hist);
    time_end(ctx);
    papr_launch_exact_capture(ctx->stream, chunk, tile0, ntiles, ctx->d_ambig + kCapRaw, ctx->d_ambig + 2 * kCapRaw,
                              kCapRaw, ctx->d_raw_store);
    HIPCHK(ctx, hipGetLastError());
    if (last && tail)
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_tail, chunk + 2 * ntiles * PAPR_EXACT_TILE_SAMPLES, (size_t)tail * 8,
                                   hipMemcpyDeviceToDevice, ctx->stream));
    return PAPR_OK;
}

// ---- one-sweep mode: set-up, launch and bookkeeping shared by resident shards and file ingest -------------------
struct SweepRun {
    CcdfPlan bands;               // band edges lo_0 < hi_0 < lo_1 < ... in LUT form
    std::vector<uint32_t> gkeys;  // guessed keys (band centres), unique, ascending
    uint32_t half = 0;            // half-width of a band in bit patterns
    int variant = 0;
    int blocks = 0;               // workgroups of the largest launch (= stash segments)
    uint64_t tile = 0;            // samples per workgroup iteration
    size_t stash_lds = 0;
    uint32_t nbins = 0;           // 2 * bands + 1 + the NaN trash bin
    uint64_t seg_cap = 0;         // floats per stash segment
};

// Plan the bands for `guess_levels`, size and clear the buffers, upload the LUT.  *reason != PAPR_SWEEP_OK: the guess
// has no band form (or memory is short) and the caller runs the plain pass instead.  `n_shard` sizes the stash,
// `n_launch` (a whole resident shard, or one ingest chunk) the grid.
int sweep_prepare(papr_hip_ctx *ctx, const float *guess_levels, int nlevels, uint64_t n_shard, uint64_t n_launch,
                  SweepRun *run, int *reason)
{
    papr_hip_sweep_info &info = ctx->sweep_info;
    info.band_log2 = ctx->tune.sweep_band_log2 > 0 ? ctx->tune.sweep_band_log2 : kSweepBandLog2;
    *reason = PAPR_SWEEP_NO_BANDS;
    if (nlevels <= 0 || nlevels > PAPR_HIP_MAX_LEVELS)
        return PAPR_OK;
    // widest band (<= the configured width) that has a band form (papr_sweep_bands) and whose edges have a LUT form
    int vblock = 512;
    run->variant = variant_of(ctx, SWEEP);
    (void)papr_sweep_geometry(run->variant, &vblock, &run->tile, &run->stash_lds);
    std::vector<uint32_t> &gkeys = run->gkeys;
    CcdfPlan &bands = run->bands;
    run->half = 0;
    for (int log2w = info.band_log2; log2w >= std::max(info.band_log2 - 3, 8) && !run->half; log2w--) {
        gkeys.assign((size_t)nlevels, 0);
        bands.keys.assign(2 * (size_t)nlevels, 0);
        const int m = papr_sweep_bands(guess_levels, nlevels, log2w, gkeys.data(), bands.keys.data());
        if (m <= 0)
            continue;  // (a narrower band may still fit between crowded thresholds)
        gkeys.resize((size_t)m);
        bands.keys.resize(2 * (size_t)m);
        char keep[sizeof(ctx->err)];
        memcpy(keep, ctx->err, sizeof(keep));
        const bool fits = finish_plan(ctx, &bands, vblock, run->stash_lds) == PAPR_OK && bands.lut;
        memcpy(ctx->err, keep, sizeof(keep));  // not an error of this call: a narrower band or the plain pass follows
        if (fits) {
            run->half = 1u << log2w;
            info.band_log2 = log2w;
        }
    }
    if (!run->half)
        return PAPR_OK;
    // the sweep kernel's LUT has a sentinel cell at either end and its histogram one more (NaN) bin
    bands.P.table_words = 2 * (bands.P.ncells + 2);
    run->nbins = bands.P.nkeys + 2;
    bands.lds_bytes = (size_t)bands.P.table_words * 4 + (size_t)bands.P.copies * run->nbins * 4;
    if (bands.lds_bytes + run->stash_lds > (size_t)papr_ccdf_max_dynamic_lds())
        return PAPR_OK;

    run->blocks = pick_blocks(ctx, SWEEP, n_launch / run->tile);
    // buffers: band histogram with the stash-segment lengths right behind it; stash = 1/4 of the shard's samples
    // (as floats: 1/8 of its bytes), one equal segment per workgroup
    constexpr size_t kMaxSweepBlocks = 65536;
    if (!ctx->d_sweep_hist) {
        const size_t bytes = (2 * (size_t)PAPR_HIP_MAX_LEVELS + 2 + kMaxSweepBlocks) * sizeof(unsigned long long);
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_sweep_hist, bytes));
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_sweep_hist, bytes, hipHostMallocDefault));
    }
    run->seg_cap = std::max<uint64_t>((n_shard / 4 / (uint64_t)run->blocks + 3) & ~3ull, 4096);
    const uint64_t want_stash = run->seg_cap * (uint64_t)run->blocks;
    if (ctx->stash_cap < want_stash) {
        if (ctx->d_stash) HIPCHK(ctx, hipFree(ctx->d_stash));
        ctx->d_stash = nullptr;
        ctx->stash_cap = 0;
        if (hipMalloc((void **)&ctx->d_stash, want_stash * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            ctx->d_stash = nullptr;
            *reason = PAPR_SWEEP_STASH_FULL;
            return PAPR_OK;
        }
        ctx->stash_cap = want_stash;
    }
    int rc = ensure_table(ctx, bands.P.table_words);
    if (rc)
        return rc;
    {
        // lut[0] = below everything, lut[1 + c] = {edges below cell c, the edge inside it or never},
        // lut[ncells + 1] = above every edge; a NaN pattern compares >= 0x7F800001 and lands in the trash bin
        const papr_ccdf_params &P = bands.P;
        uint32_t *tab = ctx->h_table;
        tab[0] = 0;
        tab[1] = kNever;
        uint32_t k = 0;
        for (uint32_t c = 0; c < P.ncells; c++) {
            uint32_t in_cell = kNever;
            const uint32_t below = k;
            if (k < P.nkeys && (bands.keys[k] >> P.shift) == P.cell_lo + c)
                in_cell = bands.keys[k++];
            tab[2 * (c + 1)] = below;
            tab[2 * (c + 1) + 1] = in_cell;
        }
        tab[2 * (P.ncells + 1)] = P.nkeys;
        tab[2 * (P.ncells + 1) + 1] = 0x7F800001u;
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_table, ctx->h_table, (size_t)P.table_words * 4, hipMemcpyHostToDevice,
                                   ctx->stream));
    }
    HIPCHK(ctx, hipMemsetAsync(ctx->d_sweep_hist, 0, ((size_t)run->nbins + (size_t)run->blocks) * sizeof(unsigned long long),
                               ctx->stream));
    *reason = PAPR_SWEEP_OK;
    return PAPR_OK;
}

// One launch of the sweep kernel over [data, data + n): full tiles by the grid, the sub-tile remainder binned by the
// last workgroup (its pass-1 half belongs to papr_stats_finalize).  Histogram and stash segments accumulate over launches.
int sweep_launch(papr_hip_ctx *ctx, const SweepRun &run, const float *data, uint64_t n, uint64_t base_index, size_t slot,
                 int *nrecords)
{
    const uint64_t ntiles = n / run.tile;
    const uint32_t tail = (uint32_t)(n - ntiles * run.tile);
    const int blocks = (int)std::min<uint64_t>((uint64_t)run.blocks, std::max<uint64_t>(ntiles, 1));
    const int map = effective_map(ctx, SWEEP, blocks);
    int rc = ensure_partials(ctx, slot + (size_t)blocks + 1);
    if (rc)
        return rc;
    time_begin(ctx, 3, n * 8);
    papr_launch_sweep(ctx->stream, run.variant, blocks, run.bands.lds_bytes + run.stash_lds, data, ntiles, base_index, map,
                      ctx->d_partials + slot, data + 2 * (n - tail), tail, ctx->d_table, run.bands.P, ctx->d_sweep_hist,
                      ctx->d_stash, ctx->d_sweep_hist + run.nbins, run.seg_cap);
    time_end(ctx);
    HIPCHK(ctx, hipGetLastError());
    *nrecords = blocks;
    return PAPR_OK;
}

// queue the copy of the band histogram + segment lengths to the host (valid after the next stream synchronisation)
int sweep_fetch(papr_hip_ctx *ctx, const SweepRun &run)
{
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_sweep_hist, ctx->d_sweep_hist,
                               ((size_t)run.nbins + (size_t)run.blocks) * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                               ctx->stream));
    return PAPR_OK;
}

// what the sweep decided: samples in even bins above each band; odd bins are exactly the stash
int sweep_collect(papr_hip_ctx *ctx, const SweepRun &run)
{
    papr_hip_sweep_info &info = ctx->sweep_info;
    const unsigned long long *H = ctx->h_sweep_hist;
    uint64_t stash_count = 0, in_bands = 0;
    bool overflow = false;
    for (int b = 0; b < run.blocks; b++) {
        stash_count += H[run.nbins + b];
        overflow = overflow || H[run.nbins + b] > run.seg_cap;
    }
    for (uint32_t b = 1; b < run.nbins; b += 2)
        in_bands += H[b];
    if (in_bands != stash_count)
        return fail(ctx, PAPR_E_INTERNAL, "one-sweep invariant broken: %llu samples binned inside bands, %llu stashed",
                    (unsigned long long)in_bands, (unsigned long long)stash_count);
    const size_t m = run.gkeys.size();
    ctx->sweep_even_above.assign(m, 0);
    uint64_t above = 0;
    for (size_t j = m; j-- > 0;) {
        above += H[2 * j + 2];
        ctx->sweep_even_above[j] = above;
    }
    ctx->sweep_keys = run.gkeys;
    ctx->sweep_half = run.half;
    ctx->sweep_stash_count = stash_count;
    ctx->sweep_seg_cap = run.seg_cap;
    ctx->sweep_nsegs = (uint32_t)run.blocks;
    ctx->sweep_nbins = run.nbins;
    ctx->sweep_overflow = overflow;
    ctx->sweep_valid = true;
    info.swept = 1;
    info.reason = PAPR_SWEEP_OK;
    info.stash_samples = stash_count;
    info.stash_capacity = run.seg_cap * (uint64_t)run.blocks;  // what this sweep could use (one segment per workgroup)
    return PAPR_OK;
}

enum StreamPass { PASS_LOAD_STATS, PASS_STREAM_STATS, PASS_STREAM_CCDF, PASS_STREAM_CCDF_EXACT, PASS_STREAM_NAN };

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
    const bool to_resident = (pass == PASS_LOAD_STATS);
    const bool timed = (pass == PASS_LOAD_STATS || pass == PASS_STREAM_STATS);
    double t_mark = now_s();
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
    // queue the slices of chunk c for the reader threads (buffer c % kNumBuf must be free)
    std::vector<ReadBatch> batches(nchunks);
    const FileSrc *fsp = &fs;
    const uint64_t file_first = ctx->file_first, shard_n = ctx->n;
    auto submit_chunk = [&](uint64_t c) {
        const int b = (int)(c % kNumBuf);
        const uint64_t s0 = c * chunk_samples;
        const uint64_t cnt = std::min(chunk_samples, shard_n - s0);
        unsigned char *hbuf = (unsigned char *)ctx->h_stage[b];
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
    // copy chunk c (already read into its pinned buffer) to the device and run the pass kernel on it
    auto process_chunk = [&](uint64_t c) -> int {
        const int b = (int)(c % kNumBuf);
        const uint64_t s0 = c * chunk_samples;
        const uint64_t cnt = std::min(chunk_samples, ctx->n - s0);
        unsigned char *hbuf = (unsigned char *)ctx->h_stage[b];
        float *dst = to_resident ? ctx->d_iq + 2 * s0 : (float *)ctx->d_stage[b];
        if (!to_resident && c >= (uint64_t)kNumBuf)
            HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_kernel[b], 0));
        HIPCHK(ctx, hipMemcpyAsync(dst, hbuf, cnt * 8, hipMemcpyHostToDevice, ctx->copy_stream));
        HIPCHK(ctx, hipEventRecord(ctx->ev_copy[b], ctx->copy_stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_copy[b], 0));
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
    for (; submitted < std::min<uint64_t>(kReadAhead, nchunks); submitted++)
        submit_chunk(submitted);
    for (uint64_t c = 0; c < nchunks && rc == PAPR_OK; c++) {
        t_mark = now_s();
        if (ctx->pool->wait(&batches[c]))
            rc = fail(ctx, PAPR_E_IO, "read error in %s", ctx->path.c_str());
        if (timed)
            ctx->ingest.read_s += now_s() - t_mark;
        if (rc)
            break;
        t_mark = now_s();
        rc = process_chunk(c);
        if (timed)
            ctx->ingest.issue_s += now_s() - t_mark;
        // read ahead: the next unread chunk goes into the buffer used kNumBuf chunks earlier, which is
        // free once that chunk's H2D copy has completed
        if (rc == PAPR_OK && submitted < nchunks) {
            t_mark = now_s();
            if (submitted >= (uint64_t)kNumBuf &&
                hipEventSynchronize(ctx->ev_copy[submitted % kNumBuf]) != hipSuccess)
                rc = fail(ctx, PAPR_E_HIP, "hipEventSynchronize failed while recycling a staging buffer");
            if (timed)
                ctx->ingest.buffer_wait_s += now_s() - t_mark;
            if (rc == PAPR_OK)
                submit_chunk(submitted++);
        }
    }
    // on any failure let the reads already queued finish before `fs` and the batches go away
    for (uint64_t k = 0; k < submitted; k++)
        (void)ctx->pool->wait(&batches[k]);
    close_file_src(&fs);
    if (nrecords_out)
        *nrecords_out = records;
    return rc;
}

// finalize pass 1: tail + merge of `records` partials, NaN bookkeeping
int finish_stats(papr_hip_ctx *ctx, size_t records, const float *tail_ptr, uint32_t tail_samples, uint64_t tail_base,
                 papr_stats *out)
{
    papr_launch_stats_finalize(ctx->stream, tail_ptr, tail_samples, tail_base, ctx->d_partials, (uint32_t)records,
                               ctx->h_result_dev);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    partial_to_stats(*ctx->h_result, ctx->n, out);
    out->flags |= ctx->shard_flags;
    return PAPR_OK;
}

// the double sum of a resident shard came out NaN: find the first NaN power and the sign x86 gives it
int resolve_resident_nan(papr_hip_ctx *ctx, papr_stats *out)
{
    if (!std::isnan(out->sum))
        return PAPR_OK;
    unsigned long long key = ~0ull;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_nan_key, &key, 8, hipMemcpyHostToDevice, ctx->stream));
    papr_launch_first_nan(ctx->stream, 1024, ctx->d_iq, ctx->n, ctx->base, ctx->d_nan_key);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(&key, ctx->d_nan_key, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    apply_nan_key(out, key);
    return PAPR_OK;
}

}  // namespace

// =============================================================================
// C ABI
// =============================================================================
extern "C" {

int papr_hip_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        fail(nullptr, PAPR_E_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
        return e == hipErrorNoDevice ? 0 : PAPR_E_NO_DEVICE;
    }
    return n;
}

const char *papr_hip_last_error(const papr_hip_ctx *ctx)
{
    return ctx ? ctx->err : g_open_error;
}

int papr_hip_open(papr_hip_ctx **out, int device)
{
    if (!out)
        return PAPR_E_ARG;
    *out = nullptr;
    int n = papr_hip_device_count();
    if (n <= 0)
        return fail(nullptr, PAPR_E_NO_DEVICE, "no HIP device available (libpaprhip has no CPU fallback)");
    if (device < 0 || device >= n)
        return fail(nullptr, PAPR_E_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    papr_hip_ctx *ctx = new (std::nothrow) papr_hip_ctx();
    if (!ctx)
        return PAPR_E_NOMEM;
    ctx->device = device;
    auto bail = [&](int code) {
        snprintf(g_open_error, sizeof(g_open_error), "%s", ctx->err);
        papr_hip_close(ctx);
        return code;
    };
#define OPENCHK(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fail(ctx, PAPR_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_));                  \
            return bail(PAPR_E_HIP);                                                               \
        }                                                                                          \
    } while (0)
    OPENCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    OPENCHK(hipGetDeviceProperties(&prop, device));
    snprintf(ctx->name, sizeof(ctx->name), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    OPENCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    OPENCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    OPENCHK(hipHostMalloc((void **)&ctx->h_result, sizeof(papr_partial), hipHostMallocMapped));
    OPENCHK(hipHostGetDevicePointer((void **)&ctx->h_result_dev, ctx->h_result, 0));
    OPENCHK(hipMalloc((void **)&ctx->d_hist, (PAPR_HIP_MAX_LEVELS + 1) * sizeof(unsigned long long)));
    OPENCHK(hipHostMalloc((void **)&ctx->h_hist, (PAPR_HIP_MAX_LEVELS + 1) * sizeof(unsigned long long),
                          hipHostMallocDefault));
    OPENCHK(hipMalloc((void **)&ctx->d_nan_key, sizeof(unsigned long long)));
#undef OPENCHK
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
        free_b = (size_t)64 << 30;
    const int budget_mb = env_int("PAPR_HBM_BUDGET_MB", 0);
    ctx->hbm_budget = budget_mb > 0 ? (size_t)budget_mb << 20 : free_b / 10 * 9;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    parse_tune_env(&ctx->tune);

    if (ensure_partials(ctx, 4096) != PAPR_OK)
        return bail(PAPR_E_HIP);
    papr_kernels_prepare_device();  // function attributes are per device
    *out = ctx;
    return PAPR_OK;
}

void papr_hip_close(papr_hip_ctx *ctx)
{
    if (!ctx)
        return;
    if (ctx->device >= 0)
        (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    delete ctx->pool;
    release_shard(ctx);
    for (auto &t : ctx->timed) {
        (void)hipEventDestroy(t.a);
        (void)hipEventDestroy(t.b);
    }
    for (int b = 0; b < kNumBuf; b++) {
        if (ctx->h_stage[b]) (void)hipHostFree(ctx->h_stage[b]);
        if (ctx->d_stage[b]) (void)hipFree(ctx->d_stage[b]);
        if (ctx->ev_copy[b]) (void)hipEventDestroy(ctx->ev_copy[b]);
        if (ctx->ev_kernel[b]) (void)hipEventDestroy(ctx->ev_kernel[b]);
    }
    if (ctx->d_tail) (void)hipFree(ctx->d_tail);
    if (ctx->d_sweep_hist) (void)hipFree(ctx->d_sweep_hist);
    if (ctx->h_sweep_hist) (void)hipHostFree(ctx->h_sweep_hist);
    if (ctx->d_stash) (void)hipFree(ctx->d_stash);
    if (ctx->d_tile_sums) (void)hipFree(ctx->d_tile_sums);
    if (ctx->d_block_sums) (void)hipFree(ctx->d_block_sums);
    if (ctx->d_tile_E) (void)hipFree(ctx->d_tile_E);
    if (ctx->d_seg_D) (void)hipFree(ctx->d_seg_D);
    if (ctx->d_groups) (void)hipFree(ctx->d_groups);
    if (ctx->h_program) (void)hipHostFree(ctx->h_program);
    if (ctx->d_mixed_list) (void)hipFree(ctx->d_mixed_list);
    if (ctx->d_raw_list) (void)hipFree(ctx->d_raw_list);
    if (ctx->d_plan) (void)hipFree(ctx->d_plan);
    if (ctx->d_ambig) (void)hipFree(ctx->d_ambig);
    if (ctx->d_raw_store) (void)hipFree(ctx->d_raw_store);
    if (ctx->d_partials) (void)hipFree(ctx->d_partials);
    if (ctx->h_result) (void)hipHostFree(ctx->h_result);
    if (ctx->d_hist) (void)hipFree(ctx->d_hist);
    if (ctx->h_hist) (void)hipHostFree(ctx->h_hist);
    if (ctx->d_table) (void)hipFree(ctx->d_table);
    if (ctx->h_table) (void)hipHostFree(ctx->h_table);
    if (ctx->d_nan_key) (void)hipFree(ctx->d_nan_key);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    delete ctx;
}

int papr_hip_device_name(const papr_hip_ctx *ctx, char *buf, int buflen)
{
    if (!ctx || !buf || buflen <= 0)
        return PAPR_E_ARG;
    snprintf(buf, (size_t)buflen, "%s", ctx->name);
    return PAPR_OK;
}

int papr_hip_set_tuning(papr_hip_ctx *ctx, const papr_hip_tuning *t)
{
    if (!ctx || !t)
        return PAPR_E_ARG;
    int vb, vu;
    if (t->stats_blocks < 0 || t->stats_blocks > 65536 || t->ccdf_blocks < 0 || t->ccdf_blocks > 65536 ||
        t->stats_map < 0 || t->stats_map > 3 || t->ccdf_map < 0 || t->ccdf_map > 3 || t->hist_copies < 0 ||
        (t->stats_variant != 0 && papr_variant_geometry(t->stats_variant - 1, &vb, &vu) != 0) ||
        (t->ccdf_variant != 0 && papr_variant_geometry(t->ccdf_variant - 1, &vb, &vu) != 0) ||
        t->sweep_blocks < 0 || t->sweep_blocks > 65536 || t->sweep_map < 0 || t->sweep_map > 3 ||
        (t->sweep_variant != 0 && papr_sweep_variant(t->sweep_variant - 1) < 0) ||
        (t->sweep_band_log2 != 0 && (t->sweep_band_log2 < 8 || t->sweep_band_log2 > 20)) || t->estimate_ratio < 0 ||
        t->estimate_ratio > 65536)
        return fail(ctx, PAPR_E_ARG, "bad tuning values");
    ctx->tune = *t;
    return PAPR_OK;
}

int papr_hip_set_timing(papr_hip_ctx *ctx, int enabled)
{
    if (!ctx)
        return PAPR_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->timing = enabled != 0;
    ctx->timed_used = 0;
    return PAPR_OK;
}

int papr_hip_get_timing(papr_hip_ctx *ctx, papr_hip_timing *out)
{
    if (!ctx || !out)
        return PAPR_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memset(out, 0, sizeof(*out));
    for (size_t k = 0; k < ctx->timed_used; k++) {
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->timed[k].a, ctx->timed[k].b));
        if (ctx->timed[k].kind == 3) {
            out->sweep_ms += ms;
            out->sweep_launches++;
            out->sweep_bytes += ctx->timed[k].bytes;
        } else if (ctx->timed[k].kind == 4) {
            out->aux_ms += ms;
            out->aux_launches++;
            out->aux_bytes += ctx->timed[k].bytes;
        } else if (ctx->timed[k].kind == 2) {
            out->exact_ms += ms;
            out->exact_launches++;
            out->exact_bytes += ctx->timed[k].bytes;
        } else if (ctx->timed[k].kind == 0) {
            out->stats_ms += ms;
            out->stats_launches++;
            out->stats_bytes += ctx->timed[k].bytes;
        } else {
            out->ccdf_ms += ms;
            out->ccdf_launches++;
            out->ccdf_bytes += ctx->timed[k].bytes;
        }
    }
    return PAPR_OK;
}

// ---- shard residency ------------------------------------------------------------

int papr_hip_adopt(papr_hip_ctx *ctx, void *device_iq, uint64_t nsamples, uint64_t base_index)
{
    if (!ctx || (!device_iq && nsamples))
        return PAPR_E_ARG;
    if (((uintptr_t)device_iq & 15u) != 0)
        return fail(ctx, PAPR_E_ARG, "adopted device memory must be 16-byte aligned");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    release_shard(ctx);
    ctx->d_iq = (float *)device_iq;
    ctx->owns_iq = false;
    ctx->cap = ctx->n = nsamples;
    ctx->base = base_index;
    ctx->loaded = ctx->resident = true;
    return PAPR_OK;
}

int papr_hip_upload(papr_hip_ctx *ctx, const float *iq, uint64_t nsamples, uint64_t base_index)
{
    if (!ctx || (!iq && nsamples))
        return PAPR_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->owns_iq)
        release_shard(ctx);
    int rc = ensure_owned_capacity(ctx, nsamples);
    if (rc)
        return rc;
    if (nsamples)
        HIPCHK(ctx, hipMemcpy(ctx->d_iq, iq, nsamples * 8, hipMemcpyHostToDevice));
    ctx->n = nsamples;
    ctx->base = base_index;
    ctx->loaded = ctx->resident = true;
    ctx->have_file_stats = false;
    ctx->exact_valid = false;
    ctx->sweep_valid = false;
    ctx->shard_flags = 0;
    ctx->path.clear();
    return PAPR_OK;
}

int papr_hip_generate(papr_hip_ctx *ctx, const papr_synth_spec *spec, uint64_t first_index, uint64_t nsamples)
{
    if (!ctx || !spec)
        return PAPR_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!(ctx->d_iq && ctx->cap >= nsamples)) {
        int rc = ensure_owned_capacity(ctx, nsamples);
        if (rc)
            return rc;
    }
    if (nsamples) {
        const int blocks = (int)std::min<uint64_t>((nsamples + PAPR_BLOCK - 1) / PAPR_BLOCK, 8192);
        papr_launch_generate(ctx->stream, blocks, ctx->d_iq, nsamples, first_index, *spec);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->n = nsamples;
    ctx->base = first_index;
    ctx->loaded = ctx->resident = true;
    ctx->have_file_stats = false;
    ctx->exact_valid = false;
    ctx->sweep_valid = false;
    ctx->shard_flags = 0;
    ctx->path.clear();
    return PAPR_OK;
}

int papr_hip_download(papr_hip_ctx *ctx, float *iq, uint64_t first, uint64_t nsamples)
{
    if (!ctx || (!iq && nsamples))
        return PAPR_E_ARG;
    if (!ctx->loaded || !ctx->resident)
        return fail(ctx, PAPR_E_STATE, "no resident shard to download from");
    if (first + nsamples > ctx->n)
        return fail(ctx, PAPR_E_ARG, "download range past the end of the shard");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (nsamples)
        HIPCHK(ctx, hipMemcpy(iq, ctx->d_iq + 2 * first, nsamples * 8, hipMemcpyDeviceToHost));
    return PAPR_OK;
}

}  // extern "C"

namespace {

// papr_hip_load_file, optionally as a one-sweep ingest (guess != nullptr): the per-chunk kernel then also bins
// against the guessed bands and stashes, so that papr_hip_ccdf needs no second read of the shard — or of the file
int load_file_impl(papr_hip_ctx *ctx, const char *path, uint64_t first_sample, uint64_t nsamples, const float *guess,
                   int nguess)
{
    if (!ctx || !path)
        return PAPR_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const double t_begin = now_s();
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
    if (nsamples == UINT64_MAX || first_sample + nsamples > fs.nsamples)
        nsamples = fs.nsamples - first_sample;

    const bool fits = (nsamples + PAPR_TILE_SAMPLES_MAX) * 8 <= ctx->hbm_budget;
    if (!ctx->owns_iq || !fits)
        release_shard(ctx);
    if (fits) {
        rc = ensure_owned_capacity(ctx, nsamples);
        if (rc)
            return rc;
    }
    ctx->path = path;
    ctx->file_first = first_sample;
    ctx->n = nsamples;
    ctx->base = first_sample;
    ctx->resident = fits;
    ctx->loaded = true;
    ctx->have_file_stats = false;
    ctx->exact_valid = false;
    ctx->sweep_valid = false;
    ctx->shard_flags = (fs.odd && first_sample + nsamples == fs.nsamples && nsamples > 0) ? PAPR_FLAG_ODD_TAIL : 0;

    SweepRun run;
    if (guess) {
        int reason = PAPR_SWEEP_MODE;
        if (!ctx->exact && nsamples) {
            rc = ensure_ingest(ctx, !fits);  // fixes the chunk size
            if (rc == PAPR_OK)
                rc = sweep_prepare(ctx, guess, nguess, nsamples, ctx->stage_bytes / 8, &run, &reason);
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
    return PAPR_OK;
}

}  // namespace

extern "C" {

int papr_hip_load_file(papr_hip_ctx *ctx, const char *path, uint64_t first_sample, uint64_t nsamples)
{
    return load_file_impl(ctx, path, first_sample, nsamples, nullptr, 0);
}

int papr_hip_shard_fits(const papr_hip_ctx *ctx, uint64_t nsamples)
{
    if (!ctx)
        return PAPR_E_ARG;
    return (nsamples + PAPR_TILE_SAMPLES_MAX) * 8 <= ctx->hbm_budget ? 1 : 0;
}

int papr_hip_load_file_sweep(papr_hip_ctx *ctx, const char *path, uint64_t first_sample, uint64_t nsamples,
                             const float *guess_levels, int nlevels)
{
    if (nlevels < 0 || (nlevels && !guess_levels))
        return PAPR_E_ARG;
    static const float none = 0.0f;
    return load_file_impl(ctx, path, first_sample, nsamples, guess_levels ? guess_levels : &none, nlevels);
}

// Mean estimate of a file range without loading it: the same 1-in-`ratio` tile sample as papr_hip_estimate, read
// by the ingest's reader threads into the staging buffers and summed on the device.
int papr_hip_estimate_file(papr_hip_ctx *ctx, const char *path, uint64_t first_sample, uint64_t nsamples, papr_stats *est)
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
    if (nsamples == UINT64_MAX || first_sample + nsamples > fs.nsamples)
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
    std::vector<ReadBatch> batches(nbatches);
    const FileSrc *fsp = &fs;
    auto submit = [&](uint64_t bi) {
        const uint64_t g0 = bi * per_batch, g1 = std::min(ngroups, g0 + per_batch);
        unsigned char *hbuf = (unsigned char *)ctx->h_stage[bi % kNumBuf];
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
        const int b = (int)(bi % kNumBuf);
        const uint64_t cnt = std::min(ngroups, (bi + 1) * per_batch) - bi * per_batch;
        if (hipMemcpyAsync(ctx->d_stage[b], ctx->h_stage[b], cnt * kTileBytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            rc = fail(ctx, PAPR_E_HIP, "hipMemcpyAsync of the estimate sample failed");
            break;
        }
        const int blocks = (int)std::min<uint64_t>(cnt, (uint64_t)blocks_max);
        time_begin(ctx, 4, cnt * kTileBytes);
        papr_launch_estimate(ctx->stream, blocks, ctx->d_stage[b], cnt, 1, ctx->d_partials + records);
        time_end(ctx);
        records += (size_t)blocks;
        if (hipEventRecord(ctx->ev_copy[b], ctx->stream) != hipSuccess)
            rc = fail(ctx, PAPR_E_HIP, "hipEventRecord failed");
        if (rc == PAPR_OK && submitted < nbatches) {
            // the buffer about to be refilled was consumed kNumBuf batches ago
            if (submitted >= (uint64_t)kNumBuf && hipEventSynchronize(ctx->ev_copy[submitted % kNumBuf]) != hipSuccess)
                rc = fail(ctx, PAPR_E_HIP, "hipEventSynchronize failed while recycling a staging buffer");
            if (rc == PAPR_OK)
                submit(submitted++);
        }
    }
    for (uint64_t k = 0; k < submitted; k++)
        (void)ctx->pool->wait(&batches[k]);
    close_file_src(&fs);
    if (rc)
        return rc;
    papr_launch_stats_finalize(ctx->stream, nullptr, 0, 0, ctx->d_partials, (uint32_t)records, ctx->h_result_dev);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    est->sum = ctx->h_result->sum;
    est->n = ngroups * PAPR_ESTIMATE_TILE_SAMPLES;
    ctx->sweep_info.estimate_samples = est->n;
    return PAPR_OK;
}

int papr_hip_get_ingest_timing(const papr_hip_ctx *ctx, papr_hip_ingest_timing *out)
{
    if (!ctx || !out)
        return PAPR_E_ARG;
    *out = ctx->ingest;
    return PAPR_OK;
}

// ---- pass 1 ---------------------------------------------------------------------

int papr_hip_stats(papr_hip_ctx *ctx, papr_stats *out)
{
    if (!ctx || !out)
        return PAPR_E_ARG;
    if (!ctx->loaded)
        return fail(ctx, PAPR_E_STATE, "papr_hip_stats called before a shard was loaded");
    if (ctx->have_file_stats) {  // computed while the file streamed in (a sweep made during that load stays valid)
        *out = ctx->file_stats;
        return PAPR_OK;
    }
    ctx->sweep_valid = false;  // a sweep only serves the papr_hip_ccdf calls that directly follow it
    if (!ctx->resident)
        return fail(ctx, PAPR_E_STATE, "the shard is not resident and has no pass-1 result: reload it");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int nrec = 0;
    int rc = ensure_partials(ctx, (size_t)blocks_of(ctx, PASS1) + 1);
    if (rc)
        return rc;
    rc = launch_stats_range(ctx, ctx->d_iq, ctx->n, ctx->base, 0, &nrec);
    if (rc)
        return rc;
    const uint32_t tail = (uint32_t)(ctx->n % tile_samples(ctx, PASS1));
    rc = finish_stats(ctx, (size_t)nrec, ctx->d_iq + 2 * (ctx->n - tail), tail, ctx->base + ctx->n - tail, out);
    if (rc)
        return rc;
    rc = resolve_resident_nan(ctx, out);
    if (rc)
        return rc;
    ctx->exact_valid = ctx->exact;
    return PAPR_OK;
}

// ---- one-sweep mode (papr_sweep.hip) -----------------------------------------------

int papr_hip_estimate(papr_hip_ctx *ctx, papr_stats *est)
{
    if (!ctx || !est)
        return PAPR_E_ARG;
    if (!ctx->loaded)
        return fail(ctx, PAPR_E_STATE, "papr_hip_estimate called before a shard was loaded");
    papr_stats_init(est);
    ctx->sweep_info.estimate_samples = 0;
    if (ctx->have_file_stats) {  // pass 1 already ran while the file streamed in: the "estimate" is the real thing
        est->sum = ctx->file_stats.sum;
        est->n = ctx->file_stats.n;
        return PAPR_OK;
    }
    if (!ctx->resident)
        return fail(ctx, PAPR_E_STATE, "the shard is not resident and has no pass-1 result: reload it");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t ntiles = ctx->n / PAPR_ESTIMATE_TILE_SAMPLES;
    if (ntiles == 0)
        return PAPR_OK;  // nothing to sample: n = 0 tells the caller there is no estimate
    uint64_t ratio = ctx->tune.estimate_ratio > 0 ? (uint64_t)ctx->tune.estimate_ratio : (uint64_t)kEstimateRatio;
    ratio = std::max<uint64_t>(1, std::min<uint64_t>(ratio, ntiles / kEstimateMinTiles));
    const uint64_t ngroups = ntiles / ratio;
    const int blocks = (int)std::min<uint64_t>(ngroups, (uint64_t)ctx->num_cus * 8);
    int rc = ensure_partials(ctx, (size_t)blocks + 1);
    if (rc)
        return rc;
    time_begin(ctx, 4, ngroups * PAPR_ESTIMATE_TILE_SAMPLES * 8);
    papr_launch_estimate(ctx->stream, blocks, ctx->d_iq, ngroups, (uint32_t)ratio, ctx->d_partials);
    time_end(ctx);
    HIPCHK(ctx, hipGetLastError());
    papr_launch_stats_finalize(ctx->stream, nullptr, 0, 0, ctx->d_partials, (uint32_t)blocks, ctx->h_result_dev);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    est->sum = ctx->h_result->sum;
    est->n = ngroups * PAPR_ESTIMATE_TILE_SAMPLES;
    ctx->sweep_info.estimate_samples = est->n;
    return PAPR_OK;
}

int papr_hip_stats_sweep(papr_hip_ctx *ctx, const float *guess_levels, int nlevels, papr_stats *out)
{
    if (!ctx || !out || nlevels < 0 || (nlevels && !guess_levels))
        return PAPR_E_ARG;
    if (!ctx->loaded)
        return fail(ctx, PAPR_E_STATE, "papr_hip_stats_sweep called before a shard was loaded");
    papr_hip_sweep_info &info = ctx->sweep_info;
    info.swept = info.resolved = 0;
    info.stash_samples = 0;
    ctx->sweep_valid = false;
    auto plain = [&](int reason) {
        info.reason = reason;
        return papr_hip_stats(ctx, out);
    };
    if (ctx->have_file_stats || !ctx->resident || ctx->exact)
        return plain(PAPR_SWEEP_MODE);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    SweepRun run;
    int reason = PAPR_SWEEP_OK;
    int rc = sweep_prepare(ctx, guess_levels, nlevels, ctx->n, ctx->n, &run, &reason);
    if (rc)
        return rc;
    if (reason != PAPR_SWEEP_OK)
        return plain(reason);
    int nrec = 0;
    rc = sweep_launch(ctx, run, ctx->d_iq, ctx->n, ctx->base, 0, &nrec);
    if (rc)
        return rc;
    rc = sweep_fetch(ctx, run);
    if (rc)
        return rc;
    const uint32_t tail = (uint32_t)(ctx->n % run.tile);
    rc = finish_stats(ctx, (size_t)nrec, ctx->d_iq + 2 * (ctx->n - tail), tail, ctx->base + ctx->n - tail, out);  // synchronises
    if (rc)
        return rc;
    if (std::isnan(out->sum))  // NaN in the data: the sweep's integer-max trackers do not apply (papr_sweep.hip)
        return plain(PAPR_SWEEP_NO_BANDS);
    return sweep_collect(ctx, run);
}

int papr_hip_get_sweep_info(const papr_hip_ctx *ctx, papr_hip_sweep_info *out)
{
    if (!ctx || !out)
        return PAPR_E_ARG;
    *out = ctx->sweep_info;
    return PAPR_OK;
}

// ---- bit-exact mean ---------------------------------------------------------------

int papr_hip_set_exact(papr_hip_ctx *ctx, int enabled)
{
    if (!ctx)
        return PAPR_E_ARG;
    if ((enabled != 0) != ctx->exact) {
        ctx->exact = enabled != 0;
        ctx->exact_valid = false;
        if (ctx->resident)
            ctx->have_file_stats = false;  // pass 1 is re-run over the resident shard in the other mode
    }
    return PAPR_OK;
}

}  // extern "C"

namespace {

int exact_preconditions(papr_hip_ctx *ctx, double before, bool allow_restreamed)
{
    if (!ctx->exact || !ctx->exact_valid || !ctx->loaded)
        return fail(ctx, PAPR_E_STATE, "exact sum needs papr_hip_set_exact(1) and papr_hip_stats on the current shard first");
    if (!ctx->resident && !allow_restreamed)
        return fail(ctx, PAPR_E_STATE, "papr_hip_exact_program needs a shard that is resident in HBM "
                                       "(re-streamed shards: papr_hip_ccdf_exact)");
    if (!(before >= 0.0) || !std::isfinite(before))
        return fail(ctx, PAPR_E_ARG, "`before` must be a finite, non-negative sum");
    return PAPR_OK;
}

int reserve_program(papr_hip_ctx *ctx, size_t want)  // grow the pinned program buffer, keeping its contents
{
    if (want <= ctx->h_program_cap)
        return PAPR_OK;
    unsigned char *fresh = nullptr;
    const size_t cap = std::max<size_t>(want + want / 4, (size_t)1 << 20);
    HIPCHK(ctx, hipHostMalloc((void **)&fresh, cap, hipHostMallocMapped));
    if (ctx->h_program) {
        memcpy(fresh, ctx->h_program, ctx->h_program_cap);
        (void)hipHostFree(ctx->h_program);
    }
    ctx->h_program = fresh;
    ctx->h_program_cap = cap;
    return PAPR_OK;
}

// Device side of the exact sum: classify tiles, one sweep over the samples for the per-segment
// rounding functions (with pass 2 fused in when `fused` is given), pre-compose the groups and gather
// the sum program into mapped host memory — one stream synchronisation in total.  *bytes = 0 means
// the device-side gather overflowed its lists and the caller has to assemble the program itself.
int run_exact_device(papr_hip_ctx *ctx, double before, uint64_t n_total, const CcdfPlan *fused, size_t *bytes)
{
    const uint64_t ntiles = ctx->n / PAPR_EXACT_TILE_SAMPLES;
    const uint64_t ngroups = (ntiles + PAPR_EXACT_GROUP_TILES - 1) / PAPR_EXACT_GROUP_TILES;
    const uint32_t tail = (uint32_t)(ctx->n - ntiles * PAPR_EXACT_TILE_SAMPLES);
    // margin >= the worst-case relative drift of a sequential double sum of n_total non-negative terms
    const double delta = std::max(1.0e-6, 8.0 * (double)std::max<uint64_t>(n_total, ctx->n) * 1.1102230246251565e-16);
    *bytes = 0;
    int rc = ensure_exact_buffers(ctx);
    if (rc)
        return rc;
    if (!ctx->d_plan) {
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_mixed_list, kCapMixed * sizeof(uint32_t)));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_raw_list, kCapRaw * sizeof(uint32_t)));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_plan, sizeof(papr_exact_plan)));
    }
    rc = reserve_program(ctx, sizeof(papr_exact_header) + ngroups * sizeof(papr_exact_group_rec) +
                                  (size_t)kCapMixed * sizeof(papr_exact_mixed_rec) +
                                  (size_t)kCapRaw * sizeof(papr_exact_raw_rec) + (size_t)tail * 8);
    if (rc)
        return rc;
    unsigned char *program_dev = nullptr;
    HIPCHK(ctx, hipHostGetDevicePointer((void **)&program_dev, ctx->h_program, 0));
    const uint64_t nsegs = 2 * ntiles;
    if (ctx->resident) {
        // plain sweep: 4-wave workgroups, 2 per CU; fused sweep: 8-wave workgroups, 2 per CU (16 waves share the LDS tables)
        const int per_cu = std::max(1, env_int("PAPR_EXACT_WG_PER_CU", 2));
        const uint64_t wg_waves = fused ? (uint64_t)papr_exact_fused_waves() : 4;
        const int blocks = (int)std::max<uint64_t>(
            1, std::min<uint64_t>((nsegs + wg_waves - 1) / wg_waves, (uint64_t)ctx->num_cus * per_cu));
        time_begin(ctx, 2, ctx->n * 8);
        papr_launch_exact_classify(ctx->stream, ctx->d_tile_sums, ntiles, ctx->d_block_sums, before, delta, ctx->d_tile_E,
                                   nullptr, 0, nullptr, nullptr);
        if (fused)
            papr_launch_exact_segments_ccdf(ctx->stream, blocks, ctx->d_iq, nsegs, ctx->d_tile_E, ctx->d_seg_D,
                                            ctx->d_iq + 2 * ntiles * PAPR_EXACT_TILE_SAMPLES, tail, ctx->d_table,
                                            fused->P, fused->lds_bytes, ctx->d_hist);
        else
            papr_launch_exact_segments(ctx->stream, blocks, ctx->d_iq, nsegs, ctx->d_tile_E, ctx->d_seg_D);
        papr_launch_exact_groups(ctx->stream, ctx->d_tile_E, ntiles, ctx->d_seg_D, ngroups, ctx->d_groups);
        time_end(ctx);
        papr_launch_exact_pack(ctx->stream, ctx->d_groups, ngroups, ctx->d_tile_E, ntiles, ctx->d_seg_D, ctx->d_iq, nullptr,
                               ctx->d_iq + 2 * ntiles * PAPR_EXACT_TILE_SAMPLES, ctx->n, tail, ctx->d_mixed_list,
                               kCapMixed, ctx->d_raw_list, kCapRaw, ctx->d_plan, program_dev);
    } else {
        // re-streamed shard: classify (also listing the unprovable tiles), then the file goes through the
        // staging buffers once more with the fused sweep on every chunk
        if (!fused)
            return fail(ctx, PAPR_E_STATE, "a re-streamed shard builds its sum program in the pass-2 sweep only");
        if (!ctx->d_ambig) {
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_ambig, (2 * kCapRaw + 1) * sizeof(uint32_t)));
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_raw_store, (size_t)kCapRaw * PAPR_EXACT_TILE_SAMPLES * 8));
        }
        HIPCHK(ctx, hipMemsetAsync(ctx->d_ambig + 2 * kCapRaw, 0, sizeof(uint32_t), ctx->stream));
        papr_launch_exact_classify(ctx->stream, ctx->d_tile_sums, ntiles, ctx->d_block_sums, before, delta, ctx->d_tile_E,
                                   ctx->d_ambig, kCapRaw, ctx->d_ambig + 2 * kCapRaw, ctx->d_ambig + kCapRaw);
        HIPCHK(ctx, hipGetLastError());
        rc = stream_file(ctx, PASS_STREAM_CCDF_EXACT, fused, nullptr);
        if (rc)
            return rc;
        papr_launch_exact_groups(ctx->stream, ctx->d_tile_E, ntiles, ctx->d_seg_D, ngroups, ctx->d_groups);
        papr_launch_exact_pack(ctx->stream, ctx->d_groups, ngroups, ctx->d_tile_E, ntiles, ctx->d_seg_D, nullptr,
                               ctx->d_raw_store, ctx->d_tail, ctx->n, tail, ctx->d_mixed_list, kCapMixed, ctx->d_raw_list,
                               kCapRaw, ctx->d_plan, program_dev);
    }
    HIPCHK(ctx, hipGetLastError());
    if (fused)
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_hist, ctx->d_hist, (size_t)(fused->P.nkeys + 1) * sizeof(unsigned long long),
                                   hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    papr_exact_header h;
    memcpy(&h, ctx->h_program, sizeof(h));
    if (h.magic == PAPR_EXACT_MAGIC && h.reserved == 0 && h.ngroups == ngroups && h.nsamples == ctx->n &&
        !env_int("PAPR_EXACT_HOST_ASSEMBLY", 0))  // (the env switch lets the tests exercise the fallback)
        *bytes = sizeof(h) + ngroups * sizeof(papr_exact_group_rec) + (size_t)h.nmixed * sizeof(papr_exact_mixed_rec) +
                 (size_t)h.nraw * sizeof(papr_exact_raw_rec) + (size_t)tail * 8;
    return PAPR_OK;
}

// Host-driven assembly, for the (never yet seen) case that a shard has more mixed groups / raw tiles
// than the device-side lists hold.
int assemble_program_on_host(papr_hip_ctx *ctx, const void **program, size_t *bytes);

int assemble_program_on_host(papr_hip_ctx *ctx, const void **program, size_t *bytes)
{
    const uint64_t ntiles = ctx->n / PAPR_EXACT_TILE_SAMPLES;
    const uint64_t ngroups = (ntiles + PAPR_EXACT_GROUP_TILES - 1) / PAPR_EXACT_GROUP_TILES;
    const uint32_t tail = (uint32_t)(ctx->n - ntiles * PAPR_EXACT_TILE_SAMPLES);
    std::vector<papr_exact_group> groups(ngroups);
    if (ngroups) {
        HIPCHK(ctx, hipMemcpyAsync(groups.data(), ctx->d_groups, ngroups * sizeof(papr_exact_group),
                                   hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    // what has to travel in detail: mixed groups (per-tile classes + per-segment pairs) ...
    std::vector<uint64_t> mixed;
    for (uint64_t g = 0; g < ngroups; g++)
        if (groups[g].E == PAPR_EXACT_AMBIG)
            mixed.push_back(g);
    auto reserve = [&](size_t want) -> int { return reserve_program(ctx, want); };
    static_assert(sizeof(papr_exact_group) == sizeof(papr_exact_group_rec), "group record layout");
    const size_t off_groups = sizeof(papr_exact_header);
    const size_t off_mixed = off_groups + ngroups * sizeof(papr_exact_group_rec);
    const size_t off_raw = off_mixed + mixed.size() * sizeof(papr_exact_mixed_rec);
    int rc = reserve(off_raw + (size_t)tail * 8);
    if (rc)
        return rc;
    if (ngroups)
        memcpy(ctx->h_program + off_groups, groups.data(), ngroups * sizeof(papr_exact_group_rec));
    for (size_t k = 0; k < mixed.size(); k++) {
        const uint64_t g = mixed[k];
        papr_exact_mixed_rec *m = (papr_exact_mixed_rec *)(ctx->h_program + off_mixed + k * sizeof(papr_exact_mixed_rec));
        const uint64_t t0 = g * PAPR_EXACT_GROUP_TILES, t1 = std::min<uint64_t>(t0 + PAPR_EXACT_GROUP_TILES, ntiles);
        m->group = g;
        for (uint64_t j = t1 - t0; j < PAPR_EXACT_GROUP_TILES; j++)
            m->tile_E[j] = PAPR_EXACT_ZERO;
        memset(m->seg_D, 0, sizeof(m->seg_D));
        HIPCHK(ctx, hipMemcpyAsync(m->tile_E, ctx->d_tile_E + t0, (t1 - t0) * sizeof(int32_t), hipMemcpyDeviceToHost,
                                   ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(m->seg_D, ctx->d_seg_D + 4 * t0, (t1 - t0) * 2 * 2 * sizeof(double),
                                   hipMemcpyDeviceToHost, ctx->stream));
    }
    if (!mixed.empty())
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // ... and their tiles that are not provably inside one binade (raw samples)
    std::vector<uint64_t> raw;
    for (size_t k = 0; k < mixed.size(); k++) {
        const papr_exact_mixed_rec *m =
            (const papr_exact_mixed_rec *)(ctx->h_program + off_mixed + k * sizeof(papr_exact_mixed_rec));
        for (uint64_t j = 0; j < PAPR_EXACT_GROUP_TILES; j++)
            if (m->tile_E[j] == PAPR_EXACT_AMBIG)
                raw.push_back(mixed[k] * PAPR_EXACT_GROUP_TILES + j);
    }
    const size_t off_tail = off_raw + raw.size() * sizeof(papr_exact_raw_rec);
    const size_t total = off_tail + (size_t)tail * 8;
    rc = reserve(total);
    if (rc)
        return rc;
    for (size_t k = 0; k < raw.size(); k++) {
        papr_exact_raw_rec *r = (papr_exact_raw_rec *)(ctx->h_program + off_raw + k * sizeof(papr_exact_raw_rec));
        r->tile = raw[k];
        HIPCHK(ctx, hipMemcpyAsync(r->iq, ctx->d_iq + 2 * raw[k] * PAPR_EXACT_TILE_SAMPLES, PAPR_EXACT_TILE_SAMPLES * 8,
                                   hipMemcpyDeviceToHost, ctx->stream));
    }
    if (tail)
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_program + off_tail, ctx->d_iq + 2 * ntiles * PAPR_EXACT_TILE_SAMPLES,
                                   (size_t)tail * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (!raw.empty() || tail)
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    papr_exact_header h;
    memset(&h, 0, sizeof(h));
    h.magic = PAPR_EXACT_MAGIC;
    h.version = PAPR_EXACT_VERSION;
    h.nsamples = ctx->n;
    h.ntiles = ntiles;
    h.ngroups = ngroups;
    h.tail_samples = tail;
    h.nmixed = (uint32_t)mixed.size();
    h.nraw = (uint32_t)raw.size();
    memcpy(ctx->h_program, &h, sizeof(h));
    *program = ctx->h_program;
    *bytes = total;
    return PAPR_OK;
}

void counts_from_histogram(const papr_hip_ctx *ctx, const CcdfPlan &plan, int nlevels, uint64_t *counts_above)
{
    // samples above unique key i = everything binned at i + 1 or higher
    const uint32_t m = plan.P.nkeys;
    std::vector<uint64_t> above(m);
    uint64_t run = 0;
    for (uint32_t i = m; i-- > 0;) {
        run += ctx->h_hist[i + 1];
        above[i] = run;
    }
    for (int j = 0; j < nlevels; j++)
        counts_above[j] = plan.pos[j] >= 0 ? above[plan.pos[j]] : 0;
}

// Answer papr_hip_ccdf from the last one-sweep pass if every true threshold lies inside the band of its guess:
// samples outside the bands were decided by the sweep, the stash holds the rest.
int resolve_from_sweep(papr_hip_ctx *ctx, const CcdfPlan &plan, const float *levels, int nlevels, uint64_t *counts_above,
                       bool *done)
{
    papr_hip_sweep_info &info = ctx->sweep_info;
    *done = false;
    info.resolved = 0;
    // every true threshold must lie inside one of the bands (papr_sweep_resolve, first without stash counts: a dry run)
    const int band_log2 = __builtin_ctz(ctx->sweep_half);  // of the sweep that left this state behind
    std::vector<uint64_t> stash_above((size_t)nlevels, 0);
    if (!papr_sweep_resolve(ctx->sweep_keys.data(), (int)ctx->sweep_keys.size(), band_log2, ctx->sweep_even_above.data(),
                            levels, nlevels, stash_above.data(), counts_above)) {
        info.reason = PAPR_SWEEP_OUT_OF_BAND;
        return PAPR_OK;
    }
    if (ctx->sweep_overflow) {
        info.reason = PAPR_SWEEP_STASH_FULL;
        return PAPR_OK;
    }
    const uint32_t m = plan.P.nkeys;
    if (ctx->sweep_stash_count) {
        int rc = upload_ccdf_table(ctx, plan);
        if (rc)
            return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_hist, 0, (size_t)(m + 1) * sizeof(unsigned long long), ctx->stream));
        time_begin(ctx, 4, ctx->sweep_stash_count * 4);
        // enough workgroups to fill the chip: every segment is split over `split` of them
        const uint32_t split = std::max<uint32_t>(1, (uint32_t)(ctx->num_cus * 8) / ctx->sweep_nsegs);
        papr_launch_ccdf_power(ctx->stream, (int)(ctx->sweep_nsegs * split), plan.lut, plan.lds_bytes, ctx->d_stash,
                               ctx->d_sweep_hist + ctx->sweep_nbins, ctx->sweep_seg_cap, ctx->sweep_nsegs, split,
                               ctx->d_table, plan.P, ctx->d_hist);
        time_end(ctx);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_hist, ctx->d_hist, (size_t)(m + 1) * sizeof(unsigned long long),
                                   hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    } else {
        memset(ctx->h_hist, 0, (size_t)(m + 1) * sizeof(unsigned long long));
    }
    counts_from_histogram(ctx, plan, nlevels, stash_above.data());  // stash powers above each level ...
    (void)papr_sweep_resolve(ctx->sweep_keys.data(), (int)ctx->sweep_keys.size(), band_log2, ctx->sweep_even_above.data(),
                             levels, nlevels, stash_above.data(), counts_above);  // ... + everything above its band
    info.resolved = 1;
    info.reason = PAPR_SWEEP_OK;
    *done = true;
    return PAPR_OK;
}

}  // namespace

extern "C" {

int papr_hip_exact_program(papr_hip_ctx *ctx, double before, uint64_t n_total, const void **program, size_t *bytes)
{
    if (!ctx || !program || !bytes)
        return PAPR_E_ARG;
    int rc = exact_preconditions(ctx, before, false);
    if (rc)
        return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    rc = run_exact_device(ctx, before, n_total, nullptr, bytes);
    if (rc)
        return rc;
    *program = ctx->h_program;
    return *bytes ? PAPR_OK : assemble_program_on_host(ctx, program, bytes);
}

int papr_hip_ccdf_exact(papr_hip_ctx *ctx, const float *levels, int nlevels, uint64_t *counts_above, double before,
                        uint64_t n_total, const void **program, size_t *bytes)
{
    if (!ctx || !program || !bytes || nlevels < 0 || (nlevels && (!levels || !counts_above)))
        return PAPR_E_ARG;
    if (nlevels > PAPR_HIP_MAX_LEVELS)
        return fail(ctx, PAPR_E_LIMIT, "%d levels exceeds PAPR_HIP_MAX_LEVELS (%d)", nlevels, PAPR_HIP_MAX_LEVELS);
    int rc = exact_preconditions(ctx, before, true);
    if (rc)
        return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CcdfPlan plan;
    if (nlevels) {
        rc = plan_ccdf(ctx, levels, nlevels, &plan);
        if (rc)
            return rc;
    }
    // the fused sweep holds the LUT form of the table plus the 36 KiB transpose buffer in LDS
    size_t fused_lds = 0;
    if (nlevels && plan.lut && plan.P.nkeys) {
        plan.P.copies = std::min<uint32_t>(plan.P.copies, 4);
        fused_lds = (size_t)plan.P.table_words * 4 + (size_t)plan.P.copies * (plan.P.nkeys + 1) * 4;
        while (plan.P.copies > 1 && fused_lds > 12 * 1024) {  // two workgroups per CU: 2 x (64 KiB + this) <= 160 KiB
            plan.P.copies--;
            fused_lds = (size_t)plan.P.table_words * 4 + (size_t)plan.P.copies * (plan.P.nkeys + 1) * 4;
        }
        plan.lds_bytes = fused_lds;
    }
    const bool fuse = fused_lds != 0 && fused_lds + papr_exact_transpose_lds_bytes() <= 150 * 1024 && ctx->n > 0;
    if (!fuse) {  // unusual level table: the two sweeps run one after the other
        rc = papr_hip_ccdf(ctx, levels, nlevels, counts_above);
        if (rc)
            return rc;
        return papr_hip_exact_program(ctx, before, n_total, program, bytes);
    }
    rc = upload_ccdf_table(ctx, plan);
    if (rc)
        return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_hist, 0, (size_t)(plan.P.nkeys + 1) * sizeof(unsigned long long), ctx->stream));
    rc = run_exact_device(ctx, before, n_total, &plan, bytes);
    if (rc)
        return rc;
    counts_from_histogram(ctx, plan, nlevels, counts_above);
    *program = ctx->h_program;
    if (*bytes)
        return PAPR_OK;
    if (!ctx->resident)  // the raw tiles of a re-streamed shard are gone once their chunk has left the device
        return fail(ctx, PAPR_E_LIMIT, "too many binade crossings for the device-side program lists");
    return assemble_program_on_host(ctx, program, bytes);
}

// ---- pass 2 ---------------------------------------------------------------------

int papr_hip_ccdf(papr_hip_ctx *ctx, const float *levels, int nlevels, uint64_t *counts_above)
{
    if (!ctx || nlevels < 0 || (nlevels && (!levels || !counts_above)))
        return PAPR_E_ARG;
    if (!ctx->loaded)
        return fail(ctx, PAPR_E_STATE, "papr_hip_ccdf called before a shard was loaded");
    if (nlevels > PAPR_HIP_MAX_LEVELS)
        return fail(ctx, PAPR_E_LIMIT, "%d levels exceeds PAPR_HIP_MAX_LEVELS (%d)", nlevels, PAPR_HIP_MAX_LEVELS);
    ctx->sweep_info.resolved = 0;
    if (nlevels == 0)
        return PAPR_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CcdfPlan plan;
    int rc = plan_ccdf(ctx, levels, nlevels, &plan);
    if (rc)
        return rc;
    const uint32_t m = plan.P.nkeys;
    if (m == 0 || ctx->n == 0) {
        for (int j = 0; j < nlevels; j++)
            counts_above[j] = 0;
        return PAPR_OK;
    }
    if (ctx->sweep_valid) {  // the one-sweep pass already decided everything outside the bands
        bool done = false;
        rc = resolve_from_sweep(ctx, plan, levels, nlevels, counts_above, &done);
        if (rc || done)
            return rc;
    }
    rc = upload_ccdf_table(ctx, plan);
    if (rc)
        return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_hist, 0, (size_t)(m + 1) * sizeof(unsigned long long), ctx->stream));
    if (ctx->resident)
        rc = launch_ccdf_range(ctx, plan, ctx->d_iq, ctx->n);
    else
        rc = stream_file(ctx, PASS_STREAM_CCDF, &plan, nullptr);
    if (rc)
        return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_hist, ctx->d_hist, (size_t)(m + 1) * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    counts_from_histogram(ctx, plan, nlevels, counts_above);
    return PAPR_OK;
}

}  // extern "C"
