// papr_runtime_internal.h — shared by the translation units of the host runtime behind include/papr_hip.h
// (papr_runtime.cpp: contexts, shards, the two passes; papr_ingest.cpp: file -> pinned host -> HBM;
// papr_sweep_rt.cpp: the one-sweep mode; papr_exact_rt.cpp: the bit-exact sequential sum).  Internal: not part of the C ABI.
#ifndef PAPR_RUNTIME_INTERNAL_H
#define PAPR_RUNTIME_INTERNAL_H

#include "papr_hip.h"
#include "papr_exchange.h"
#include "papr_hip_measure.h"
#include "papr_exact_format.h"
#include "papr_kernels.h"
#include "papr_readbatch.h"

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
#include <ctime>
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

namespace papr_rt {

constexpr uint64_t kChunkAlign = PAPR_TILE_SAMPLES_MAX;  // chunk boundaries stay tile aligned for every variant
constexpr int kMaxBuf = 16;     // pinned staging buffers: capacity (ctx->num_buf of them are used: PAPR_STAGE_BUFS)
constexpr int kNumBufDefault = 4;
constexpr int kReadAheadDefault = 2;   // chunks being read ahead of the one being copied (ctx->read_ahead: PAPR_READ_AHEAD)
constexpr int kMaxTimed = 4096;

// ---- a tiny pool of file-reader threads -------------------------------------
// Jobs are grouped in batches (one batch = the slices of one chunk); the
// submitter can queue the next chunk's batch before waiting for the current
// one, so the readers never go idle between chunks.
// (struct ReadBatch {pending, error}: papr_readbatch.h, shared with papr_uring.h)

// CPUs of the NUMA node the GPU hangs off (its PCIe root): the ingest's reader threads run there and the pinned
// staging buffers are first touched there, so that the H2D DMA never crosses the socket interconnect.
// Empty set = unknown / single node / PAPR_NUMA=0.
struct CpuSet {
    cpu_set_t set;
    bool valid = false;
};

class UringReader;  // papr_uring.h

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
    int kind;  // 0 stats, 1 ccdf, 2 exact-sum kernels, 3 one-sweep kernel, 4 estimate / stash recount, 5 exact-sum helpers
               // (4 and 5: only at papr_hip_set_timing level 1)
    uint64_t bytes;
};

struct SweepRun;

}  // namespace papr_rt

// PAPR_HOST_TRACE=1: wall-clock marks of one papr_hip_analyze call, printed to stderr when it returns (where the host's
// share of a step goes; a measurement aid, off by default)
struct HostTrace {
    int on = -1, n = 0;
    const char *what[32];
    double t[32];
    void mark(const char *w)
    {
        if (on < 0)
            on = getenv("PAPR_HOST_TRACE") && atoi(getenv("PAPR_HOST_TRACE")) ? 1 : 0;
        if (!on || n >= 32)
            return;
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        what[n] = w;
        t[n++] = ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
    }
    void dump()
    {
        if (on == 1 && n) {
            char line[2048];
            int k = snprintf(line, sizeof(line), "papr host trace (us since entry):");
            for (int i = 0; i < n && k < (int)sizeof(line) - 64; i++)
                k += snprintf(line + k, sizeof(line) - k, " %s %.1f", what[i], t[i] - t[0]);
            fprintf(stderr, "%s\n", line);
        }
        n = 0;
    }
};

extern "C" const double *papr_level_pow_table(int graph, int *count);

struct papr_hip_ctx {
    HostTrace trace;
    int device = -1;
    hipStream_t stream = nullptr;     // compute
    hipStream_t copy_stream = nullptr;
    hipStream_t copy_stream2 = nullptr;  // chunks alternate between the two: two DMA engines on the link (PAPR_COPY_STREAMS)
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
    int num_buf = 0, read_ahead = 0;       // (set by ensure_ingest)
    void *h_stage[papr_rt::kMaxBuf] = {};
    void *h_stage_dev[papr_rt::kMaxBuf] = {};  // the same buffers as the device sees them (mapped); null: not mapped
    int h2d_pull = -1;                         // the H2D leg by papr_pull_kernel (1) or by hipMemcpyAsync (0): PAPR_H2D
    void *d_stage[papr_rt::kMaxBuf] = {};
    hipEvent_t ev_copy[papr_rt::kMaxBuf] = {};
    hipEvent_t ev_kernel[papr_rt::kMaxBuf] = {};
    // exact-sum one-read step: the sum program is complete (ev_program) before the stash recount has run; work the caller
    // wants done in that window (papr_hip_analyze: the exchange / replay of the programs) — run once, by the next call
    // that is about to wait for the stream (run_overlap_work), then cleared
    hipEvent_t ev_program = nullptr;
    bool program_pending = false;        // ev_program was recorded for the current step
    bool exact_program_launched = false; // stats_sweep_fused already ran run_exact_swept for the current sweep ...
    double exact_program_before = 0.0;   // ... with this sum in front of the shard (0 without peers)
    // peers, single-wait step: every rank's program slot after the in-stream all-gather (device; mapped host)
    unsigned char *d_xprog = nullptr, *d_xprog_all = nullptr, *h_xprog_all = nullptr, *h_xprog_all_dev = nullptr;
    size_t xprog_slot = 0;               // bytes of a slot to begin with (every rank sizes it from the same numbers)
    std::vector<size_t> xprog_sizes, xprog_offs;  // per rank: its slot and where it starts — the same on every rank: a rank
                                         // whose program outgrew its slot is seen by all (the headers), and all enlarge it
    size_t xprog_cap_mine = 0, xprog_cap_all = 0;  // what d_xprog / d_xprog_all, h_xprog_all hold
    int xprog_world = 0;                 // slots in d_xprog_all / h_xprog_all
    bool xprog_ready = false;            // h_xprog_all holds the current step's programs (behind ev_program)
    const unsigned char *program_view = nullptr;  // where the current step's own program is, if not in h_program
    std::function<void()> overlap_work;
    size_t stage_bytes = 0;
    papr_rt::ReaderPool *pool = nullptr;
    papr_rt::UringReader *uring = nullptr;  // papr_uring.h: the reader for O_DIRECT streams (nullptr: not tried / not offered)
    bool uring_tried = false;
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
    unsigned long long *d_sweep_hist = nullptr;  // 2 L + 2 bins, then one stash-segment length per workgroup (v2: slots, then powers)
    unsigned long long *h_sweep_hist = nullptr;  // pinned
    float *d_stash = nullptr;                    // in-band powers of the last sweep
    uint64_t stash_cap = 0;
    bool sweep_valid = false;                    // the fields below describe the CURRENT shard
    uint32_t sweep_half = 0;                     // half-width of a band, in bit patterns
    std::vector<uint32_t> sweep_keys;            // unique guessed keys (band centres), ascending
    std::vector<uint64_t> sweep_even_above;      // per guessed key j: samples in even bins >= 2 j + 2
    uint64_t sweep_stash_count = 0;
    uint64_t sweep_seg_cap = 0;                  // floats per stash segment
    uint32_t sweep_nsegs = 0, sweep_nbins = 0, sweep_seg_off = 0;
    int xcd_first = -1;                  // XCD of the stream's workgroup 0 (probed once, then from every sweep's record); -1: not known
    uint32_t sweep_blocks_last = 0;      // workgroups of the last sweep launch (their records lead d_partials)
    papr_guess_out *d_guess = nullptr, *h_guess = nullptr, *h_guess_dev = nullptr;  // papr_guess_bands_kernel's output (device; mapped host)
    double *d_pow_tab = nullptr;  // [2][PAPR_POW_TABLE]: the host libm's pow(10, x_j) of the two level tables (papr_host.c)
    papr_true_out *d_true = nullptr, *h_true = nullptr, *h_true_dev = nullptr;     // papr_true_table_kernel's output
    papr_partial *d_result_copy = nullptr;  // the finalize kernel's record once more, for papr_true_table_kernel
    unsigned long long *h_sweep_hist_dev = nullptr;  // device address of h_sweep_hist
    bool spec_recount_valid = false;  // h_hist holds the stash recount for the table in h_true (stats_sweep_fused)
    bool sweep_overflow = false;
    // the single-wait step with peers (stats_sweep_fused over an in-stream exchange): what the collectives left behind
    unsigned char *d_peer = nullptr;             // device scratch: estimate records, pass-1 records, the all-reduce vector
    size_t peer_cap = 0;
    papr_peer_out *h_peer = nullptr, *h_peer_dev = nullptr;  // mapped: the merged record, `before`, the file's length
    unsigned long long *h_xvec = nullptr;        // pinned: the all-reduced [sweep bins | recount bins | flags]
    bool peer_global = false;                    // sweep_even_above_global / recount_global hold the FILE's numbers
    bool counts_global = false;                  // the last papr_hip_ccdf answered with the file's counts (no exchange needed)
    std::vector<uint64_t> sweep_even_above_global;
    std::vector<unsigned long long> recount_global;
    uint64_t peer_epoch = 0;                     // bumped by every call that changes shard state or mode (see stats_sweep_fused)
    uint64_t peer_agreed_key = 0;                // the shard state for which the ranks agreed ...
    bool peer_agreed_ok = false;                 // ... that every one of them can take the single-wait step
    papr_hip_sweep_info sweep_info{};
    const papr_rt::SweepRun *ingest_run = nullptr;        // set while papr_hip_load_file_sweep streams the file in
    int32_t *d_tile_E_spec = nullptr;            // exact one-read sweep: speculated binade per tile (same capacity as d_tile_E)
    double *d_est_sq = nullptr, *h_est_sq = nullptr; // papr_hip_estimate: per-workgroup sums of squared piece sums (-> standard error)
    double *d_est_groups = nullptr;              // papr_hip_estimate: sampled sum per estimate group (exact one-read sweep)
    uint64_t est_groups_cap = 0, est_ngroups = 0, est_ratio = 0;
    bool est_groups_valid = false;               // ... describe the CURRENT shard
    bool est_file_valid = false;                 // ... came from papr_hip_estimate_file for this file range (exact mode):
    uint64_t est_file_first = 0, est_file_n = 0; //     papr_hip_load_file_sweep of the same range may speculate from them
    float *d_redo_store = nullptr;               // streamed shards: the tiles to redo, read back from the file
    size_t redo_store_tiles = 0;
    int band_hint = 0;                           // papr_hip_set_band: half-width (log2) for the next sweeps, 0 = default
    double exact_before_hint = 0.0;              // estimated sum of everything before this shard (papr_hip_set_exact_hint)
    uint32_t *h_redo_count = nullptr, *h_redo_count_dev = nullptr;  // pinned (and its device address)
    uint32_t *d_redo = nullptr;                  // [0, kCapRedo) tiles whose pairs must be rebuilt, [kCapRedo] their count
    bool exact_swept = false;                    // the last sweep left speculated pairs in d_seg_D / d_tile_E_spec

    papr_hip_ingest_timing ingest{};
    papr_hip_tuning tune{};
    bool timing = false, timing_aux = false;  // (timing_aux: also the small estimate / recount kernels, kind 4)
    std::vector<papr_rt::TimedLaunch> timed;
    size_t timed_used = 0;
    bool time_skipped = false;  // the bracket that is open is not being timed
};

namespace papr_rt {

extern char g_open_error[256];

#define HIPCHK(ctx, call)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, PAPR_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_));      \
    } while (0)

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
constexpr int kSweepVariant = PAPR_SWEEP_VARIANT, kSweepPerCU = 4, kSweepMap = PAPR_MAP_GRID_STRIDE;  // (papr_sweep_kernel: one workgroup per CU)
constexpr int kStashSkewFloats = 0;  // (see stash_segment_floats; 1 KiB units)
constexpr int kSweepExactVariant = PAPR_SWEEP3_VARIANT;  // papr_sweep3_kernel

constexpr int kSweepBandLog2 = 14, kEstimateRatio = 64;

constexpr uint64_t kEstimateMinTiles = 8192;  // sample at least 16 Mi samples (or everything)

enum Pass { PASS1 = 0, PASS2 = 1, SWEEP = 2 };

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

// ---- file source ---------------------------------------------------------------
struct FileSrc {
    int fd = -1;
    int fd_direct = -1;  // O_DIRECT view of the same file (PAPR_O_DIRECT=1), -1 when not usable
    uint64_t size = 0, nfloats = 0, nsamples = 0;
    bool odd = false;
    float partner = 0.0f;  // Q of the phantom sample
};

constexpr uint32_t kCapMixed = 256, kCapRaw = 1024;  // (papr_exact.hip sizes its lists by the same numbers: PAPR_EXACT_CAP_*)
constexpr uint32_t kCapRedo = 65536;  // tiles (of 16 KiB) the one-read sweep may have to redo before a full second read is cheaper  // beyond this the program is assembled by the host path

// ---- one-sweep mode: set-up, launch and bookkeeping shared by resident shards and file ingest -------------------
struct SweepRun {
    CcdfPlan bands;               // band edges lo_0 < hi_0 < lo_1 < ... in LUT form
    std::vector<uint32_t> gkeys;  // guessed keys (band centres), unique, ascending
    uint32_t half = 0;            // half-width of a band in bit patterns
    int variant = 0;
    bool lut2 = false;            // the compact two-edges-per-cell table (papr_sweep2_kernel always; papr_sweep_kernel variants 20-29)
    bool v2 = false;              // papr_sweep2_kernel (wave-private segments, compact LUT) instead of papr_sweep_kernel
    bool exact = false;           // v2 / v3: the kernel also builds the exact-sum pairs for speculated binades
    bool v3 = false;              // papr_sweep3_kernel: v2's launch and segments, papr_sweep_kernel's table and bins (always exact)
    int threads = 0;              // v2: workgroup size
    int blocks = 0;               // workgroups of the largest launch (= stash segments)
    uint64_t tile = 0;            // samples per workgroup iteration (v2: per wave segment; exact: per 2048-sample tile)
    size_t stash_lds = 0;         // LDS besides table and histogram (v2: rings + transpose buffers)
    uint32_t nbins = 0;           // v1: 2 * bands + 1 + the NaN trash bin; v2: 2 * bands + 1
    uint32_t seg_off = 0;         // where the per-segment arrays start in d_sweep_hist (0 = right behind the nbins bins)
    uint64_t seg_cap = 0;         // floats per stash segment
};

enum StreamPass { PASS_LOAD_STATS, PASS_STREAM_STATS, PASS_STREAM_CCDF, PASS_STREAM_CCDF_EXACT, PASS_STREAM_NAN };

// Nothing may unwind through the C ABI: the entry points that build std::vector / std::string state run inside this.
template <class F>
inline int guarded(papr_hip_ctx *ctx, F &&body) noexcept
{
    int fail(papr_hip_ctx *ctx, int code, const char *fmt, ...);
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(ctx, PAPR_E_NOMEM, "out of host memory");
    } catch (...) {
        return fail(ctx, PAPR_E_INTERNAL, "unexpected C++ exception inside libpaprhip");
    }
}

// ---- functions shared between the translation units ----
double now_s();
CpuSet numa_cpus_of_device(int device);
int fail(papr_hip_ctx *ctx, int code, const char *fmt, ...);
int env_int(const char *name, int dflt);
void parse_tune_env(papr_hip_tuning *t);
int variant_of(const papr_hip_ctx *ctx, Pass p);
int blocks_of(const papr_hip_ctx *ctx, Pass p);
uint64_t tile_samples(const papr_hip_ctx *ctx, Pass p);
int map_of(const papr_hip_ctx *ctx, Pass p);
int pick_blocks(const papr_hip_ctx *ctx, Pass p, uint64_t ntiles);
int effective_map(const papr_hip_ctx *ctx, Pass p, int blocks);
bool use_nt(const papr_hip_ctx *ctx);
int ensure_partials(papr_hip_ctx *ctx, size_t count);
int ensure_table(papr_hip_ctx *ctx, size_t words);
void release_shard(papr_hip_ctx *ctx);
int ensure_owned_capacity(papr_hip_ctx *ctx, uint64_t nsamples);
void time_begin(papr_hip_ctx *ctx, int kind, uint64_t bytes);
void time_end(papr_hip_ctx *ctx);
void time_begin_kernel(papr_hip_ctx *ctx, int kind, uint64_t bytes);  // around exactly one papr_sweep.hip launch
void time_end_kernel(papr_hip_ctx *ctx);
int ensure_exact_buffers(papr_hip_ctx *ctx);
}  // namespace papr_rt
extern "C" bool papr_exchange_is_identity(const papr_exchange *x);  // papr_exchange.cpp (not part of the ABI)
namespace papr_rt {
struct PeerStep {       // what the single-wait step with peers hands papr_hip_analyze besides the shard's own record
    bool global = false;  // total / before are valid: exchange 1 has happened, in the stream
    papr_stats total;
    double before = 0.0;
};
int stats_sweep_fused(papr_hip_ctx *ctx, papr_exchange *x, int graph, double max_db, float spoil, papr_stats *out, bool *done,
                      PeerStep *peer);
// papr_exchange.cpp: collectives queued on the context's stream (RCCL transport only)
bool xch_in_stream(const papr_exchange *x, const papr_hip_ctx *ctx);
int xch_allgather_host(papr_exchange *x, const void *send, void *recv, size_t bytes_per_rank);  // (host memory; one wait)
int xch_rank(const papr_exchange *x);
int xch_world(const papr_exchange *x);
int xch_allgather_dev(papr_exchange *x, papr_hip_ctx *ctx, const void *send_dev, void *recv_dev, size_t bytes_per_rank);
int xch_allgatherv_dev(papr_exchange *x, papr_hip_ctx *ctx, const void *send_dev, void *recv_dev, const size_t *sizes,
                       const size_t *offs);
int xch_allreduce_u64_dev(papr_exchange *x, papr_hip_ctx *ctx, const void *send_dev, void *recv_dev, size_t count);
void xch_wait_begin(papr_exchange *x, const char *what);  // PAPR_XCH_TIMEOUT_S: this rank now waits for its peers ...
void xch_wait_end(papr_exchange *x);                      // ... and no longer
bool xch_cancelled(const papr_exchange *x);               // papr_exchange_abort has run (by a peer, or the watchdog)
int run_exact_swept(papr_hip_ctx *ctx, double before, uint64_t n_total, const double *before_dev = nullptr,
                    const unsigned long long *n_total_dev = nullptr, unsigned char *slot_dev = nullptr, uint64_t slot_cap = 0);
int mark_program_ready(papr_hip_ctx *ctx);
int reserve_exact_lists(papr_hip_ctx *ctx);            // the small device lists / pinned words run_exact_swept needs
size_t exact_program_slot_bytes(uint64_t nsamples);   // slot of the in-stream program exchange for shards of up to nsamples
const unsigned char *current_program(const papr_hip_ctx *ctx);
void run_overlap_work(papr_hip_ctx *ctx);  // (see papr_hip_ctx::overlap_work)
int launch_stats_range(papr_hip_ctx *ctx, const float *data, uint64_t n, uint64_t base_index, size_t slot,
                       int *nrecords);
void partial_to_stats(const papr_partial &r, uint64_t n, papr_stats *out);
void apply_nan_key(papr_stats *out, unsigned long long key);
int finish_plan(papr_hip_ctx *ctx, CcdfPlan *plan, int vblock, size_t extra_lds);
int plan_ccdf(papr_hip_ctx *ctx, const float *levels, int nlevels, CcdfPlan *plan);
int upload_ccdf_table(papr_hip_ctx *ctx, const CcdfPlan &plan);
int launch_ccdf_range(papr_hip_ctx *ctx, const CcdfPlan &plan, const float *data, uint64_t n);
double page_cache_fraction(int fd, uint64_t size);
void close_file_src(FileSrc *fs);
int open_file_src(papr_hip_ctx *ctx, const char *path, FileSrc *fs);
int read_samples(const FileSrc &fs, uint64_t s0, uint64_t cnt, unsigned char *dst);
int ensure_ingest(papr_hip_ctx *ctx, bool need_device_stage);
hipError_t make_staging(papr_hip_ctx *ctx);
int launch_fused_chunk(papr_hip_ctx *ctx, const CcdfPlan &plan, const float *chunk, uint64_t s0, uint64_t cnt, bool last);
int sweep_prepare(papr_hip_ctx *ctx, const float *guess_levels, int nlevels, uint64_t n_shard, uint64_t n_launch,
                  SweepRun *run, int *reason);
int sweep_launch(papr_hip_ctx *ctx, const SweepRun &run, const float *data, uint64_t n, uint64_t base_index, size_t slot,
                 int *nrecords);
int sweep_fetch(papr_hip_ctx *ctx, const SweepRun &run);
int sweep_collect(papr_hip_ctx *ctx, const SweepRun &run);
int stream_file(papr_hip_ctx *ctx, StreamPass pass, const CcdfPlan *plan, size_t *nrecords_out);
int finish_stats(papr_hip_ctx *ctx, size_t records, const float *tail_ptr, uint32_t tail_samples, uint64_t tail_base,
                 papr_stats *out, const unsigned long long *copy_src = nullptr /* words the finalize kernel copies ... */,
                 unsigned long long *copy_dst = nullptr /* ... to mapped host memory on the way */, uint32_t copy_words = 0);
int resolve_resident_nan(papr_hip_ctx *ctx, papr_stats *out);
int load_file_impl(papr_hip_ctx *ctx, const char *path, uint64_t first_sample, uint64_t nsamples, const float *guess,
                   int nguess);
int exact_preconditions(papr_hip_ctx *ctx, double before, bool allow_restreamed);
int reserve_program(papr_hip_ctx *ctx, size_t want);  // grow the pinned program buffer, keeping its contents
int run_exact_device(papr_hip_ctx *ctx, double before, uint64_t n_total, const CcdfPlan *fused, size_t *bytes);
int assemble_program_on_host(papr_hip_ctx *ctx, const void **program, size_t *bytes);
int run_exact_full_redo(papr_hip_ctx *ctx);
size_t swept_program_bytes(papr_hip_ctx *ctx);
void counts_from_histogram(const papr_hip_ctx *ctx, const CcdfPlan &plan, int nlevels, uint64_t *counts_above);
int resolve_from_sweep(papr_hip_ctx *ctx, const CcdfPlan &plan, const float *levels, int nlevels, uint64_t *counts_above,
                       bool *done);

}  // namespace papr_rt

#endif
