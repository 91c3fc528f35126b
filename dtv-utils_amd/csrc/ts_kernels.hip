// ts_kernels.hip — gfx950 kernels of the transport-stream packet scan (include/ts_hip.h; reference xport.c).
//
// From a clean sync position every "regular" packet sits at a fixed stride and is independent of the others, so
// one launch takes a whole stretch of them: each lane ONE packet header (8 aligned bytes out of the unit's 188 /
// 192: the scan touches a third of the stream's 64-byte sectors, not its payload), per-workgroup count / first /
// last tables in LDS (3 x 32 KiB), and — because a launch cannot know in advance where the stretch ends — every
// workgroup works on one contiguous span and stops at ITS first irregular packet; ts_merge_kernel then folds the
// tables of the workgroups up to and including the first one that stopped and reports where the stretch ended.
// Irregular = anything whose effect on the reference's state machine is not local to the packet (ts_host.c walks
// those): sync byte missing, packet cut off by the end of the stream, adaptation field longer than the packet, or
// the packet ends exactly one byte past a 16384-byte read of the reference while its payload is skipped in one
// step (xport.c:4302).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ts_hip.h"
#include "ts_kernels.h"
#include "ts_synth.h"

namespace {

constexpr int kMergeBlock = 256;
constexpr uint32_t kNone = 0xFFFFFFFFu;

}  // namespace

// UNR = packets per lane between two workgroup barriers (their header loads are in flight together); kBlock = threads.
// AGG: the per-PID tables are updated once per (wave, PID) instead of once per packet — a transport stream is a
// handful of PIDs, one of them most of the packets, so 64 lanes adding to the same three LDS words is the common case:
// the wave takes the PID of its lowest unserved lane, ballots who else has it (count = popcount; the unit numbers grow
// with the lane, so first = the lowest of them, last = the highest), lets that one lane do the three atomics, and goes
// on with who is left; after eight rounds the remaining lanes (a wave full of different PIDs) update one by one.
template <int UNR, int kBlock, bool AGG>
__global__ __launch_bounds__(kBlock) void ts_scan_kernel(const ts_scan_params p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t ts_smem[];  // 3 x TS_PIDS words = 96 KiB (one workgroup per CU)
    uint32_t *s_count = ts_smem, *s_first = ts_smem + TS_PIDS, *s_last = ts_smem + 2 * TS_PIDS;
    __shared__ uint32_t s_irregular, s_entries, s_events;
    const uint32_t t = threadIdx.x;
    for (uint32_t k = t; k < TS_PIDS; k += kBlock) {
        s_count[k] = 0;
        s_first[k] = kNone;
        s_last[k] = 0;
    }
    if (t == 0) {
        s_irregular = kNone;
        s_entries = 0;
        s_events = 0;
    }
    __syncthreads();

    // this workgroup's span of units (unit = 188 or 192 bytes; the sync byte sits p.sync_offset bytes into it)
    const uint64_t per = (p.nunits + gridDim.x - 1) / gridDim.x;
    const uint64_t j0 = (uint64_t)blockIdx.x * per;
    const uint64_t j1 = j0 + per < p.nunits ? j0 + per : p.nunits;
    for (uint64_t jb = j0; jb < j1; jb += (uint64_t)UNR * kBlock) {  // workgroup-uniform trip count
        // the five bytes that matter — sync, two PID bytes, adaptation_field_control, adaptation_field_length — out of
        // two aligned dwords per packet; all UNR packets' loads issued before the first is looked at
        uint32_t w0[UNR], w1[UNR];
        bool whole[UNR];
#pragma unroll
        for (int r = 0; r < UNR; r++) {
            const uint64_t j = jb + (uint64_t)r * kBlock + t;
            const uint64_t s = p.first_unit + j * p.stride + p.sync_offset;  // file offset of the sync byte
            whole[r] = j < j1 && s + 188 <= p.nbytes;
            const uint64_t a = whole[r] ? (s & ~3ull) : 0ull;
            w0[r] = *reinterpret_cast<const uint32_t *>(p.data + a);
            w1[r] = *reinterpret_cast<const uint32_t *>(p.data + a + 4);
        }
        bool regular[UNR], quirk[UNR];
        uint32_t pid[UNR], tei[UNR];
#pragma unroll
        for (int r = 0; r < UNR; r++) {
            const uint64_t j = jb + (uint64_t)r * kBlock + t;
            const uint64_t s = p.first_unit + j * p.stride + p.sync_offset;
            const uint32_t sh = (uint32_t)(s & 3u);
            const uint32_t lo = __builtin_amdgcn_alignbyte(w1[r], w0[r], sh);  // bytes s .. s+3
            const uint32_t b4 = (w1[r] >> (8 * sh)) & 0xffu;                   // byte s+4 (sh <= 3: inside w1)
            const uint32_t b0 = lo & 0xffu, b1 = (lo >> 8) & 0xffu, b2 = (lo >> 16) & 0xffu, b3 = lo >> 24;
            tei[r] = b1 >> 7;
            pid[r] = ((b1 & 0x1fu) << 8) | b2;
            const bool has_af = (b3 & 0x20u) != 0;
            const uint32_t af_len = has_af ? b4 : 0u;
            regular[r] = whole[r] && b0 == 0x47u && af_len <= 183u;  // (!whole: cut off by the end of the stream)
            quirk[r] = false;
            // the reference's one-step payload skip, entered before the last byte of a packet that ends one byte
            // past a 16384-byte read, finishes the packet a byte early (xport.c:4302): its last byte goes to the
            // sync search.  Unless that byte is 0x47 (a false sync: irregular) the search skips it, reports
            // `skipped 1 bytes` and locks on the next packet where it would have anyway — an event for the list,
            // nothing else changes.
            const bool on_boundary = ((s + 187) & (TS_READ_CHUNK - 1)) == 0;
            if (regular[r] && on_boundary && pid[r] != 0u && pid[r] != 0x1ffbu && (!has_af || af_len <= 181u)) {
                // `skipped 1 bytes` and nothing else happens only if the search that starts on the packet's last byte
                // ends on the NEXT unit's sync byte: that unit must be there, whole, with its 0x47 in place, and the byte
                // the search tests first — the packet's last byte, or in HDMV mode (which swallows four bytes in front
                // of every search, xport.c:4317) the last byte of the next tp_extra_header — must not be a 0x47 itself.
                // Anything else (last packet of the stream: no line at all; damage behind it: ONE line with the sum of
                // the bytes skipped; a false sync: a re-lock one byte early) is the walker's.
                const uint64_t probe = s + 187 + p.sync_offset, next_sync = s + p.stride;
                const bool next_whole = next_sync + 188 <= p.nbytes;
                if (p.event_cap == 0 || !next_whole || p.data[probe] == 0x47u || p.data[next_sync] != 0x47u)
                    regular[r] = false;
                else
                    quirk[r] = true;
            }
            if (j < j1 && !regular[r])
                atomicMin(&s_irregular, (uint32_t)(j - j0));
        }
        __syncthreads();
        const uint32_t stop = s_irregular;  // relative to j0
#pragma unroll
        for (int r = 0; r < UNR; r++) {
            const uint64_t j = jb + (uint64_t)r * kBlock + t;
            if constexpr (AGG) {
                const uint32_t rel_all = (uint32_t)j;
                const bool counts = j < j1 && (uint32_t)(j - j0) < stop && tei[r] == 0;
                unsigned long long todo = __ballot(counts);
                const uint32_t lane = t & 63u;
                for (int round = 0; todo != 0ull; round++) {  // (wave-uniform)
                    if (round == 8) {  // a wave full of different PIDs: the rest one by one
                        if ((todo >> lane) & 1ull) {
                            atomicAdd(&s_count[pid[r]], 1u);
                            atomicMin(&s_first[pid[r]], rel_all);
                            atomicMax(&s_last[pid[r]], rel_all);
                        }
                        break;
                    }
                    const int leader = __ffsll((long long)todo) - 1;
                    const uint32_t lp = (uint32_t)__builtin_amdgcn_readlane((int)pid[r], leader);
                    const unsigned long long same = __ballot(counts && pid[r] == lp);
                    const int top = 63 - __clzll((long long)same);
                    const uint32_t rel_lo = (uint32_t)__builtin_amdgcn_readlane((int)rel_all, leader);
                    const uint32_t rel_hi = (uint32_t)__builtin_amdgcn_readlane((int)rel_all, top);
                    if ((int)lane == leader) {
                        atomicAdd(&s_count[lp], (uint32_t)__popcll(same));
                        atomicMin(&s_first[lp], rel_lo);
                        atomicMax(&s_last[lp], rel_hi);
                    }
                    todo &= ~same;
                }
            }
            if (j < j1 && (uint32_t)(j - j0) < stop) {
                const uint32_t rel = (uint32_t)j;  // unit number within the launch (a launch takes < 2^32 units)
                if (!AGG && tei[r] == 0) {
                    atomicAdd(&s_count[pid[r]], 1u);
                    atomicMin(&s_first[pid[r]], rel);
                    atomicMax(&s_last[pid[r]], rel);
                }
                if (quirk[r]) {  // (about one packet in 4096 of a stream whose packets sit at odd offsets)
                    const uint32_t at = atomicAdd(&s_events, 1u);
                    if (at < p.event_cap)
                        p.events[(size_t)blockIdx.x * p.event_cap + at] = rel;
                }
            }
        }
        if (stop != kNone)
            break;  // (uniform: every thread read the same value after the barrier)
    }
    __syncthreads();
    ts_wg_entry *mine = p.lists + (size_t)blockIdx.x * TS_PIDS;
    for (uint32_t k = t; k < TS_PIDS; k += kBlock) {
        if (s_count[k]) {
            const uint32_t at = atomicAdd(&s_entries, 1u);
            ts_wg_entry e;
            e.pid = k;
            e.count = s_count[k];
            e.first = s_first[k];
            e.last = s_last[k];
            mine[at] = e;
        }
    }
    __syncthreads();
    if (t == 0) {
        p.list_counts[blockIdx.x] = s_entries;
        // units of this span in front of its first irregular one (all of them if there is none)
        p.span_done[blockIdx.x] = s_irregular != kNone ? (uint64_t)s_irregular : (j1 > j0 ? j1 - j0 : 0);
        // more quirk events than the list holds: as good as an irregular packet (the host walker then reports them)
        p.span_stopped[blockIdx.x] = (s_irregular != kNone || s_events > p.event_cap) ? 1u : 0u;
        if (s_events > p.event_cap)
            p.span_done[blockIdx.x] = 0, p.list_counts[blockIdx.x] = 0;
        p.event_counts[blockIdx.x] = s_events <= p.event_cap ? s_events : 0u;
    }
}

// One workgroup per span: fold the tables of the spans up to and including the first that stopped into the stream-wide
// tables (absolute 1-based packet numbers = packet_base + unit number + 1; count: add, first: min over a table that
// starts at all-ones, last: max — order-independent, so the spans go in parallel) and tell the host how many units
// were taken.  Every workgroup works out the stop for itself from the nspans-long span tables (a few hundred words).
__global__ __launch_bounds__(kMergeBlock) void ts_merge_kernel(const ts_scan_params p, uint32_t nspans, uint64_t packet_base,
                                                               uint32_t *__restrict__ g_count,
                                                               unsigned long long *__restrict__ g_first,
                                                               unsigned long long *__restrict__ g_last,
                                                               unsigned long long *__restrict__ taken_out)
{
    __shared__ uint32_t s_stop;
    __shared__ unsigned long long s_taken;
    __shared__ uint32_t s_ev_before, s_ev_total;
    const uint32_t t = threadIdx.x;
    if (t == 0) {
        s_stop = nspans;
        s_taken = 0;
        s_ev_before = 0;
        s_ev_total = 0;
    }
    __syncthreads();
    for (uint32_t b = t; b < nspans; b += kMergeBlock)
        if (p.span_stopped[b])
            atomicMin(&s_stop, b);
    __syncthreads();
    const uint32_t last_span = s_stop < nspans ? s_stop : nspans - 1;  // the span that stopped counts up to its stop
    const uint32_t me = blockIdx.x;
    if (me > last_span)
        return;  // (workgroup-uniform)
    {
        unsigned long long taken = 0;
        uint32_t ev_before = 0, ev_total = 0;
        for (uint32_t b = t; b <= last_span; b += kMergeBlock) {
            taken += p.span_done[b];
            const uint32_t n = p.event_counts[b];
            ev_total += n;
            ev_before += b < me ? n : 0u;
        }
        if (taken)
            atomicAdd(&s_taken, taken);
        if (ev_total)
            atomicAdd(&s_ev_total, ev_total);
        if (ev_before)
            atomicAdd(&s_ev_before, ev_before);
    }
    __syncthreads();
    if (me == 0 && t == 0) {
        taken_out[0] = s_taken;
        taken_out[1] = s_ev_total;
    }
    // this span's quirk events, behind those of the spans in front of it (unordered within a span: the host sorts)
    const uint32_t nev = p.event_counts[me], ev_at = s_ev_before;
    for (uint32_t k = t; k < nev; k += kMergeBlock)
        if (ev_at + k < p.merged_event_cap)
            p.merged_events[ev_at + k] = p.events[(size_t)me * p.event_cap + k];
    const ts_wg_entry *list = p.lists + (size_t)me * TS_PIDS;
    const uint32_t n = p.list_counts[me];
    for (uint32_t k = t; k < n; k += kMergeBlock) {
        const ts_wg_entry e = list[k];
        atomicAdd(&g_count[e.pid], e.count);
        atomicMin(&g_first[e.pid], packet_base + e.first + 1);
        atomicMax(&g_last[e.pid], packet_base + e.last + 1);
    }
}

__global__ __launch_bounds__(256) void ts_generate_kernel(unsigned char *__restrict__ out, uint64_t nunits, uint32_t unit,
                                                           uint64_t seed, int hdmv)
{
    // one thread per 4 bytes (units are multiples of 4 bytes)
    const uint64_t words = nunits * (unit / 4);
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < words; w += (uint64_t)gridDim.x * 256) {
        const uint64_t k = w / (unit / 4);
        const uint32_t i = (uint32_t)(w % (unit / 4)) * 4;
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++)
            v |= (uint32_t)ts_synth_byte(seed, k, i + b, hdmv) << (8 * b);
        reinterpret_cast<uint32_t *>(out)[w] = v;
    }
}

// the (packets per lane, workgroup size, aggregated update) forms that are built: the default and the measurement knobs
// TS_SCAN_UNROLL / TS_SCAN_BLOCK / TS_SCAN_AGG of ts_runtime.cpp
// (measured, tools/gpu_session45.sh: 1.213-1.233 ms for every non-aggregated form, 1.22-1.31 for the aggregated ones —
// neither the LDS atomics nor the geometry is what holds the scan at 0.77-0.78 of peak on its header lines; the default
// stays <1, 1024, false> and the rest is built by `make MEASURE=1` only)
#ifdef PAPR_MEASURE
#define TS_FOR_EACH_SCAN_FORM(X) \
    X(1, 1024, false) X(2, 1024, false) X(4, 1024, false) X(1, 1024, true) X(2, 1024, true) X(4, 1024, true) \
    X(2, 512, false) X(4, 512, false) X(8, 512, false) X(2, 512, true) X(4, 512, true) X(8, 512, true)
#else
#define TS_FOR_EACH_SCAN_FORM(X) X(1, 1024, false) X(2, 1024, false) X(4, 1024, false)
#endif

void ts_kernels_prepare_device(void)  // function attributes belong to the current device
{
    const int lds = 3 * TS_PIDS * (int)sizeof(uint32_t);
#define X(U, B, A) (void)hipFuncSetAttribute((const void *)ts_scan_kernel<U, B, A>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    TS_FOR_EACH_SCAN_FORM(X)
#undef X
}

int ts_scan_form_exists(int unroll, int block, int agg)
{
#define X(U, B, A) if (unroll == U && block == B && (agg != 0) == A) return 1;
    TS_FOR_EACH_SCAN_FORM(X)
#undef X
    return 0;
}

void ts_launch_scan(hipStream_t st, int blocks, int unroll, int block, int agg, const ts_scan_params &p)
{
    const size_t lds = 3 * TS_PIDS * sizeof(uint32_t);
#define X(U, B, A)                                                                                   \
    if (unroll == U && block == B && (agg != 0) == A) {                                               \
        hipLaunchKernelGGL((ts_scan_kernel<U, B, A>), dim3(blocks), dim3(B), lds, st, p);             \
        return;                                                                                       \
    }
    TS_FOR_EACH_SCAN_FORM(X)
#undef X
}

void ts_launch_merge(hipStream_t st, const ts_scan_params &p, uint32_t nspans, uint64_t packet_base, uint32_t *g_count,
                     unsigned long long *g_first, unsigned long long *g_last, unsigned long long *taken_out)
{
    hipLaunchKernelGGL(ts_merge_kernel, dim3(nspans), dim3(kMergeBlock), 0, st, p, nspans, packet_base, g_count, g_first,
                       g_last, taken_out);
}

void ts_launch_generate(hipStream_t st, void *out, uint64_t nunits, uint32_t unit, uint64_t seed, int hdmv)
{
    const uint64_t words = nunits * (unit / 4);
    const int blocks = (int)((words + 255) / 256 < 16384 ? (words + 255) / 256 : 16384);
    if (blocks > 0)
        hipLaunchKernelGGL(ts_generate_kernel, dim3(blocks), dim3(256), 0, st, (unsigned char *)out, nunits, unit, seed, hdmv);
}
