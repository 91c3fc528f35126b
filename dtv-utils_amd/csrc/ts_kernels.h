// ts_kernels.h — shared between ts_kernels.hip and ts_runtime.cpp.  Internal: not part of the C ABI.
#ifndef TS_KERNELS_H
#define TS_KERNELS_H

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

// one PID of one workgroup's span: packets, first and last unit number (within the launch)
struct ts_wg_entry {
    uint32_t pid, count, first, last;
};

struct ts_scan_params {
    const unsigned char *data;  // the stream, file offset 0 at data[0]
    uint64_t nbytes;
    uint64_t first_unit;        // file offset of unit 0 of this launch (a clean position)
    uint64_t nunits;            // units the launch may take (< 2^32)
    uint32_t stride;            // 188, or 192 (HDMV)
    uint32_t sync_offset;       // 0, or 4 (HDMV: behind the tp_extra_header)
    ts_wg_entry *lists;         // per workgroup: up to TS_PIDS entries
    uint32_t *list_counts;      // per workgroup
    unsigned long long *span_done;  // per workgroup: units taken (in front of its first irregular one)
    uint32_t *span_stopped;     // per workgroup: 1 = it met an irregular unit
    uint32_t *events;           // per workgroup: event_cap unit numbers whose packet hit the read-boundary quirk harmlessly
    uint32_t *event_counts;     // per workgroup
    uint32_t event_cap;         // 0 = treat every quirk packet as irregular
    uint32_t *merged_events;    // the valid spans' events, compacted by ts_merge_kernel
    uint32_t merged_event_cap;
};

void ts_kernels_prepare_device(void);
// unroll: packets per lane between two workgroup barriers (1, 2 or 4)
void ts_launch_scan(hipStream_t st, int blocks, int unroll, int block, int agg, const ts_scan_params &p);
int ts_scan_form_exists(int unroll, int block, int agg); /* 1 if that (packets per lane, workgroup size, aggregated update) form is built */
void ts_launch_merge(hipStream_t st, const ts_scan_params &p, uint32_t nspans, uint64_t packet_base, uint32_t *g_count,
                     unsigned long long *g_first, unsigned long long *g_last, unsigned long long *taken_out);
void ts_launch_generate(hipStream_t st, void *out, uint64_t nunits, uint32_t unit, uint64_t seed, int hdmv);

#endif
