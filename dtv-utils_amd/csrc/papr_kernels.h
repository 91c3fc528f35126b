// papr_kernels.h — shared between the device code (papr_kernels.hip) and the
// host runtime (papr_runtime.cpp).  Internal: not part of the C ABI.
#ifndef PAPR_KERNELS_H
#define PAPR_KERNELS_H

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "papr_synth.h"

// Launch geometry.  The default streaming variant uses 4-wave workgroups whose
// loop iteration consumes one tile = 8 coalesced 4 KiB rows (256 lanes x 16 B),
// i.e. 32 KiB = 4096 IQ samples; other (block, unroll) variants exist for
// measurement (papr_variant_geometry).  PAPR_BLOCK is the workgroup size of
// the small helper kernels; PAPR_TILE_SAMPLES_MAX bounds every variant's tile
// (shard slack, chunk alignment).
#define PAPR_BLOCK 256
#define PAPR_TILE_SAMPLES_MAX 8192

#define PAPR_MAP_GRID_STRIDE 0
#define PAPR_MAP_BLOCK_SPAN 1
#define PAPR_MAP_XCD_SPAN 2

// One workgroup's (or the final) pass-1 record.
// val/idx order: peak power, re_pos, re_neg, im_pos, im_neg.
struct papr_partial {
    double sum;
    uint64_t idx[5];
    float val[5];
    uint32_t pad;
};

struct papr_ccdf_params {
    uint32_t shift;       // LUT: bit patterns per cell = 1 << shift
    uint32_t cell_lo;     // LUT: first cell in the table
    uint32_t ncells;      // LUT: cells in the table
    uint32_t nkeys;       // unique thresholds (histogram has nkeys + 1 bins)
    uint32_t above_lo;    // LUT: first bit pattern past the table
    uint32_t above_span;  // LUT: 0x7F800000 - above_lo
    uint32_t table_words; // 32-bit words of table to stage into LDS
    uint32_t copies;      // LDS histogram copies per workgroup
    uint32_t search_step; // search: largest power of two <= nkeys
};

int papr_variant_geometry(int variant, int *block, int *unroll); /* 0, or -1 for an unknown variant */
void papr_launch_stats(hipStream_t st, int variant, int blocks, bool nt, const void *data, uint64_t ntiles,
                       uint64_t base_index, int map, papr_partial *out);
void papr_launch_stats_finalize(hipStream_t st, const void *tail, uint32_t tail_samples, uint64_t tail_base_index,
                                const papr_partial *partials, uint32_t npartials, papr_partial *result);
void papr_launch_first_nan(hipStream_t st, int blocks, const void *data, uint64_t nsamples, uint64_t base_index,
                           unsigned long long *key);
void papr_launch_ccdf(hipStream_t st, int variant, int blocks, bool nt, bool lut, size_t lds_bytes, const void *data,
                      uint64_t ntiles, int map, const void *tail, uint32_t tail_samples, const uint32_t *table,
                      const papr_ccdf_params &P, unsigned long long *ghist);
void papr_launch_generate(hipStream_t st, int blocks, void *out, uint64_t nsamples, uint64_t first_index,
                          const papr_synth_spec &spec);
int papr_ccdf_max_dynamic_lds(void);

#endif
