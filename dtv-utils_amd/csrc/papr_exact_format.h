/* papr_exact_format.h — layout of a shard's "sum program" (internal; produced by
 * papr_hip_exact_program, consumed by papr_exact_chain).  Host-endian, every
 * block 8-byte aligned:
 *
 *   papr_exact_header
 *   ngroups x { int32 E; int32 pad; double D0, D1 }          group table
 *   nmixed  x { uint64 group; int32 tile_E[128]; double seg_D[256][2] }
 *   nraw    x { uint64 tile; float pw[2048];                  tiles the running sum changes binade in (or may): the powers,
 *               int32 run_E[128]; double run_D[128][2] }      every 16-sample run with the pair of ITS binade — the host
 *                                                             applies it when the sum is in that binade before and after,
 *                                                             and adds the run sample by sample otherwise
 *   float iq[2 * tail_samples]                                the < 1 tile tail, added sample by sample
 */
#ifndef PAPR_EXACT_FORMAT_H
#define PAPR_EXACT_FORMAT_H

#include <stdint.h>

#define PAPR_EXACT_MAGIC 0x31535850u /* "PXS1" */
#define PAPR_EXACT_VERSION 3u

#define PAPR_XF_TILE_SAMPLES 2048
#define PAPR_XF_GROUP_TILES 128
#define PAPR_XF_RUN_SAMPLES 16
#define PAPR_XF_TILE_RUNS (PAPR_XF_TILE_SAMPLES / PAPR_XF_RUN_SAMPLES)
#define PAPR_XF_AMBIG (-2147483647 - 1)
#define PAPR_XF_ZERO (-2147483647)

typedef struct papr_exact_header {
    uint32_t magic, version;
    uint64_t nsamples;
    uint64_t ntiles;
    uint64_t ngroups;
    uint32_t tail_samples;
    uint32_t nmixed;
    uint32_t nraw;
    uint32_t reserved;
} papr_exact_header;

typedef struct papr_exact_group_rec {
    int32_t E, pad;
    double D0, D1;
} papr_exact_group_rec;

typedef struct papr_exact_mixed_rec {
    uint64_t group;
    int32_t tile_E[PAPR_XF_GROUP_TILES];
    double seg_D[2 * PAPR_XF_GROUP_TILES][2];
} papr_exact_mixed_rec;

typedef struct papr_exact_raw_rec {
    uint64_t tile;
    float pw[PAPR_XF_TILE_SAMPLES];      /* fl(fl(I*I) + fl(Q*Q)) of every sample, as the device (and papr.c:103) forms it */
    int32_t run_E[PAPR_XF_TILE_RUNS];    /* the binade run_D was built for; PAPR_XF_AMBIG: none; PAPR_XF_ZERO: sixteen +0 powers */
    double run_D[PAPR_XF_TILE_RUNS][2];
} papr_exact_raw_rec;

#endif
