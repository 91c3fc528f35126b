/*
 * papr_oracle.h — CPU restatement of the reference `papr` algorithm.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may build, load or execute it, and there only as the checker.  The
 * product (libpaprhip.so + bin/papr) never links or calls this code.
 *
 * Parity pinning: the reference (drmpeg/dtv-utils) ships no tests or golden
 * vectors for papr.c, so this restatement is pinned against the reference
 * program itself: oracle/Makefile compiles /root/reference/papr.c where it
 * lies into oracle/_ref/papr, tests/golden/make_golden.py records its stdout
 * for the committed fixtures, and tests/test_oracle.py differential-fuzzes
 * this restatement against both.
 */
#ifndef PAPR_ORACLE_H
#define PAPR_ORACLE_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAPR_ORACLE_CHUNK_FLOATS 16384 /* reference papr.c:30 */

typedef struct papr_oracle_result {
    /* pass 1 (reference papr.c:100-129) */
    double sum;      /* sequential double sum of the float powers */
    int64_t n;       /* samples counted, incl. the odd-tail phantom sample */
    float peak;      /* max power, first occurrence */
    int64_t peak_idx;
    float re_pos, re_neg, im_pos, im_neg;
    int64_t re_pos_idx, re_neg_idx, im_pos_idx, im_neg_idx;
    /* host scalars (reference papr.c:131-141 / 164-173) */
    double mean;
    float papr;
    int nlevels;     /* 0 when papr is NaN or negative enough */
    float *level;    /* malloc'd, nlevels entries */
    int64_t *count;  /* malloc'd, nlevels entries: samples with power > level[j] */
} papr_oracle_result;

/* Run both passes over a .cfile exactly as the reference reads it
 * (64 KiB fread chunks, odd-float and stray-byte tail behaviour included).
 * Returns 0, or -1 when the file cannot be opened. */
int papr_oracle_run_file(const char *path, int graph, papr_oracle_result *out);

/* Same arithmetic over an in-memory stream of nfloats floats (even or odd). */
int papr_oracle_run_mem(const float *data, uint64_t nfloats, int graph, papr_oracle_result *out);

/* Pass 2 alone: counts[j] = samples of the stream with power > levels[j]. */
int papr_oracle_count_mem(const float *data, uint64_t nfloats, const float *levels, int nlevels, int64_t *counts);

/* Host scalar stage alone: mean, PAPR and the level table from pass-1 results.
 * Fills mean, papr, nlevels and level[] (caller frees via papr_oracle_free). */
void papr_oracle_levels(papr_oracle_result *r, int graph);

/* The reference's stdout, byte for byte (papr.c:132-135,154-161 / 186-190). */
void papr_oracle_print(const papr_oracle_result *r, int graph, FILE *fp);

void papr_oracle_free(papr_oracle_result *r);

/* Full command line: same argv grammar, messages and exit codes as the
 * reference main() (papr.c:53-98).  Returns the process exit status. */
int papr_oracle_main(int argc, char **argv);

#ifdef __cplusplus
}
#endif
#endif
