/*
 * papr_synth.h — index-addressable synthetic gr_complex IQ generator.
 *
 * Shared verbatim by the host tools (oracle/mkcfile.c), the CPU tests and the
 * HIP generator kernel (dtv-utils_amd/csrc/papr_hip.hip), so a sample with a
 * given (seed, index) has the same 8 bytes everywhere: only 64-bit integer
 * arithmetic plus one int->float conversion and one float multiply, no libm.
 *
 * Format produced: headerless little-endian float32 I,Q pairs, i.e. what
 * GNU Radio's blocks.file_sink(gr.sizeof_gr_complex) writes and what the
 * reference papr reads (reference dvbt-blade.py:214, papr.c:101).
 *
 * Distribution: each component is the centred sum of eight uniform 16-bit
 * integers times `scale` (default 2^-16): near-Gaussian, sigma ~0.8165,
 * |component| <= 4.0, mean power ~1.3333, natural PAPR <= ~13.8 dB.
 */
#ifndef PAPR_SYNTH_H
#define PAPR_SYNTH_H

#include <stdint.h>

#ifdef __HIPCC__
#define PAPR_SYNTH_FN __host__ __device__ static inline
#else
#define PAPR_SYNTH_FN static inline
#endif

#define PAPR_SYNTH_DEFAULT_SEED 0x5EED0001ull
#define PAPR_SYNTH_MAX_OVERRIDES 8

/* A sample whose value is forced (spikes, ties, NaN/Inf injection). */
typedef struct papr_synth_override {
    uint64_t index; /* sample index (not float index) */
    float i, q;
} papr_synth_override;

typedef struct papr_synth_spec {
    uint64_t seed;
    float scale; /* 0 => 2^-16 */
    uint32_t n_overrides; /* bits 0-7: forced samples in ov[]; bits 8-15: envelope, PAPR_SYNTH_ENV_* */
    papr_synth_override ov[PAPR_SYNTH_MAX_OVERRIDES];
} papr_synth_spec;

/* Envelopes (what the one-sweep speculation finds easy or hard):
 *   GAUSS     the default above: stationary, near-Gaussian
 *   CONSTANT  constant envelope (|I| = |Q| = 53510 * scale, signs from the hash: a QPSK-like phase-only signal): every
 *             power is the same number, i.e. every sample sits exactly on the 0 dB threshold
 *   BURSTY    on/off keying in bursts of 4096 samples, a quarter of them silent (exact zeros): a 1/64 sample of such
 *             a capture estimates its mean ten times worse than that of a stationary one */
#define PAPR_SYNTH_ENV_GAUSS 0u
#define PAPR_SYNTH_ENV_CONSTANT 1u
#define PAPR_SYNTH_ENV_BURSTY 2u
#define PAPR_SYNTH_ENV(sp) (((sp)->n_overrides >> 8) & 0xffu)
#define PAPR_SYNTH_NOV(sp) ((sp)->n_overrides & 0xffu)

PAPR_SYNTH_FN uint64_t papr_synth_mix(uint64_t x)
{
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

PAPR_SYNTH_FN uint32_t papr_synth_sum4(uint64_t h)
{
    return (uint32_t)(h & 0xFFFFu) + (uint32_t)((h >> 16) & 0xFFFFu) +
           (uint32_t)((h >> 32) & 0xFFFFu) + (uint32_t)(h >> 48);
}

/* component c (0 = I, 1 = Q) of sample `index`, before overrides */
PAPR_SYNTH_FN float papr_synth_component(uint64_t seed, float scale, uint64_t index, uint32_t c)
{
    uint64_t ctr = seed + (index * 2u + c) * 0x9E3779B97F4A7C15ull;
    uint64_t a = papr_synth_mix(ctr);
    uint64_t b = papr_synth_mix(ctr ^ 0xD1B54A32D192ED03ull);
    int32_t s = (int32_t)(papr_synth_sum4(a) + papr_synth_sum4(b)) - 262140;
    return (float)s * scale;
}

PAPR_SYNTH_FN void papr_synth_sample(const papr_synth_spec *sp, uint64_t index, float *i, float *q)
{
    float scale = sp->scale != 0.0f ? sp->scale : (1.0f / 65536.0f);
    float vi = papr_synth_component(sp->seed, scale, index, 0);
    float vq = papr_synth_component(sp->seed, scale, index, 1);
    const uint32_t env = PAPR_SYNTH_ENV(sp);
    if (env == PAPR_SYNTH_ENV_CONSTANT) {
        vi = (vi < 0.0f ? -53510.0f : 53510.0f) * scale;
        vq = (vq < 0.0f ? -53510.0f : 53510.0f) * scale;
    } else if (env == PAPR_SYNTH_ENV_BURSTY) {
        if ((papr_synth_mix(sp->seed ^ ((index >> 12) * 0xA24BAED4963EE407ull)) & 3u) == 0) {
            vi = 0.0f;
            vq = 0.0f;
        }
    }
    for (uint32_t k = 0; k < PAPR_SYNTH_NOV(sp) && k < PAPR_SYNTH_MAX_OVERRIDES; k++) {
        if (sp->ov[k].index == index) {
            vi = sp->ov[k].i;
            vq = sp->ov[k].q;
        }
    }
    *i = vi;
    *q = vq;
}

/*
 * The bench/golden "spike" workload: two identical spikes of
 * power 1023 * 4/3 (I = 36.9375, Q = 0) at floor(0.731 n) and floor(0.9 n).
 * PAPR ~30.1 dB => 31 default-mode lines, 301-302 "-g" lines; the second
 * spike exercises the first-index tie-break (and lands in another shard when
 * the sample axis is split across GPUs).
 */
PAPR_SYNTH_FN void papr_synth_spike_spec(papr_synth_spec *sp, uint64_t seed, uint64_t n)
{
    sp->seed = seed;
    sp->scale = 0.0f;
    sp->n_overrides = n >= 16 ? 2u : 0u;
    for (int k = 0; k < PAPR_SYNTH_MAX_OVERRIDES; k++) {
        sp->ov[k].index = 0;
        sp->ov[k].i = 0.0f;
        sp->ov[k].q = 0.0f;
    }
    if (n >= 16) {
        sp->ov[0].index = (n / 1000u) * 731u + ((n % 1000u) * 731u) / 1000u;
        sp->ov[0].i = 36.9375f;
        sp->ov[1].index = (n / 10u) * 9u + ((n % 10u) * 9u) / 10u;
        sp->ov[1].i = 36.9375f;
    }
}

#endif /* PAPR_SYNTH_H */
