// papr_skew_walk.h — which tiles a persistent workgroup of the sweep kernels folds (see papr_sweep_dev.h).  Plain C++ with no
// device dependency, so that the walk can be checked on the host (tests/c/skew_harness.cpp: every tile exactly once, every
// workgroup's tiles in increasing order, the shares' ratio).
#ifndef PAPR_SKEW_WALK_H
#define PAPR_SKEW_WALK_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PAPR_HD __host__ __device__ __forceinline__
#else
#define PAPR_HD inline
#endif

struct SkewWalk {
    uint64_t base;        // first tile of the current period
    uint64_t period;      // tiles per period: (R - 1) full rounds + one round of the workgroups on the even XCDs
    uint32_t r, rounds_mine, full_rounds, nblocks, b;
    // `slow`: the parity (block & 1) of the workgroups that sit the last round of every period out — those on the odd XCDs,
    // whichever parity the queue's round-robin gave them this time (papr_sweep_rt.cpp: xcd_slow_parity)
    PAPR_HD void init(uint32_t block, uint32_t blocks, uint32_t R, uint32_t slow = 1u)
    {
        b = block;
        nblocks = blocks;
        base = 0;
        r = 0;
        if (R < 2 || (blocks & 7u) != 0) {  // no skew: plain grid stride
            full_rounds = 0xFFFFFFFFu;
            rounds_mine = 0xFFFFFFFFu;
            period = 0;
        } else {
            full_rounds = R - 1;
            rounds_mine = (block & 1u) == (slow & 1u) ? R - 1 : R;
            period = (uint64_t)blocks * (R - 1) + blocks / 2;
        }
    }
    PAPR_HD uint64_t tile() const
    {
        return base + (r < full_rounds ? (uint64_t)r * nblocks + b : (uint64_t)full_rounds * nblocks + (b >> 1));
    }
    PAPR_HD void advance()
    {
        if (++r == rounds_mine) {
            r = 0;
            base += period;
        }
    }
};

#endif
