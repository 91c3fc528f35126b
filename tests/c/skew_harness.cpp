// skew_harness — the sweep kernels' walk (dtv-utils_amd/csrc/papr_skew_walk.h) on the host: for `blocks` workgroups over `ntiles`
// tiles with skew period R, every tile is folded exactly once, every workgroup meets its tiles in increasing order, and the odd
// workgroups (the even ones with a fourth argument of 0: whichever sit on the odd XCDs) fold (R - 1) / R of what the others
// fold.  Prints "ok <even share> <odd share>" or what is wrong.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "papr_skew_walk.h"

int main(int argc, char **argv)
{
    if (argc != 4 && argc != 5)
        return 2;
    const uint64_t ntiles = strtoull(argv[1], nullptr, 0);
    const uint32_t blocks = (uint32_t)atoi(argv[2]), R = (uint32_t)atoi(argv[3]);
    const uint32_t slow = argc == 5 ? (uint32_t)atoi(argv[4]) : 1u;
    std::vector<unsigned char> seen(ntiles, 0);
    uint64_t even = 0, odd = 0;
    for (uint32_t b = 0; b < blocks; b++) {
        SkewWalk w;
        w.init(b, blocks, R, slow);
        uint64_t last = 0;
        bool first = true;
        for (uint64_t t = w.tile(); t < ntiles; w.advance(), t = w.tile()) {
            if (!first && t <= last) {
                printf("workgroup %u: tile %llu after %llu\n", b, (unsigned long long)t, (unsigned long long)last);
                return 1;
            }
            if (seen[t]++) {
                printf("tile %llu folded twice (workgroup %u)\n", (unsigned long long)t, b);
                return 1;
            }
            last = t;
            first = false;
            ((b & 1u) ? odd : even)++;
        }
    }
    for (uint64_t t = 0; t < ntiles; t++)
        if (!seen[t]) {
            printf("tile %llu never folded\n", (unsigned long long)t);
            return 1;
        }
    printf("ok %llu %llu\n", (unsigned long long)even, (unsigned long long)odd);
    return 0;
}
